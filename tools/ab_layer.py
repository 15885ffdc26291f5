#!/usr/bin/env python3
"""Interleaved A/B of layer-kernel variants (separate processes per variant, several rounds).
    python tools/ab_layer.py                      # the round-1 set
    python tools/ab_layer.py name=path/to/lib.so  # base vs the named alternative builds (MOFA_LIB)
Alternative builds: MOFA_SETPRIO=1 MOFA_LIB_OUT=build_arms/prio1.so python -m mofanerf_amd.build --force"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = {"base(2 WG/CU)": {}}
if len(sys.argv) > 1:
    for a in sys.argv[1:]:
        name, path = a.split("=", 1)
        if path.startswith("env:"):                   # name=env:KEY=VALUE[,KEY=VALUE...]  (environment knobs instead of another build)
            variants[name] = dict(kv.split("=", 1) for kv in path[4:].split(","))
        else:                                         # name=path/to/lib.so[,KEY=VALUE...]  (another build, optionally with knobs)
            path, *knobs = path.split(",")
            variants[name] = {"MOFA_LIB": os.path.join(root, path), **dict(k.split("=", 1) for k in knobs)}
else:
    variants.update({"waves3(spills)": {"MOFA_LIB": os.path.join(root, "mofanerf_amd", "libmofanerf_hip_w3.so")}, "BN64(4 WG/CU)": {"MOFA_BN64": "1"}})
code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tools'); import microbench_layer as m; "
        "print(' '.join(f'{m.run(*c, iters=20)[1]:.1f}' for c in [(196608,1024,1024,0),(196608,256,256,0),(32768,1024,1024,0)]), "
        "f'{m.run_bwd(196608,1024,1024,iters=20)[1]:.1f}')" % (root, root))
for rnd in range(3):
    for name, env in variants.items():
        e = dict(os.environ); e.update(env); e["MOFA_STAGE"] = "glds"
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
        print(f"round {rnd} {name:16s} TFLOP/s [W1024 big, W256, W1024 small, BWD W1024]: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
