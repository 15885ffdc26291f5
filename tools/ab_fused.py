#!/usr/bin/env python3
"""A/B of the persistent network kernels on the 256x8 + 256x8 variant: 512x512 frames, arms interleaved, rays/s by wall clock around
synchronised frames.  Arms are settings of the library's bit-identical run-time knobs (name=ENV:VALUE[,ENV:VALUE]).

    python tools/ab_fused.py pipelined=MOFA_FUSED:1 generic=MOFA_PIPE:0 perlayer=MOFA_FUSED:0"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mofanerf_amd import lib, synth

arms = [a.split("=") for a in (sys.argv[1:] or ["pipelined=MOFA_FUSED:1", "generic=MOFA_PIPE:0", "perlayer=MOFA_FUSED:0"])]
bench.ARCH = (8, 256, 8, 256)
dev = torch.device("cuda", 0)
render, kw, args = bench.build_product(dev)
bm, tex, exp = (t.to(dev) for t in synth.codes(0))
K = synth.intrinsics(512, 512)
pose = bench.pose_spherical(0.0, 0.0, 16.0)[:3, :4].to(dev)
def frame():
    with torch.no_grad():
        return render.render_fitting(512, 512, K, chunk=args.chunk, c2w=pose, shapeCodes=bm, uvCodes=tex, expType=20, expCodes=exp, **kw)[0]
def setenv(spec):
    for k in ("MOFA_FUSED", "MOFA_PIPE"):
        os.environ.pop(k, None)
    for kv in spec.split(","):
        k, v = kv.split(":")
        os.environ[k] = v
    lib.reload_env()
res = {n: [] for n, _ in arms}
ref = None
for rnd in range(4):
    for name, spec in arms:
        setenv(spec)
        if rnd == 0:
            out = frame()
            ref = out.clone() if ref is None else ref
            print(f"{name:12s} bit-identical to the first arm: {bool(torch.equal(out, ref))}", flush=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        frame(); frame()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[name].append(2 * 512 * 512 / dt)
for n, v in res.items():
    print(f"{n:12s} " + " ".join(f"{x / 1e3:7.1f}" for x in v) + f"   k rays/s (max {max(v) / 1e3:.1f})", flush=True)
