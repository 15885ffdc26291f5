#!/usr/bin/env python3
"""Timing-only A/B of the persistent network kernel (k_mlp_fused) on the 256x8 + 256x8 variant: one 512x512 frame per arm per
round, arms interleaved, rays/s by wall clock around synchronised frames.  MOFA_DEPHASE carries EXPERIMENT bits (timing-only arms
produce wrong pixels): see the arms list."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mofanerf_amd import lib, synth

arms = [a.split("=") for a in (sys.argv[1:] or ["base=0", "nostore=1", "noepi=2"])]
bench.ARCH = (8, 256, 8, 256)
dev = torch.device("cuda", 0)
render, kw, args = bench.build_product(dev)
bm, tex, exp = (t.to(dev) for t in synth.codes(0))
K = synth.intrinsics(512, 512)
pose = bench.pose_spherical(0.0, 0.0, 16.0)[:3, :4].to(dev)
def frame():
    with torch.no_grad():
        render.render_fitting(512, 512, K, chunk=args.chunk, c2w=pose, shapeCodes=bm, uvCodes=tex, expType=20, expCodes=exp, **kw)
res = {n: [] for n, _ in arms}
for rnd in range(4):
    for name, val in arms:
        os.environ["MOFA_DEPHASE"] = val
        lib.reload_env()
        if rnd == 0:
            frame()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        frame(); frame()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res[name].append(2 * 512 * 512 / dt)
for n, v in res.items():
    print(f"{n:12s} " + " ".join(f"{x / 1e3:7.1f}" for x in v) + f"   k rays/s (max {max(v) / 1e3:.1f})", flush=True)
