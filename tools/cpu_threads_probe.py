#!/usr/bin/env python3
"""Time the CPU oracle on a small sample at several thread counts (to choose an honest cpu_baseline setting)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mofanerf_amd import synth
from oracle import mofa_oracle as orc
o = orc.OracleRenderer(synth.nerf_state(8, 256, 0, "coarse"), synth.nerf_state(10, 1024, 0, "fine"), synth.style_state(0),
                       synth.exp_sigma(0), netchunk=196608)
bm, tex, exp = synth.codes(0)
ro, rd = orc.get_rays(512, 512, synth.intrinsics(512, 512), bench.pose_spherical(0., 0., 16.)[:3, :4])
n = 128
ro, rd = ro.reshape(-1, 3)[131072:131072 + n], rd.reshape(-1, 3)[131072:131072 + n]
print("cpu_count", os.cpu_count(), flush=True)
for th in (16, 32, 64, 128, os.cpu_count()):
    torch.set_num_threads(th)
    with torch.no_grad():
        o.render(ro[:8], rd[:8], 8, bm, 20, 8., 26., tex_code=tex, exp_codes=exp, N_samples=64, N_importance=64)
        t0 = time.perf_counter()
        o.render(ro, rd, 4096, bm, 20, 8., 26., tex_code=tex, exp_codes=exp, N_samples=64, N_importance=64)
        dt = time.perf_counter() - t0
    print(f"threads={th}: {n / dt:.1f} rays/s ({dt:.1f} s)", flush=True)
