#!/usr/bin/env python3
"""Timing of BASELINE configs 3 and 5 on one MI355X at the shipped network sizes:
  fit   — run_fit.py photometric step: N_rand = 1024 rays, forward + backward to codes/light (no weight gradients)
  train — run_train.py step: N_rand = 4096 rays, texture encoder, forward + backward incl. weight gradients + Adam
Prints one JSON line per mode."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mofanerf_amd import dist as mdist, lib, schema, steps, synth

dev = torch.device("cuda", 0)
render, kw_test, args = bench.build_product(dev)
L = lib.load()
K = synth.intrinsics(512, 512)


def sample_rays(n, seed):
    c2w = bench.pose_spherical(15.0, 0.0, 16.0)[:3, :4].contiguous().to(dev)
    o, d, v = (torch.empty(512 * 512, 3, device=dev) for _ in range(3))
    lib.check(L.mofa_get_rays(512, 512, 1200., 1200., 256., 256., lib.ptr(c2w), 0, 512 * 512, lib.ptr(o), lib.ptr(d), lib.ptr(v),
                              lib.stream()), "rays")
    idx = torch.from_numpy(np.random.default_rng(seed).choice(512 * 512, n, replace=False)).to(dev)
    return torch.stack([o[idx], d[idx]], 0)


def timed(fn, warm=1, it=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it


fwd_flop = bench.flops_per_ray(True)
for mode in sys.argv[1:] or ["fit", "train"]:
    if mode == "fit":
        n = 1024
        rays = sample_rays(n, 0)
        bm, tex, exp = [t.to(dev).clone().requires_grad_(True) for t in synth.codes(0)]
        light = torch.ones(1, device=dev, requires_grad=True)
        opts = [torch.optim.Adam([bm, tex, exp], lr=1e-3), torch.optim.Adam([light], lr=1e-3)]
        target = torch.rand(n, 3, device=dev)
        kw = dict(kw_test)
        dt = timed(lambda: steps.fit_step(render, kw, opts, 512, 512, K, rays, target, bm, tex, exp, light, chunk=n))
        flop = 2 * fwd_flop * n          # forward + backward-data (no dW)
    else:
        n = 4096
        rays = sample_rays(n, 1)
        kw = dict(kw_test); kw["perturb"] = 1.0
        render.train()
        params = list(kw["network_fn"].parameters()) + list(kw["network_fine"].parameters()) + list(render.grad_parameter())
        opt = torch.optim.Adam(params, lr=5e-5)
        bucket = mdist.GradBucket(params)
        uv = torch.rand(512, 512, 3, device=dev)
        target = torch.rand(n, 3, device=dev)
        bm = synth.codes(0)[0].to(dev).expand(n, -1)
        dt = timed(lambda: steps.train_step(render, kw, opt, bucket, 512, 512, K, rays, target, bm, uv, 3, chunk=n), warm=4, it=4)  # MIOpen searches conv solvers during the first calls
        flop = 3 * fwd_flop * n          # forward + backward-data + weight gradients
    print(json.dumps({"mode": mode, "rays_per_step": n, "ms_per_step": round(dt * 1e3, 2), "rays_per_s": round(n / dt, 1),
                      "algorithmic_tflops": round(flop / dt / 1e12, 2), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}),
          flush=True)
