#!/usr/bin/env python3
"""Benchmark of the MoFaNeRF ray-marching hot path on MI355X.

Default (`--mode render`, the headline = BASELINE.json configs[1]): one "step" = one 512x512 novel view (262,144 rays;
64 coarse + 128 fine network samples per ray) through the shipped network sizes (coarse 256x8, fine 1024x10;
tools/config_parser.py:17-24) with chunk = netchunk = 196608 (configs/exp_mofanerf.txt:9-10).  Weights are the seeded
synthetic recipe (no checkpoint is downloadable), inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode render|fit|train]

`--gpus N` with N > 1 and no torchrun environment: bench.py re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` (one process per
GPU, RCCL), so `python bench.py --gpus 8` alone produces the line; launched under torchrun it uses the environment it finds.

  render  N > 1: the frame's rows are split into N contiguous blocks, each rank runs the whole coarse->fine pipeline on its
          block, ONE RCCL all-gather of the [rays/N, 5] tiles reassembles the frame on every rank inside the timed region
          (scaling "strong": total work fixed).
  fit     BASELINE configs[2]: run_fit.py's photometric step, N_rand = 1024 rays, forward + backward to codes / light
          (no weight gradients); N > 1 = independent replicas (too small to shard), scaling "weak".
  train   BASELINE configs[4]: run_train.py's step, N_rand = 4096 rays per GPU, texture encoder, forward + backward incl.
          weight gradients, ONE RCCL all-reduce of the flat gradient bucket, Adam; scaling "weak".

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel of the mode (the MFMA kernel kind with the largest summed time):
algorithmic FLOPs of its launches / their summed duration measured with HIP events on the launch stream, against the fp32 MFMA
peak of 157.3 TFLOP/s.  `roofline_hbm` is SURVEY section 8d's second roofline: the HBM-bound compositing / resampling kernels of the
same timed region, algorithmic bytes / HIP-event time against the 8 TB/s HBM peak.
`cpu_baseline` times the CPU oracle on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mofanerf_amd import dist as mdist, factory, lib, schema, steps as msteps, synth  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0              # same guide, "HBM3E (288 GB, 8 TB/s peak)" (~6.3 TB/s achievable by a streaming kernel)
H = W = 512
ARCH = (8, 256, 10, 1024)
N_SAMPLES, N_IMPORTANCE = 64, 64
# (symbol, role, description) per profiler kind of libmofanerf_hip.so (include/mofanerf_hip.h, MOFA_PROF_KINDS); k_layer's template =
# <BN, L0, BWD, PERRAY, PIPE, policy>.  Kinds 0-7 and 11 are the fp32-MFMA kernels (work = FLOPs), 8-10 the HBM-bound ray kernels (work = rays).
KERNELS = [("mofa::k_layer<128,false,false,false,true,mofa::ShippedPolicy>", "forward", "fp32 MFMA Linear+bias+ReLU, software-pipelined K loop"),
           ("mofa::k_mlp_fused", "forward, persistent", "persistent fp32-MFMA network kernel, 256-wide layers pipelined across layer boundaries"),
           ("mofa::k_layer<128,false,true,false,true,mofa::ShippedPolicy>", "BWD backward-data", "fp32 MFMA backward-data GEMM + ReLU mask, the same K loop"),
           ("mofa::k_wgrad<128,256>", "weight gradient", "fp32 MFMA weight-gradient GEMM, contraction over points"),
           ("mofa::k_layer<128,false,false,true,true,mofa::ShippedPolicy>", "forward, PERRAY view layer", "the same kernel with the view layer's per-ray bias"),
           ("mofa::k_net_chain<0>", "forward, chained", "every fp32-MFMA layer of a wide network in one launch: the layer kernel's tiles behind per-XCD queues and "
                                                        "row-tile dependency counters (inference, or keeping the fp32 tape)"),
           ("mofa::k_net_chain<2>", "BWD backward-data, chained", "the backward-data products of a wide network's fitting step in two launches of the same queues"),
           ("mofa::k_net_chain<1>", "forward + mask tape, chained", "the chained forward whose contiguous-store epilogue also writes (y > 0) as one bit per activation "
                                                                    "(the fitting step's forward)"),
           ("mofa::k_composite<1>", "raw2outputs, coarse pass", "one wavefront per ray, 64 samples: coalesced float4 loads of raw, wavefront prefix product"),
           ("mofa::k_composite<2>", "raw2outputs, fine pass", "the same with two samples per lane (128 samples)"),
           ("mofa::k_sample_pdf_merge<false>", "sample_pdf + sort(cat) + std", "one wavefront per ray: cdf (fp64 prefix), inverse-cdf search in LDS, merge"),
           ("mofa::k_net_chain_train", "BWD backward-data + weight gradients, chained", "the training backward of a wide network in two launches: backward-data "
                                                                                        "tiles and the weight gradient's [128 x 256] units behind the same queues")]
MFMA_KINDS = (0, 1, 2, 3, 4, 5, 6, 7, 11)
# ALGORITHMIC HBM bytes per ray of the ray-side kernels: coarse compositing reads raw + z (64 x 20 + 12 B) and writes
# the weights + 5 scalars (256 + 20 B); the fine pass reads 128 x 20 + 12 B and writes its 5 scalars + rgb0 / disp0 / acc0 / z_std (20 + 24 B;
# its weights are an extra the kernel's contract writes but nobody needs: not counted); the resampler reads z + weights (2 x 256 B) and writes
# z_samples, the merged 128 positions and z_std (256 + 512 + 4 B).
HBM_KINDS = {8: 64 * 20 + 12 + 64 * 4 + 20, 9: 128 * 20 + 12 + 20 + 24, 10: 4 * (64 + 64 + 64 + 128 + 1)}


def pose_spherical(phi_deg, theta_deg, radius):
    """Camera-to-world of tools/load_facescape.py:33-38 (host-side, 3 matrix products)."""
    ph, th = np.deg2rad(phi_deg), np.deg2rad(theta_deg)
    t = np.eye(4, dtype=np.float32); t[2, 3] = radius
    rx = np.array([[1, 0, 0, 0], [0, np.cos(th), -np.sin(th), 0], [0, np.sin(th), np.cos(th), 0], [0, 0, 0, 1]], np.float32)
    ry = np.array([[np.cos(ph), 0, -np.sin(ph), 0], [0, 1, 0, 0], [np.sin(ph), 0, np.cos(ph), 0], [0, 0, 0, 1]], np.float32)
    return torch.from_numpy(ry @ (rx @ t))


def flops_per_ray(folded=True):
    Dc, Wc, Df, Wf = ARCH
    return 2 * (N_SAMPLES * schema.mac_per_point(Dc, Wc, folded) +
                (N_SAMPLES + N_IMPORTANCE) * schema.mac_per_point(Df, Wf, folded))


def build_product(device, seed=0, with_tex=False):
    Dc, Wc, Df, Wf = ARCH
    args = factory.default_args(netdepth=Dc, netwidth=Wc, netdepth_fine=Df, netwidth_fine=Wf, no_reload=True,
                                device=device, basedir="/nonexistent", N_samples=N_SAMPLES, N_importance=N_IMPORTANCE)
    _, kw, _, _, _, _, render = factory.create_nerf(args)
    kw["network_fn"].load_state_dict(synth.nerf_state(Dc, Wc, seed, "coarse"))
    kw["network_fine"].load_state_dict(synth.nerf_state(Df, Wf, seed, "fine"))
    render.idSpecificMod.load_state_dict(synth.style_state(seed))
    if with_tex:
        render.texEncoder.load_state_dict(synth.tex_encoder_state(seed))
    for dst, src in zip(render.expCodes_Sigma, synth.exp_sigma(seed)):
        dst.data[:] = src.to(dst.device)
    kw.update(near=8.0, far=26.0)
    return render.eval(), kw, args


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n_rays, seed=0, backward=False, reps=3, tile_reps=1):
    """Time the CPU oracle (restatement of the reference, proven equal to it by tests/test_oracle_golden.py) the way BASELINE.md §3
    lays out: `reps` repetitions of an `n_rays` batch (centre rows of the same 512x512 frame; median reported as `value`) and
    `tile_reps` of the 64x64 tile of BASELINE config 1 (4,096 rays at chunk 4096, K/8; render mode only), no_grad, anomaly
    detection off, on the host cores of this box (`backward`: forward + autograd backward to the codes instead)."""
    from oracle import mofa_oracle as orc
    Dc, Wc, Df, Wf = ARCH
    cores = os.cpu_count() or 1
    o = orc.OracleRenderer(synth.nerf_state(Dc, Wc, seed, "coarse"), synth.nerf_state(Df, Wf, seed, "fine"),
                           synth.style_state(seed), synth.exp_sigma(seed), netchunk=196608)
    bm, tex, exp = synth.codes(seed)
    ro, rd = orc.get_rays(H, W, synth.intrinsics(H, W), pose_spherical(0.0, 0.0, 16.0)[:3, :4])
    b = (H // 2) * W
    ro, rd = ro.reshape(-1, 3)[b:b + n_rays], rd.reshape(-1, 3)[b:b + n_rays]

    def run(ro_, rd_, chunk=4096):
        t0 = time.perf_counter()
        if backward:
            cs = [t.clone().requires_grad_(True) for t in (bm, tex, exp)]
            rgb, _, _, _ = o.render(ro_, rd_, chunk, cs[0], 20, 8.0, 26.0, tex_code=cs[1], exp_codes=cs[2],
                                    N_samples=N_SAMPLES, N_importance=N_IMPORTANCE)
            rgb.abs().mean().backward()
        else:
            with torch.no_grad():
                o.render(ro_, rd_, chunk, bm, 20, 8.0, 26.0, tex_code=tex, exp_codes=exp, N_samples=N_SAMPLES,
                         N_importance=N_IMPORTANCE)
        return time.perf_counter() - t0

    # torch's intra-op pool oversubscribes badly on a 2-socket host (measured: all 256 logical CPUs are 25x slower than 16
    # threads), so "all host cores" is not the fastest setting: the thread count is calibrated on a 64-ray slice and the
    # fastest of 8/16/32/64 is used for every timed repetition (reported as `cores`).
    best, best_t = 1, float("inf")
    for th in sorted({t for t in (8, 16, 32, 64) if t <= cores} | {min(cores, 8)}):
        torch.set_num_threads(th)
        run(ro[:8], rd[:8])
        t = run(ro[:min(64, n_rays)], rd[:min(64, n_rays)])
        if t < best_t:
            best, best_t = th, t
    torch.set_num_threads(best)
    run(ro[:8], rd[:8])
    times = sorted(run(ro, rd) for _ in range(max(1, reps)))
    dt = times[len(times) // 2]
    what = "forward + backward to the codes" if backward else "forward, no_grad"
    out = {"value": round(n_rays / dt, 2), "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
           "cpu_model": cpu_model(), "logical_cpus": cores, "reps": len(times),
           "batch_s": [round(t, 2) for t in times],
           "sample": f"median of {len(times)} passes over {n_rays} rays (centre rows of the same 512x512 frame), same networks/codes, "
                     f"{what}; torch CPU fp32 oracle, anomaly detection off; threads = fastest of 8/16/32/64 on a 64-ray "
                     f"calibration slice ({cores} logical CPUs: the full pool oversubscribes)",
           "full_frame_extrapolated_s": round(H * W / (n_rays / dt), 1)}
    if tile_reps > 0 and not backward:
        Ht = 64
        rot, rdt = orc.get_rays(Ht, Ht, synth.intrinsics(Ht, Ht), pose_spherical(0.0, 0.0, 16.0)[:3, :4])
        rot, rdt = rot.reshape(-1, 3), rdt.reshape(-1, 3)
        tt = sorted(run(rot, rdt, chunk=4096) for _ in range(tile_reps))
        out["tile_64x64"] = {"rays": Ht * Ht, "chunk": 4096, "reps": tile_reps, "median_s": round(tt[len(tt) // 2], 2),
                             "rays_per_s": round(Ht * Ht / tt[len(tt) // 2], 2), "what": "BASELINE.json configs[0] shape (64x64, K/8, chunk 4096)"}
    return out


def parity_sample(render, kw, args, K, frame, angle, bm, tex, exp, dev, n=256, seed=11):
    """SURVEY §8(d) "parity gate beside the timing": `n` rays of the LAST TIMED frame against the CPU oracle, teacher-forced —
    the rays are rendered once more on the device by themselves (results do not depend on the chunking: asserted bit-equal to the
    timed frame's pixels), the coarse pass is compared ray by ray, and the device's own resampled positions are fed to the
    oracle's fine network + compositing, so what is compared is exactly the fused PE -> MLP -> compositing arithmetic
    (tolerance 1e-4 max-abs on RGB / acc, the north star's; end-to-end comparisons see the resampler's 1e-5 branch flip
    between any two fp32 implementations — tests/harness.py)."""
    from oracle import mofa_oracle as orc
    t0 = time.perf_counter()
    idx = torch.from_numpy(np.random.default_rng(seed).choice(H * W, n, replace=False)).sort()[0]
    c2w = pose_spherical(angle, 0.0, 16.0)[:3, :4]
    ro, rd = orc.get_rays(H, W, K, c2w)
    ro, rd = ro.reshape(-1, 3)[idx].contiguous(), rd.reshape(-1, 3)[idx].contiguous()
    with torch.no_grad():
        rgb, disp, acc, ex = render.render_fitting(H, W, K, chunk=args.chunk, rays=torch.stack([ro, rd], 0).to(dev), shapeCodes=bm,
                                                   uvCodes=tex, expType=20, expCodes=exp, verbose=True, **kw)
    same = bool(torch.equal(rgb, frame[idx.to(dev), :3]) and torch.equal(acc, frame[idx.to(dev), 4]))
    Dc, Wc, Df, Wf = ARCH
    o = orc.OracleRenderer(synth.nerf_state(Dc, Wc, 0, "coarse"), synth.nerf_state(Df, Wf, 0, "fine"), synth.style_state(0),
                           synth.exp_sigma(0), netchunk=196608)
    o.exp_sigma.append(exp.cpu())
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    bm_c, tex_c = bm.cpu(), tex.cpu()
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    with torch.no_grad():
        t = torch.linspace(0., 1., N_SAMPLES)
        zc = (8.0 * (1. - t) + 26.0 * t).expand(n, N_SAMPLES)
        raw0 = o.run_network(ro[:, None, :] + rd[:, None, :] * zc[:, :, None], vd, o.coarse, bm_c, tex_c, 20)
        rgb0_r, _, acc0_r, _, _ = orc.raw2outputs(raw0, zc, rd)
        zf = ex["_z_fine"].cpu()
        raw1 = o.run_network(ro[:, None, :] + rd[:, None, :] * zf[:, :, None], vd, o.fine, bm_c, tex_c, 20)
        rgb_r, disp_r, acc_r, _, _ = orc.raw2outputs(raw1, zf, rd)
    # the resampler itself (k_sample_pdf_merge), which teacher forcing by construction does not exercise: the oracle's sample_pdf on the
    # DEVICE's own coarse weights (identical input) against the device's positions — every position must agree within a few ulp of z or
    # be explained by the algorithm's own `denom < 1e-5` branch / conditioning (tests/harness.py::classify_samples)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from harness import classify_samples
    w_dev, zs_dev = ex["_weights0"].cpu(), ex["_z_samples"].cpu()
    zs_ref = orc.sample_pdf(.5 * (zc[:, 1:] + zc[:, :-1]), w_dev[:, 1:-1], torch.linspace(0., 1., N_IMPORTANCE))
    agree, expl = classify_samples(zc, w_dev, torch.linspace(0., 1., N_IMPORTANCE), zs_dev, zs_ref, w_err=0.0)
    zf_sorted = bool((zf[:, 1:] >= zf[:, :-1]).all())
    err = lambda a_, b_: float((a_.cpu() - b_).abs().max())
    out = {"tolerance": 1e-4, "rays": n, "rgb_max_abs": err(rgb, rgb_r), "acc_max_abs": err(acc, acc_r),
           "coarse_rgb_max_abs": err(ex["rgb0"], rgb0_r), "coarse_acc_max_abs": err(ex["acc0"], acc0_r),
           "disp_nan_pattern_equal": bool(torch.equal(torch.isnan(disp.cpu()), torch.isnan(disp_r))),
           "pixels_bit_identical_to_timed_frame": same,
           "resampler": {"positions": int(agree.numel()), "frac_within_6e-6_of_oracle_on_same_weights": round(float(agree.float().mean()), 6),
                         "frac_agree_or_explained": round(float((agree | expl).float().mean()), 6), "merged_positions_sorted": zf_sorted},
           "method": "TEACHER-FORCED vs the CPU oracle (oracle/mofa_oracle.py, pinned to the reference by tests/golden): coarse pass "
                     "ray by ray; the device's resampled positions fed to the oracle's fine network + raw2outputs; the resampler is "
                     "checked separately on identical coarse weights (`resampler`)",
           "seconds": round(time.perf_counter() - t0, 1)}
    out = {k: (float(f"{v:.3e}") if isinstance(v, float) and k.endswith("max_abs") else v) for k, v in out.items()}
    out["pass_teacher_forced"] = bool(same and out["disp_nan_pattern_equal"] and max(out["rgb_max_abs"], out["acc_max_abs"], out["coarse_rgb_max_abs"],
                                                                                    out["coarse_acc_max_abs"]) <= 1e-4)
    out["pass_resampler"] = bool(zf_sorted and out["resampler"]["frac_agree_or_explained"] == 1.0)
    out["pass"] = out["pass_teacher_forced"] and out["pass_resampler"]
    return out


def variant_series(arch, steps, dev, L, bm, tex, exp, K, rays, angles, args):
    """The labelled second series of BASELINE.md section 2 — the same frames with the fine network as small as the coarse one (256 x 8,
    the size BASELINE.json's prose assumes) — timed AFTER the headline's region with the same barrier + synchronize bracket and its own
    HIP-event session: rays/s, ms per frame and the roofline of ITS dominant kernel (the persistent k_mlp_fused)."""
    global ARCH
    keep = ARCH
    ARCH = tuple(arch)
    try:
        render, kw, _ = build_product(dev)

        def frame(i):
            with torch.no_grad():
                return render.render_fitting(H, W, K, chunk=args.chunk, rays=rays[angles[i % len(angles)]], shapeCodes=bm, uvCodes=tex,
                                             expType=20, expCodes=exp, **kw)[0]
        frame(0)
        torch.cuda.synchronize()
        lib.check(L.mofa_prof_begin(), "mofa_prof_begin")
        t0 = time.perf_counter()
        for i in range(steps):
            last = frame(1 + i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        NK = lib.PROF_KINDS
        ms, launches, pflops = (ctypes.c_double * NK)(), (ctypes.c_int64 * NK)(), (ctypes.c_double * NK)()
        lib.check(L.mofa_prof_end(ms, launches, pflops), "mofa_prof_end")
        assert bool(torch.isfinite(last).all())
        dom = max(MFMA_KINDS, key=lambda k: ms[k])
        ach = pflops[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
        return {"workload": f"{H}x{W} novel view, 64 coarse + 128 fine samples/ray, coarse {arch[1]}x{arch[0]} + fine {arch[3]}x{arch[2]} (VARIANT: "
                            "BASELINE.json's prose size; the headline is the shipped 1024x10 fine network)",
                "value": round(H * W * steps / dt, 1), "unit": "rays/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 2),
                "gflop_per_ray_folded": round(flops_per_ray(True) / 1e9, 4),
                "roofline": {"bound": "mfma", "kernel": f"{KERNELS[dom][0]} ({KERNELS[dom][2]})", "achieved": round(ach, 2),
                             "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                             "launches": int(launches[dom]), "avg_launch_ms": round(ms[dom] / max(1, launches[dom]), 4),
                             "share_of_timed_region": round(ms[dom] * 1e-3 / dt, 4)}}
    finally:
        ARCH = keep


def all_rays(L, K, dev, angle, b, n):
    """Rays of pixels [b, b + n) of the H x W view at `angle` (mofa_get_rays: resident in HBM before any timed region)."""
    c2w = pose_spherical(angle, 0.0, 16.0)[:3, :4].contiguous().to(dev)
    o, d, v = (torch.empty(n, 3, dtype=torch.float32, device=dev) for _ in range(3))
    lib.check(L.mofa_get_rays(H, W, float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]), lib.ptr(c2w), b, n,
                              lib.ptr(o), lib.ptr(d), lib.ptr(v), lib.stream()), "mofa_get_rays")
    return o, d


def make_grad_step(mode, n, render, kw, dev, rank, L, K, bm, tex, exp, timed_comm=lambda fn: fn()):
    """One step of BASELINE configs[2] (`fit`: run_fit.py:268-313, N_rand rays, forward + backward to codes / light, Adam) or configs[4]
    (`train`: run_train.py:278-364, N_rand rays per GPU, texture encoder, forward + backward incl. weight gradients, the flat gradient
    bucket's all-reduce, Adam) on `n` seeded rays of the H x W view — the step function the fit / train modes AND the headline
    line's `variants.fit1024` / `variants.train4096` time."""
    n_total = H * W
    o, d = all_rays(L, K, dev, 15.0 + rank, 0, n_total)
    idx = torch.from_numpy(np.random.default_rng(rank).choice(n_total, n, replace=False)).to(dev)
    rays_b = torch.stack([o[idx], d[idx]], 0)
    target = torch.rand(n, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
    if mode == "fit":
        cs = [t.clone().requires_grad_(True) for t in (bm, tex, exp)]
        light = torch.ones(1, device=dev, requires_grad=True)
        opts = [torch.optim.Adam(cs, lr=1e-3), torch.optim.Adam([light], lr=1e-3)]
        return lambda i: msteps.fit_step(render, dict(kw), opts, H, W, K, rays_b, target, cs[0], cs[1], cs[2], light, chunk=n)[0]
    kwt = dict(kw); kwt["perturb"] = 1.0
    render.train()
    params = list(kw["network_fn"].parameters()) + list(kw["network_fine"].parameters()) + list(render.grad_parameter())
    opt = torch.optim.Adam(params, lr=5e-5)
    bucket = mdist.GradBucket(params)
    sync0 = bucket.sync
    bucket.sync = lambda: timed_comm(sync0)
    uv = torch.rand(512, 512, 3, device=dev)
    bm_n = bm.expand(n, -1)
    return lambda i: msteps.train_step(render, kwt, opt, bucket, H, W, K, rays_b, target, bm_n, uv, 3, chunk=n)


def roofline_of(ms, launches, pflops, dt):
    """The dominant MFMA kernel of a timed region (largest summed HIP-event time) against the fp32 matrix peak."""
    dom = max(MFMA_KINDS, key=lambda k: ms[k])
    ach = pflops[dom] / (ms[dom] * 1e-3) / 1e12 if ms[dom] > 0 else 0.0
    others = [{"kernel": KERNELS[k][0], "role": KERNELS[k][1], "launches": int(launches[k]), "total_ms": round(ms[k], 2),
               "tflops": round(pflops[k] / (ms[k] * 1e-3) / 1e12, 2)} for k in MFMA_KINDS if k != dom and ms[k] > 0]
    return dom, {"bound": "mfma", "kernel": f"{KERNELS[dom][0]} ({KERNELS[dom][2]})", "role": KERNELS[dom][1], "achieved": round(ach, 2),
                 "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                 "launches": int(launches[dom]), "avg_launch_ms": round(ms[dom] / max(1, launches[dom]), 4),
                 "algorithmic_gflop_per_launch": round(pflops[dom] / max(1, launches[dom]) / 1e9, 3),
                 "share_of_timed_region": round(ms[dom] * 1e-3 / dt, 4), "other_mfma_kernels": others}


def roofline_hbm_of(ms, launches, work, dt, traffic_json=None, digest=None):
    """SURVEY section 8d's second roofline (VERDICT r5 missing 3): the HBM-bound ray kernels — compositing (coarse / fine pass) and the
    resampler — against the 8 TB/s HBM peak.  `achieved` = algorithmic bytes (SURVEY's per-ray figure x the rays of the launches, `work`)
    / the summed HIP-event time of those launches on their own stream; `traffic` = what the L2s' fabric ports moved per launch by
    rocprofv3 --pmc (profiles/hbm_traffic_rays.json: separate passes on a driver that launches these kernels at the benchmark's
    shapes), quoted only while the kernel sources hash to what the passes were taken on."""
    out = []
    for k, per_ray in HBM_KINDS.items():
        if ms[k] <= 0:
            continue
        n = int(launches[k])
        gbs = work[k] * per_ray / (ms[k] * 1e-3) / 1e9
        rec = {"bound": "hbm", "kernel": f"{KERNELS[k][0]} ({KERNELS[k][2]})", "role": KERNELS[k][1], "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS,
               "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes_per_ray": per_ray, "launches": n,
               "rays_per_launch_avg": round(work[k] / max(1, n), 1), "avg_launch_us": round(ms[k] / max(1, n) * 1e3, 2),
               "share_of_timed_region": round(ms[k] * 1e-3 / dt, 6), "traffic": None}
        tj = (traffic_json or {}).get("kernels", {}).get(KERNELS[k][0])
        if tj is not None:
            if traffic_json.get("csrc_sha256") == digest:
                rec["traffic"] = tj.get("bytes_per_launch")
                rec["traffic_rays_per_launch"] = tj.get("rays_per_launch")
                rec["traffic_algorithmic_bytes_same_shape"] = tj.get("algorithmic_bytes_per_launch")
                if tj.get("sq_insts_valu_per_ray"):
                    # what actually binds these kernels (round 6): a ray is ONE wavefront, and its vector instructions alone — exp / sigmoid /
                    # division, the prefix product and five butterflies; the fp64 cdf scan and searches — take this long on the chip's 1,024
                    # SIMDs (4 cycles per wave64 instruction at the measured 2.39 GHz): SURVEY 8d classed them HBM-bound a priori
                    floor_us = rec["rays_per_launch_avg"] * tj["sq_insts_valu_per_ray"] * 4 / (1024 * 2.39e9) * 1e6
                    rec["valu_insts_per_ray_pmc"] = tj["sq_insts_valu_per_ray"]
                    rec["valu_issue_floor_us"] = round(floor_us, 2)
                    rec["binding"] = ("valu issue" if floor_us >= 0.8 * rec["avg_launch_us"] else
                                      "launch latency (a few thousand rays per launch)" if rec["avg_launch_us"] < 25.0 else "hbm")
            else:
                rec["traffic_source"] = f"null: profiles/hbm_traffic_rays.json was taken on kernel sources {str(traffic_json.get('csrc_sha256'))[:16]}, this build is {str(digest)[:16]}"
        out.append(rec)
    return out


def bulk_variant(dev, size=256, identities=2, expressions=2, views=3, workers=4):
    """BASELINE configs[3] inside the headline's line (VERDICT r5 missing 4): render_refine_trainSet.py's job shape (:238-295) on one GPU —
    `identities` x `expressions` x `views` frames at `size`^2 through `render_path` (texture encoder on the identity's 512^2 UV map, cached
    per map; the shipped network sizes) with ONE `PngSink` for the job writing PNGs into a temporary directory: rays/s INCLUDING encoder,
    quantisation, D2H and PNG encoding (every file on disk inside the timed region), the same loop with the output stage disabled
    (`savedir=None`: frames still return to the host as `render_path` does) on one identity, and the sink's queue high-water mark —
    the evidence for SURVEY section 8(f3)'s claim that the output stage is off the critical path."""
    import shutil
    import tempfile
    from mofanerf_amd import rays as mrays
    from mofanerf_amd.io import PngSink
    render, kw, args = build_product(dev, with_tex=True)
    K = synth.intrinsics(size, size)

    def job(idents, exprs, angles, out_dir):
        n = 0
        with torch.no_grad():
            for ident in idents:
                shape = synth.codes(ident)[0].to(dev)
                uv = torch.from_numpy(np.random.default_rng(ident).uniform(0, 1, (1, 512, 512, 3)).astype(np.float32)).to(dev)
                d = None
                if out_dir is not None:
                    d = os.path.join(out_dir, f"{ident:03d}")
                    os.makedirs(d, exist_ok=True)
                for e in range(exprs):
                    for v, ang in enumerate(angles):
                        pose = mrays.pose_spherical(float(ang), 0.0, 16.0)[None]
                        render.render_path(pose, [size, size, float(K[0][0])], K, args.chunk, kw, uvMap=uv, expType=torch.tensor([e]), savedir=d,
                                           shapeCodes=shape, name=f"{e:02d}_{v}")
                        n += 1
        return n

    angles = list(np.linspace(-60, 60, views))
    job([1000], 1, angles[:1], None)                       # untimed: MIOpen picks its convolution solvers, the nets are packed
    torch.cuda.synchronize()
    tmp = tempfile.mkdtemp(prefix="mofa_bulk_")
    try:
        render.png_sink = PngSink(workers=workers)         # one sink for the job: PNG encoding overlaps the following frames
        t0 = time.perf_counter()
        n_sink = job(list(range(identities)), expressions, angles, tmp)
        render.png_sink.close()                            # every file is on disk inside the timed region
        torch.cuda.synchronize()
        dt_sink = time.perf_counter() - t0
        high, render.png_sink = render.png_sink.high_water, None
        files = [os.path.join(r, f) for r, _, fs in os.walk(tmp) for f in fs if f.endswith(".png")]
        png_bytes = sum(os.path.getsize(f) for f in files)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    t0 = time.perf_counter()
    n_plain = job([identities], 1, angles, None)            # a fresh identity (its texture code is not cached), one expression, no output stage
    torch.cuda.synchronize()
    dt_plain = time.perf_counter() - t0
    render.check_launches(block=True)
    assert len(files) == n_sink, (len(files), n_sink)
    return {"workload": f"render_refine_trainSet.py job shape: {identities} identities x {expressions} expressions x {views} views at {size}x{size} "
                        f"through render_path (texture encoder per identity, coarse {ARCH[1]}x{ARCH[0]} + fine {ARCH[3]}x{ARCH[2]}, 64 + 128 samples/ray), one PngSink "
                        f"({workers} workers) writing to a temporary directory (BASELINE.json configs[3], one GPU's share)",
            "value": round(n_sink * size * size / dt_sink, 1), "unit": "rays/s",
            "includes": "texture encoder, device quantisation (to8b), pinned D2H, PNG encoding, files on disk",
            "frames": n_sink, "seconds": round(dt_sink, 3), "ms_per_frame": round(dt_sink / n_sink * 1e3, 2), "png_files": len(files), "png_bytes": png_bytes,
            "sink_queue_high_water": int(high), "sink_workers": workers,
            "without_output_stage": {"value": round(n_plain * size * size / dt_plain, 1), "unit": "rays/s", "frames": n_plain, "seconds": round(dt_plain, 3),
                                     "ms_per_frame": round(dt_plain / n_plain * 1e3, 2),
                                     "what": "the same loop with savedir=None on one further identity and one expression (frames still return to the host as numpy arrays, as render_path does)"},
            "output_stage_cost_fraction": round(1.0 - (dt_plain / n_plain) / (dt_sink / n_sink), 4)}


def step_variant(mode, n, steps, warmup, dev, L, K, bm, tex, exp):
    """BASELINE configs[2] / configs[4] inside the headline's line (VERDICT r4 missing 3): a fresh product of the shipped sizes, `warmup`
    untimed + `steps` timed steps of `make_grad_step` with the same synchronize bracket and its own HIP-event session, timed AFTER the
    headline's region.  One GPU: no collective (the N > 1 forms are `--mode fit|train --gpus N`)."""
    render, kw, _ = build_product(dev, with_tex=(mode == "train"))
    step = make_grad_step(mode, n, render, kw, dev, 0, L, K, bm, tex, exp)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    lib.check(L.mofa_prof_begin(), "mofa_prof_begin")
    t0 = time.perf_counter()
    for i in range(steps):
        last = step(warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    NK = lib.PROF_KINDS
    ms, launches, pflops = (ctypes.c_double * NK)(), (ctypes.c_int64 * NK)(), (ctypes.c_double * NK)()
    lib.check(L.mofa_prof_end(ms, launches, pflops), "mofa_prof_end")
    peak_extra = torch.cuda.max_memory_allocated(dev) - mem0
    render.check_launches(block=True)
    assert bool(torch.isfinite(last).all())
    _, roof = roofline_of(ms, launches, pflops, dt)
    work = flops_per_ray(True) * (2 if mode == "fit" else 3)
    what = {"fit": f"run_fit.py photometric step: {n} rays of a {H}x{W} view, forward + backward to codes / light (mask-only tape), Adam "
                   "(BASELINE.json configs[2])",
            "train": f"run_train.py step: {n} rays of a {H}x{W} view, texture encoder, forward + backward incl. weight gradients, flat gradient "
                     "bucket, Adam (BASELINE.json configs[4], one GPU's share)"}[mode]
    return {"workload": what + f", coarse {ARCH[1]}x{ARCH[0]} + fine {ARCH[3]}x{ARCH[2]}", "value": round(n * steps / dt, 1), "unit": "rays/s",
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2), "gflop_per_ray_folded": round(work / 1e9, 4),
            "whole_path_tflops": round(work * n * steps / dt / 1e12, 2), "roofline": roof,
            "step_peak_extra_memory_gb": round(peak_extra / 2 ** 30, 3)}


def self_spawn(n):
    """`python bench.py --gpus N` without a torchrun environment: re-execute under torch.distributed.run (one rank per GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL buffer registration)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    global H, W, ARCH
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", choices=["render", "fit", "train"], default="render",
                    help="render = the headline (BASELINE configs[1]); fit / train = configs[2] / configs[4] (forward + backward)")
    ap.add_argument("--cpu-rays", type=int, default=None, help="rays in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="repetitions of the CPU-baseline batch (median reported)")
    ap.add_argument("--cpu-tile-reps", type=int, default=1, help="repetitions of the 64x64 CPU tile (BASELINE.md section 3; 0 = skip; ~45 s each)")
    ap.add_argument("--parity-rays", type=int, default=256, help="rays of the last timed frame checked against the CPU oracle (0 = skip)")
    ap.add_argument("--arch", type=int, nargs=4, default=list(ARCH), metavar=("Dc", "Wc", "Df", "Wf"),
                    help="network sizes; default = shipped config (8 256 10 1024).  '8 256 8 256' is the labelled variant "
                         "BASELINE.md lists (fine net as small as the coarse one)")
    ap.add_argument("--netchunk", type=int, default=None, help="labelled variant: points per network launch (default 196608, the reference's)")
    ap.add_argument("--tape", choices=["keep", "recompute", "fp32"], default="keep",
                    help="fit / train: what the forward keeps for the backward.  keep (default) = training: every layer output (fp32 tape); "
                         "fitting: one bit per activation (mask-only tape).  fp32 = fitting with the fp32 tape (labelled A/B arm).  recompute = "
                         "re-run each sub-batch's forward inside its backward (labelled variant: Renderer.tape_recompute)")
    ap.add_argument("--size", type=int, default=512, help="image side (512 = the benchmark; smaller only for functional tests)")
    ap.add_argument("--variant-steps", type=int, default=3, help="frames of the labelled fine-256x8 series after the headline loop (BASELINE.md "
                    "section 2 asks for both series; render mode, N = 1, shipped --arch only; 0 = skip)")
    ap.add_argument("--fit-steps", type=int, default=10, help="steps of BASELINE configs[2] (run_fit.py step, 1,024 rays) timed after the headline "
                    "loop into variants.fit1024 (render mode, N = 1, shipped sizes; 0 = skip)")
    ap.add_argument("--train-steps", type=int, default=4, help="steps of BASELINE configs[4] (run_train.py step, 4,096 rays) timed after the "
                    "headline loop into variants.train4096 (same conditions; 0 = skip)")
    ap.add_argument("--bulk-identities", type=int, default=2, help="identities of BASELINE configs[3]'s job shape (x 2 expressions x 3 views at 256^2 "
                    "through render_path with the texture encoder and the PNG sink) timed after the headline loop into variants.bulk256 (same conditions; 0 = skip)")
    ap.add_argument("--rays", type=int, default=None, help="fit / train: N_rand per GPU (default 1024 / 4096)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(a.gpus))
    H = W = a.size
    ARCH = tuple(a.arch)
    d_steps, d_warm = {"render": (2, 1), "fit": (10, 3), "train": (4, 4)}[a.mode]   # train: MIOpen searches conv solvers first
    a.steps = d_steps if a.steps is None else a.steps
    a.warmup = d_warm if a.warmup is None else a.warmup
    if a.cpu_rays is None:
        a.cpu_rays = {"render": 1024, "fit": 256, "train": 0}[a.mode]

    rank, world, local = mdist.init_from_env()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU fallback for the product path"
    local = local % torch.cuda.device_count()     # (several ranks may share a GPU only in the gloo functional test)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    L = lib.load()

    render, kw, args = build_product(dev, with_tex=(a.mode == "train"))
    if a.netchunk:
        render.netchunk = int(a.netchunk)
    render.tape_recompute = a.tape == "recompute"
    if a.tape == "fp32":
        render.fit_tape = "fp32"
    bm, tex, exp = (t.to(dev) for t in synth.codes(0))
    K = synth.intrinsics(H, W)
    n_total = H * W
    comm_ms = []                                                       # per-step collective time on this rank (N > 1)
    compute_ev = []                                                    # per-step time of this rank's row block (render, N > 1)

    def timed_comm(fn):
        if not mdist.active():
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        comm_ms.append((e0, e1))
        return out

    if a.mode == "render":
        b, e = mdist.shard_range(n_total, rank, world, align=W)           # whole image rows per rank
        angles = [0.0, -60.0, 60.0]                                         # run_fit.py's three novel views
        rays = {ang: torch.stack(all_rays(L, K, dev, ang, b, e - b), 0) for ang in angles}   # inputs resident in HBM before timing
        frame_buf = [None]
        units_per_step, scaling = n_total, "strong"

        def step(i):
            r = rays[angles[i % len(angles)]]
            c0 = torch.cuda.Event(enable_timing=True) if mdist.active() else None
            if c0 is not None:
                c0.record()
            with torch.no_grad():                                           # render-only, as run_fit.py's novel-view loop
                rgb, disp, acc, _ = render.render_fitting(H, W, K, chunk=args.chunk, rays=r, shapeCodes=bm, uvCodes=tex,
                                                          expType=20, expCodes=exp, **kw)
            tile = torch.cat([rgb, disp[:, None], acc[:, None]], -1)        # [rays/N, 5]
            if c0 is not None:                                              # this rank's own share of the frame, without the exchange
                c1 = torch.cuda.Event(enable_timing=True)
                c1.record()
                compute_ev.append((c0, c1))
            frame_buf[0] = timed_comm(lambda: mdist.all_gather_tiles(tile, n_total, world, rank, align=W, out=frame_buf[0]))
            return frame_buf[0]
    else:
        n = a.rays or (1024 if a.mode == "fit" else 4096)
        units_per_step, scaling = n * world, "weak"
        step = make_grad_step(a.mode, n, render, kw, dev, rank, L, K, bm, tex, exp, timed_comm)

    def sync():
        if mdist.active():
            mdist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    sync()
    comm_ms.clear()
    compute_ev.clear()
    mem0 = torch.cuda.memory_allocated(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    lib.check(L.mofa_prof_begin(), "mofa_prof_begin")
    t0 = time.perf_counter()
    for i in range(a.steps):
        last = step(a.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    NK = lib.PROF_KINDS
    ms, launches, pflops = (ctypes.c_double * NK)(), (ctypes.c_int64 * NK)(), (ctypes.c_double * NK)()
    lib.check(L.mofa_prof_end(ms, launches, pflops), "mofa_prof_end")
    peak_extra = torch.cuda.max_memory_allocated(dev) - mem0           # what one step allocates on top of the resident state
    dt = mdist.barrier_max(dt, dev)
    comm = sum(e0.elapsed_time(e1) for e0, e1 in comm_ms) / max(1, len(comm_ms)) if comm_ms else 0.0
    render.check_launches(block=True)           # a chained launch that ended incomplete is an error, never a number
    compute_minmax = None
    if compute_ev and mdist.active():           # who is the slow rank, and by how much: min / max over ranks of the mean per-step compute time
        mine = sum(e0.elapsed_time(e1) for e0, e1 in compute_ev) / len(compute_ev)
        t = torch.tensor([mine, -mine], dtype=torch.float64, device="cpu" if torch.distributed.get_backend() == "gloo" else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        compute_minmax = (round(-float(t[1]), 3), round(float(t[0]), 3))
    if a.mode == "render":
        assert last.shape == (n_total, 5) and bool(torch.isfinite(last[:, :3]).all())
    else:
        assert bool(torch.isfinite(last).all())

    if rank == 0:
        units_per_s = units_per_step * a.steps / dt
        # dominant kernel = the MFMA kernel kind with the largest summed time.  Its algorithmic FLOPs are 2*M*K*N of its
        # launches; at the benchmark sizes nothing is padded (K, N multiples of 64, M a multiple of 256), except layer 0's
        # K = 63 -> 64 inside the persistent kernel (0.1 %).
        dom, roof = roofline_of(ms, launches, pflops, dt)
        ksym, krole, kdesc = KERNELS[dom]
        kname = f"{ksym} ({kdesc})"
        traffic, tinfo = None, {}
        # PMC-derived bytes/launch (separate --pmc passes), one record per kernel
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic_chain.json" if dom == 5 else "hbm_traffic.json")
        if os.path.exists(tpath) and dom in (0, 5) and a.mode == "render" and ARCH == (8, 256, 10, 1024):
            from mofanerf_amd import build as mbuild
            tj = json.load(open(tpath))
            digest = mbuild.csrc_digest()
            # the counters were collected on a single-layer driver of the SAME kernel in separate --pmc passes (rocprofv3 cannot
            # wrap this whole process); they are quoted only while they describe the kernel that just ran: same instantiation, and
            # the kernel sources hash to what the passes were taken on — otherwise `traffic` is null and says why
            if tj.get("csrc_sha256") == digest and tj.get("kernel", "").split(" ")[0] == kname.split(" ")[0]:
                traffic = tj.get("bytes_per_launch")
                tinfo = {"traffic_shape": tj.get("shape"), "traffic_algorithmic_bytes_same_shape": tj.get("algorithmic_bytes_per_launch"),
                         "mfma_busy_fraction_pmc": tj.get("mfma_busy_fraction"), "traffic_csrc_sha256": digest[:16],
                         "traffic_source": f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc passes on a driver that launches this kernel alone, "
                                           "same kernel sources by hash; not this run)"}
            else:
                tinfo = {"traffic_source": f"null: profiles/{os.path.basename(tpath)} was taken on kernel sources {str(tj.get('csrc_sha256'))[:16]} / "
                                           f"{tj.get('kernel')}, this build is {digest[:16]} / {kname.split(' ')[0]} — re-run tools/gpu_profile_round.sh"}
        # SURVEY section 8d's second roofline: the HBM-bound ray kernels of the same timed region (compositing, resampling)
        from mofanerf_amd import build as mbuild2
        rpath = os.path.join(ROOT, "profiles", "hbm_traffic_rays.json")
        roof_hbm = roofline_hbm_of(ms, launches, pflops, dt, json.load(open(rpath)) if os.path.exists(rpath) else None, mbuild2.csrc_digest())
        fwd = flops_per_ray(True)
        work = {"render": fwd, "fit": 2 * fwd, "train": 3 * fwd}[a.mode]    # + backward-data (+ weight gradients)
        metric = {"render": "rendered rays/sec (64c+128f samples) at 512^2 novel-view",
                  "fit": "fitted rays/sec (run_fit.py photometric step: forward + backward to codes/pose/light, N_rand=1024)",
                  "train": "trained rays/sec (run_train.py step: forward + backward + weight gradients + Adam, N_rand=4096/GPU)"}[a.mode]
        workload = {"render": f"{H}x{W} novel view", "fit": f"{units_per_step // world} rays/GPU of a {H}x{W} view, L1 loss",
                    "train": f"{units_per_step // world} rays/GPU of a {H}x{W} view, MSE(rgb)+MSE(rgb0), texture encoder, Adam"}[a.mode]
        par = {"render": f"ray-rows x{world} + all-gather", "fit": f"{world} independent replicas",
               "train": f"data-parallel x{world} + gradient all-reduce"}[a.mode]
        out = {
            "metric": metric, "value": round(units_per_s, 1),
            "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 2), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{workload}, 64 coarse + 128 fine samples/ray, coarse {ARCH[1]}x{ARCH[0]} + fine {ARCH[3]}x{ARCH[2]}, "
                                   f"chunk=196608, netchunk={int(a.netchunk) if a.netchunk else 196608}{' (VARIANT: the benchmark is netchunk=196608)' if a.netchunk and a.netchunk != 196608 else ''}, seeded Xavier weights (BASELINE.json configs[{ {'render': 1, 'fit': 2, 'train': 4}[a.mode] }])",
                       "mode": a.mode, **({"tape": "recompute (VARIANT: one extra forward per step, tape bounded by netchunk)"} if a.tape == "recompute" else {}),
                       **({"tape": "fp32 (VARIANT: the fitting default is the mask-only tape, one bit per activation)"} if a.tape == "fp32" else {}),
                       **({"step_peak_extra_memory_gb": round(peak_extra / 2 ** 30, 3)} if a.mode != "render" else {}),
                       "rays_per_step": units_per_step, "rays_per_rank_per_step": units_per_step // world,
                       "parallelism": par,
                       "gflop_per_ray_folded": round(work / 1e9, 4),
                       "gflop_per_ray_nominal": round(work / fwd * flops_per_ray(False) / 1e9, 4)},
            "whole_path_tflops": round(work * units_per_s / 1e12, 2),
            "roofline": {**{k: v for k, v in roof.items() if k != "other_mfma_kernels"}, "traffic": traffic,
                         "other_mfma_kernels": roof["other_mfma_kernels"], **tinfo},
            "roofline_hbm": roof_hbm,
        }
        if mdist.active():
            out["rccl_ranks"] = world
            out["backend"] = torch.distributed.get_backend()
            if mdist.shared_device:       # the single-GPU functional form of an N-rank job: the line validates the N > 1 code path, it is NOT a scaling point
                out["functional_only"] = (f"{mdist.shared_device[0]} ranks share {mdist.shared_device[1]} device(s): per-layer launches, gloo staging through the host; "
                                          "the aggregate rate is one device's — not a scaling measurement")
            out["collective"] = {"what": {"render": "all_gather_into_tensor of the [rays/N,5] fp32 tiles, written straight into the frame",
                                          "fit": "none (replicas)", "train": "all_reduce of the flat fp32 gradient bucket"}[a.mode],
                                 "avg_ms_per_step_rank0": round(comm, 4)}
            if compute_minmax is not None:
                out["collective"]["compute_ms_per_step_min_over_ranks"], out["collective"]["compute_ms_per_step_max_over_ranks"] = compute_minmax
        if a.mode == "render":
            # what the timed region produced, so that an N > 1 line can be checked against the N = 1 line of the same view: the rows are
            # rendered by different ranks but chunk invariance is bit-exact, so the gathered frame must hash to the same digest
            import hashlib
            out["frame_sha256"] = hashlib.sha256(last.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
            out["frame_view_deg"] = angles[(a.warmup + a.steps - 1) % len(angles)]
        if a.mode == "render" and a.parity_rays > 0:
            # ANY world size: the sample is drawn over the whole GATHERED frame — rows other ranks rendered included — re-rendered on this
            # rank's device (bit-equality with the gathered pixels = the exchange moved the right bytes) and teacher-forced against the oracle
            out["parity"] = parity_sample(render, kw, args, K, last, angles[(a.warmup + a.steps - 1) % len(angles)], bm, tex, exp, dev,
                                          n=a.parity_rays)
        if world == 1 and a.mode == "render" and ARCH == (8, 256, 10, 1024) and (H, W) == (512, 512) and not a.netchunk:
            out["variants"] = {}
            if a.variant_steps > 0:
                out["variants"]["fine256x8"] = variant_series((8, 256, 8, 256), a.variant_steps, dev, L, bm, tex, exp, K, rays, angles, args)
            if a.fit_steps > 0:
                out["variants"]["fit1024"] = step_variant("fit", 1024, a.fit_steps, 3, dev, L, K, bm, tex, exp)
            if a.train_steps > 0:
                out["variants"]["train4096"] = step_variant("train", 4096, a.train_steps, 4, dev, L, K, bm, tex, exp)
            if a.bulk_identities > 0:
                out["variants"]["bulk256"] = bulk_variant(dev, identities=a.bulk_identities)
        if world == 1 and a.cpu_rays > 0:
            out["cpu_baseline"] = cpu_baseline(a.cpu_rays, backward=(a.mode == "fit"), reps=a.cpu_reps,
                                               tile_reps=a.cpu_tile_reps if (H, W) == (512, 512) else 0)
        print(json.dumps(out), flush=True)
    if mdist.active():
        mdist.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
