"""BASELINE configs [2] and [4] at FULL size through the boundary, with gradient parity.

* config 2 — run_fit.py's photometric step (run_fit.py:281-313): N_rand = 1024 rays of a 512x512 view, coarse 256x8 + fine 1024x10,
  `render_fitting` with autograd (tape forward, compositing / encoding / network backward, pose gradient), L1(light * rgb, target).
* config 4 — run_train.py's step (run_train.py:333-357): N_rand = 4096 rays, `render` (texture encoder on the UV map, stratified
  jitter), MSE(rgb) + MSE(rgb0), weight gradients, the flat gradient bucket, Adam.

Parity: the gradient of the loss restricted to a 128-ray subsample is taken THROUGH the full-size graph on the device (the other
rays contribute zero upstream gradients) and compared with the CPU oracle's autograd on the same 128 rays, teacher-forced on the
device's own sample positions (the resampling is detached in the reference too, models/render_class.py:326).  Yardstick as in
tests/test_gpu_grads.py: the oracle run in DOUBLE on the same inputs is the truth; the oracle in fp32 (= what the reference's own
autograd delivers) is the allowance — HIP must be as close to the truth as that (<= 2x its distance + a floor) and point the same
way (cosine >= 0.9999)."""
import os

import numpy as np
import pytest
import torch

from harness import make_product
from mofanerf_amd import dist as mdist, rays as mrays, steps, synth
from oracle import mofa_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
ARCH = (8, 256, 10, 1024)
H = 512


def _cos(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def _judge(name, hip, g32, g64, report):
    e_hip, e_ref = _rel(hip, g64), _rel(g32, g64)
    c_hip, c_ref = _cos(hip, g64), _cos(g32, g64)
    report[name] = (f"cos {c_hip:.6f} (oracle fp32 {c_ref:.6f})", f"err {e_hip:.1e} (oracle fp32 {e_ref:.1e})")
    assert c_hip >= min(0.9999, c_ref - 5e-5), (name, report[name])
    assert e_hip <= 2.0 * e_ref + 5e-4, (name, report[name])


def _oracle(dtype, seed=0, with_tex=False):
    cast = lambda st: {k: v.to(dtype) for k, v in st.items()}
    o = orc.OracleRenderer(cast(synth.nerf_state(ARCH[0], ARCH[1], seed, "coarse")), cast(synth.nerf_state(ARCH[2], ARCH[3], seed, "fine")),
                           cast(synth.style_state(seed)), [e.to(dtype) for e in synth.exp_sigma(seed)],
                           cast(synth.tex_encoder_state(seed)) if with_tex else None, netchunk=196608)
    return o


def test_config2_fit_step_1024_rays_gradient_parity():
    render, kw, _ = make_product(ARCH, 0, 196608, DEV)
    K = synth.intrinsics(H, H)
    rng = np.random.default_rng(21)
    n, nsub = 1024, 128
    pix = torch.from_numpy(rng.choice(H * H, n, replace=False))
    rows, cols = pix // H, pix % H
    target = torch.from_numpy(rng.uniform(0, 1, (n, 3)).astype(np.float32))
    sub = torch.from_numpy(np.sort(rng.choice(n, nsub, replace=False)))
    c2w0 = mrays.pose_spherical(25.0, 0.0, 16.0)[:3, :4].contiguous()
    bm0, tex0, exp0 = synth.codes(0)
    # ---- device: the full 1024-ray step graph ---------------------------------------------------------------------------------
    c2w = c2w0.to(DEV).requires_grad_(True)
    bm, tex, exp = [t.to(DEV).clone().requires_grad_(True) for t in (bm0, tex0, exp0)]
    light = torch.full((1,), 1.1, device=DEV, requires_grad=True)
    batch = mrays.rays_at_pixels(K, c2w, rows.to(DEV), cols.to(DEV), H, H)
    rgb, disp, acc, ex = render.render_fitting(H, H, K, chunk=n, rays=batch, shapeCodes=bm.expand(n, -1), uvCodes=tex, expType=20,
                                               expCodes=exp, verbose=True, **kw)
    assert rgb.shape == (n, 3) and ex["_z_fine"].shape == (n, 128)
    loss_sub = torch.nn.functional.l1_loss(light[0] * rgb[sub.to(DEV)], target[sub].to(DEV))
    g_hip = torch.autograd.grad(loss_sub, [bm, tex, exp, c2w, light], retain_graph=True)
    loss_full = torch.nn.functional.l1_loss(light[0] * rgb, target.to(DEV))            # run_fit.py:309
    loss_full.backward()
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(t.grad).all()) and float(t.grad.abs().sum()) > 0 for t in (bm, tex, exp, c2w, light))
    assert all(p.grad is None for p in kw["network_fine"].parameters())                 # run_fit.py never steps the networks
    zf = ex["_z_fine"].detach()[sub.to(DEV)].cpu()
    rgb_sub = rgb.detach()[sub.to(DEV)].cpu()
    del rgb, disp, acc, ex, loss_sub, loss_full, batch
    torch.cuda.empty_cache()

    # ---- oracle: the same 128 rays, the device's positions, fp32 and fp64 --------------------------------------------------------
    torch.set_num_threads(min(16, os.cpu_count() or 1))

    def oracle_grads(dtype):
        o = _oracle(dtype)
        c2 = c2w0.to(dtype).clone().requires_grad_(True)        # (clone: .to() of an fp32 tensor to fp32 is the tensor itself)
        b_, t_, e_ = [t.to(dtype).clone().requires_grad_(True) for t in (bm0, tex0, exp0)]
        li = torch.full((1,), 1.1, dtype=dtype, requires_grad=True)
        o.exp_sigma.append(e_)
        r = mrays.rays_at_pixels(K, c2, rows[sub], cols[sub])                          # CPU branch: the same formula, differentiable
        ro, rd = r[0], r[1]
        vd = rd / torch.norm(rd, dim=-1, keepdim=True)
        z = zf.to(dtype)
        raw = o.run_network(ro[:, None, :] + rd[:, None, :] * z[:, :, None], vd, o.fine, b_, t_, 20)
        rgb_o = orc.raw2outputs(raw, z, rd)[0]
        loss = torch.nn.functional.l1_loss(li[0] * rgb_o, target[sub].to(dtype))
        return [g.detach().numpy() for g in torch.autograd.grad(loss, [b_, t_, e_, c2, li])], rgb_o.detach()

    g32, rgb32 = oracle_grads(torch.float32)
    g64, _ = oracle_grads(torch.float64)
    assert float((rgb_sub - rgb32).abs().max()) <= 1e-4                                  # forward, teacher-forced: the north star's 1e-4
    report = {}
    for name, a, b, c in zip(("shape code", "texture code", "expression code", "pose c2w", "light"), g_hip, g32, g64):
        _judge(name, a.detach().cpu().numpy(), b, c, report)
    print("config 2 (1024-ray fit step), 128-ray subsample vs fp64 truth:", report)

    # ---- and the step itself, as the script runs it (Adam on codes + light) -------------------------------------------------------
    cs = [t.to(DEV).clone().requires_grad_(True) for t in (bm0, tex0, exp0)]
    light = torch.ones(1, device=DEV, requires_grad=True)
    opts = [torch.optim.Adam(cs, lr=5e-3), torch.optim.Adam([light], lr=1e-3)]
    batch = mrays.rays_at_pixels(K, c2w0.to(DEV), rows.to(DEV), cols.to(DEV), H, H)
    losses = [float(steps.fit_step(render, kw, opts, H, H, K, batch, target.to(DEV), cs[0], cs[1], cs[2], light, chunk=n)[0]) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_config4_train_step_4096_rays_gradient_parity():
    render, kw_test, _ = make_product(ARCH, 0, 196608, DEV, with_tex=True)
    render.train()
    kw = dict(kw_test, perturb=1.0)
    K = synth.intrinsics(H, H)
    rng = np.random.default_rng(41)
    n, nsub, exp_type = 4096, 128, 3
    pix = torch.from_numpy(rng.choice(H * H, n, replace=False))
    rows, cols = pix // H, pix % H
    target = torch.from_numpy(rng.uniform(0, 1, (n, 3)).astype(np.float32))
    sub = torch.from_numpy(np.sort(rng.choice(n, nsub, replace=False)))
    uv = torch.from_numpy(rng.uniform(0, 1, (512, 512, 3)).astype(np.float32))
    c2w0 = mrays.pose_spherical(-35.0, 0.0, 16.0)[:3, :4].contiguous()
    bm0 = synth.codes(0)[0]
    fine, coarse = kw["network_fine"], kw["network_fn"]
    picks = {"fine.rgb_linear.weight": fine.rgb_linear.weight,
             "fine.linear_uv_xyzBiM.linears2.Linear1.weight": fine.linear_uv_xyzBiM.linears2.Linear1.weight,
             "fine.linear_BiM_xyz.linears1.Linear0.weight": fine.linear_BiM_xyz.linears1.Linear0.weight,       # [1024, 50 + 1024]: folded + per-point columns
             "fine.alpha_linear.0.bias": fine.alpha_linear[0].bias,
             "coarse.xyzEncode.linears1.Linear0.weight": coarse.xyzEncode.linears1.Linear0.weight,             # [256, 93]: encoding + expression columns
             "coarse.linear_view_xyBMuv.0.weight": coarse.linear_view_xyBMuv[0].weight}
    # ---- device -----------------------------------------------------------------------------------------------------------------
    bm = bm0.to(DEV).clone().requires_grad_(True)
    batch = mrays.rays_at_pixels(K, c2w0.to(DEV), rows.to(DEV), cols.to(DEV), H, H)
    torch.manual_seed(7)
    rgb, disp, acc, ex = render.render(H, H, K, chunk=n, rays=batch, shapeCodes=bm.expand(n, -1), uvMap=uv.to(DEV), expType=exp_type,
                                       retraw=True, verbose=True, **kw)
    code = render.decoding_texCodes
    sd = sub.to(DEV)
    loss_sub = ((rgb[sd] - target[sub].to(DEV)) ** 2).mean() + ((ex["rgb0"][sd] - target[sub].to(DEV)) ** 2).mean()
    inputs = [bm, code, render.expCodes_Sigma[exp_type]] + list(picks.values())
    g_hip = [g.detach().cpu().numpy() for g in torch.autograd.grad(loss_sub, inputs, retain_graph=False)]
    torch.cuda.synchronize()
    zc, zf = ex["_z_coarse"].detach()[sd].cpu(), ex["_z_fine"].detach()[sd].cpu()
    code_cpu = code.detach().cpu()
    rgb_sub, rgb0_sub = rgb.detach()[sd].cpu(), ex["rgb0"].detach()[sd].cpu()
    assert zc.shape == (nsub, 64) and bool((zc[:, 1:] >= zc[:, :-1]).all()) and not torch.equal(zc[0], zc[1])   # stratified jitter per ray
    del rgb, disp, acc, ex, loss_sub, batch, code
    torch.cuda.empty_cache()

    # ---- oracle -------------------------------------------------------------------------------------------------------------------
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ro, rd = orc.get_rays(H, H, K, c2w0)
    ro, rd = ro[rows[sub], cols[sub]], rd[rows[sub], cols[sub]]

    def oracle_grads(dtype):
        o = _oracle(dtype, with_tex=True)
        leaves = {}
        for name in picks:
            net, key = name.split(".", 1)
            st = o.fine if net == "fine" else o.coarse
            st[key] = st[key].clone().requires_grad_(True)
            leaves[name] = st[key]
        b_ = bm0.to(dtype).clone().requires_grad_(True)
        t_ = code_cpu.to(dtype).clone().requires_grad_(True)
        o.exp_sigma[exp_type] = o.exp_sigma[exp_type].clone().requires_grad_(True)
        r_o, r_d = ro.to(dtype), rd.to(dtype)
        vd = r_d / torch.norm(r_d, dim=-1, keepdim=True)
        outs = []
        for z, st in ((zc.to(dtype), o.coarse), (zf.to(dtype), o.fine)):
            raw = o.run_network(r_o[:, None, :] + r_d[:, None, :] * z[:, :, None], vd, st, b_, t_, exp_type)
            outs.append(orc.raw2outputs(raw, z, r_d)[0])
        tgt = target[sub].to(dtype)
        loss = ((outs[1] - tgt) ** 2).mean() + ((outs[0] - tgt) ** 2).mean()
        gs = torch.autograd.grad(loss, [b_, t_, o.exp_sigma[exp_type]] + [leaves[k] for k in picks])
        return [g.detach().numpy() for g in gs], outs[1].detach(), outs[0].detach()

    g32, rgb32, rgb032 = oracle_grads(torch.float32)
    g64, _, _ = oracle_grads(torch.float64)
    assert float((rgb_sub - rgb32).abs().max()) <= 1e-4 and float((rgb0_sub - rgb032).abs().max()) <= 1e-4
    report = {}
    for name, a, b, c in zip(["shape code", "texture code (encoder output)", f"expression sigma[{exp_type}]"] + list(picks), g_hip, g32, g64):
        _judge(name, a, b, c, report)
    print("config 4 (4096-ray training graph), 128-ray subsample vs fp64 truth:", report)

    # ---- and the step itself: bucket, all parameter groups, Adam ---------------------------------------------------------------------
    params = list(coarse.parameters()) + list(fine.parameters()) + list(render.grad_parameter())
    opt = torch.optim.Adam(params, lr=5e-5)
    bucket = mdist.GradBucket(params)
    batch = mrays.rays_at_pixels(K, c2w0.to(DEV), rows.to(DEV), cols.to(DEV), H, H)
    before = fine.linear_uv_xyzBiM.linears2.Linear1.weight.detach().clone()
    losses = [float(steps.train_step(render, kw, opt, bucket, H, H, K, batch, target.to(DEV), bm0.to(DEV).expand(n, -1), uv.to(DEV), exp_type,
                                     chunk=n)) for _ in range(2)]
    assert all(np.isfinite(losses)) and bucket.numel > 32_000_000 and float(bucket.flat.abs().sum()) > 0
    assert not torch.equal(before, fine.linear_uv_xyzBiM.linears2.Linear1.weight.detach())
