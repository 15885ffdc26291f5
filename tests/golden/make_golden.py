#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself on CPU.

Build-container only: it imports ``/root/reference`` (which never travels to the GPU box) with the
three shims of SURVEY.md Appendix A (stub ``imageio``/``cv2``; ``Tensor.cuda`` = identity; no
``config_parser``).  The fixtures are data — inputs and the reference's outputs — and are what pins
``oracle/mofa_oracle.py``.  Large weights are NOT stored: they are regenerated on both sides from the
seeded recipe in ``mofanerf_amd/synth.py``.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
for _m in ("imageio", "cv2"):
    sys.modules[_m] = types.ModuleType(_m)
torch.Tensor.cuda = lambda self, *a, **k: self
_default_tensor = torch.Tensor

from models import render_class  # noqa: E402  (reference)
from models.model import NeRF, StyleModule, get_embedder  # noqa: E402
from tools.run_nerf_helpers import get_rays, sample_pdf  # noqa: E402
from tools.load_facescape import pose_spherical  # noqa: E402

torch.autograd.set_detect_anomaly(False)

from mofanerf_amd import synth  # noqa: E402

torch.manual_seed(0)
np.random.seed(0)
torch.set_num_threads(8)


def npd(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, d):
    p = os.path.join(HERE, name)
    np.savez_compressed(p, **npd(d))
    print(f"wrote {name}: {os.path.getsize(p) / 1024:.1f} KiB, {len(d)} arrays")


def mk_nerf(D, W, seed=0, tag="nerf"):
    m = NeRF(D=D, W=W, input_ch_shapeCodes=50, input_ch_textureCodes=256, input_ch=63 + 30, output_ch=5,
             skips=[4], input_ch_views=27, use_viewdirs=True)
    m.load_state_dict(synth.nerf_state(D, W, seed, tag))
    return m.eval()


def mk_renderer(netchunk, seed=0, with_tex=False):
    embed_fn, _ = get_embedder(10, 0)
    embeddirs_fn, _ = get_embedder(4, 0)
    r = render_class.myRenderer(embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=netchunk, uvCodesLen=256,
                                expCodesLen=30)
    r.idSpecificMod.load_state_dict(synth.style_state(seed))
    if with_tex:
        r.texEncoder.load_state_dict(synth.tex_encoder_state(seed))
    for dst, src in zip(r.expCodes_Sigma, synth.exp_sigma(seed)):
        dst.data[:] = src
    return r.eval()


def kwargs_for(r, coarse, fine, perturb=0.0, raw_noise_std=0.0, white_bkgd=False, N_samples=64, N_importance=64):
    return dict(network_query_fn=r.run_network, perturb=perturb, N_importance=N_importance, network_fine=fine,
                N_samples=N_samples, network_fn=coarse, use_viewdirs=True, white_bkgd=white_bkgd,
                raw_noise_std=raw_noise_std, ndc=False, lindisp=False, near=8.0, far=26.0)


class Recorder:
    """Record the arguments/results of raw2outputs and sample_pdf as render_rays calls them."""

    def __init__(self):
        self.r2o, self.spdf = [], []
        self._r2o, self._spdf = render_class.raw2outputs, render_class.sample_pdf

    def __enter__(self):
        def r2o(raw, z, d, *a, **k):
            out = self._r2o(raw, z, d, *a, **k)
            self.r2o.append(dict(raw=raw.detach().clone(), z=z.detach().clone(), weights=out[3].detach().clone()))
            return out

        def spdf(bins, w, n, **k):
            out = self._spdf(bins, w, n, **k)
            self.spdf.append(dict(samples=out.detach().clone()))
            return out

        render_class.raw2outputs, render_class.sample_pdf = r2o, spdf
        return self

    def __exit__(self, *a):
        render_class.raw2outputs, render_class.sample_pdf = self._r2o, self._spdf


# ------------------------------------------------------------------------------------------------
def g1_kats():
    out = {}
    rng = np.random.default_rng(1)
    # Embedder, incl. |x| up to 12 (arguments up to 6e3 rad at 2^9)
    x = torch.from_numpy(np.concatenate([rng.uniform(-12, 12, (300, 3)), [[0.5, -1.25, 2.0]], [[0, 0, 0]],
                                         [[11.999, -11.999, 7.25]]]).astype(np.float32))
    out["embed_x"] = x
    out["embed_L10"] = get_embedder(10, 0)[0](x)
    out["embed_L4"] = get_embedder(4, 0)[0](x)
    # raw2outputs
    for S in (64, 128):
        R = 48
        raw = torch.from_numpy(rng.normal(0, 1.5, (R, S, 4)).astype(np.float32))
        raw[0, :, 3] = -1.0                  # zero-opacity ray -> acc 0 -> NaN disp
        raw[1, :, 3] = 50.0                  # saturated from the first sample
        z = torch.sort(torch.from_numpy(rng.uniform(8, 26, (R, S)).astype(np.float32)), -1)[0]
        z[2] = torch.linspace(8, 26, S)
        d = torch.from_numpy(rng.normal(0, 1, (R, 3)).astype(np.float32))
        for wb in (False, True):
            o = render_class.raw2outputs(raw, z, d, 0, wb)
            for n, v in zip(("rgb", "disp", "acc", "weights", "depth"), o):
                out[f"r2o{S}_{int(wb)}_{n}"] = v
        out[f"r2o{S}_raw"], out[f"r2o{S}_z"], out[f"r2o{S}_d"] = raw, z, d
        o = render_class.raw2outputs(raw, z, d, 0.7, False, pytest=True)     # noise = seed-0 np.random.rand * std
        for n, v in zip(("rgb", "disp", "acc", "weights", "depth"), o):
            out[f"r2o{S}_noise_{n}"] = v
    o = render_class.raw2outputs(torch.tensor([[[0.1, -0.2, 0.3, 0.5], [1, 0, -1, 2], [0, 0.5, 0.25, -1],
                                                [-0.5, 0.2, 0.1, 0.7]]]), torch.tensor([[8., 14, 20, 26]]),
                                 2 * torch.tensor([[0, 0.6, -0.8]]))
    out["r2o_anchor_rgb"], out["r2o_anchor_disp"], out["r2o_anchor_weights"], out["r2o_anchor_depth"] = o[0], o[1], o[3], o[4]
    # sample_pdf
    R = 40
    bins = torch.sort(torch.from_numpy(rng.uniform(8, 26, (R, 63)).astype(np.float32)), -1)[0]
    w = torch.from_numpy((rng.uniform(0, 1, (R, 62)) ** 6).astype(np.float32))
    w[0] = 0.0                                # all-zero weights: uniform pdf
    w[1] = 0.0; w[1, 30] = 1.0                # one spike: the denom<1e-5 branch everywhere else
    w[2, :31] = 0.0                           # leading empty bins
    out["spdf_bins"], out["spdf_w"] = bins, w
    out["spdf_det"] = sample_pdf(bins, w, 64, det=True)
    out["spdf_rand"] = sample_pdf(bins, w, 64, det=False, pytest=True)   # u = seed-0 np.random.rand(R,64)
    out["spdf_anchor"] = sample_pdf(torch.linspace(8, 26, 7)[None], torch.tensor([[0, .1, .6, .2, .05, 0]]), 8, det=True)
    # get_rays
    for ang in (-60.0, 0.0, 60.0):
        K = np.array([[600., 0, 128], [0, 600., 128], [0, 0, 1]])
        c2w = pose_spherical(ang, 0.0, 16.0)
        ro, rd = get_rays(256, 256, K, c2w[:3, :4])
        tag = f"rays{int(ang)}"
        out[tag + "_c2w"], out[tag + "_o"], out[tag + "_d_sub"] = c2w, ro[0, 0], rd[::37, ::41]
    out["pose_m17_23_16"] = pose_spherical(-17.0, 23.0, 16.0)
    # NeRF.forward at tiny widths; StyleModule
    for D, W in ((8, 64), (10, 64), (8, 96)):
        net = mk_nerf(D, W)
        n = 200
        a = torch.from_numpy(rng.normal(0, 1, (n, 93)).astype(np.float32))
        b = torch.from_numpy(rng.normal(0, 0.05, (1, 50)).astype(np.float32)).expand(n, -1)
        c = torch.from_numpy(rng.normal(0, 1, (n, 27)).astype(np.float32))
        t = torch.from_numpy(rng.normal(0.2, 0.3, (1, 256)).astype(np.float32)).expand(n, -1)
        with torch.no_grad():
            out[f"nerf{D}x{W}_out"] = net(a, b, c, t)
        out[f"nerf{D}x{W}_pts"], out[f"nerf{D}x{W}_bm"], out[f"nerf{D}x{W}_views"], out[f"nerf{D}x{W}_tex"] = a, b[:1], c, t[:1]
    sm = StyleModule()
    sm.load_state_dict(synth.style_state(0))
    bm = synth.codes(0)[0]
    with torch.no_grad():
        s, b = sm(bm)
    out["style_bm"], out["style_scale"], out["style_bias"] = bm, s, b
    save("kat.npz", out)


def _case_setup(H, K, angle, Dc, Wc, Df, Wf, chunk, netchunk, perturb=0.0, noise=0.0, white=False, pytest=False, seed=0,
                N_samples=64, N_importance=64):
    r = mk_renderer(netchunk, seed)
    coarse, fine = mk_nerf(Dc, Wc, seed, "coarse"), mk_nerf(Df, Wf, seed, "fine")
    kw = kwargs_for(r, coarse, fine, perturb, noise, white, N_samples, N_importance)
    if pytest:
        kw["pytest"] = True
    bm, tex, exp = synth.codes(seed)
    c2w = pose_spherical(angle, 0.0, 16.0)[:3, :4]
    meta = dict(c2w=c2w, K=K, bm=bm, tex=tex, exp=exp, H=H, chunk=chunk, netchunk=netchunk,
                arch=np.array([Dc, Wc, Df, Wf]), seed=seed, perturb=perturb, noise=noise, white=int(white))
    if (N_samples, N_importance) != (64, 64):        # (the 64 + 64 fixtures keep their round-1 key set: they regenerate byte for byte)
        meta.update(N_samples=N_samples, N_importance=N_importance)
    call = lambda: r.render_fitting(H, H, K, chunk=chunk, c2w=c2w, shapeCodes=bm, uvCodes=tex, expType=20, expCodes=exp,
                                    retraw=True, **kw)
    return call, meta


def _render_case(name, H, K, angle, Dc, Wc, Df, Wf, chunk, netchunk, perturb=0.0, noise=0.0, white=False,
                 pytest=False, intermediates=True, seed=0, N_samples=64, N_importance=64):
    call, out = _case_setup(H, K, angle, Dc, Wc, Df, Wf, chunk, netchunk, perturb, noise, white, pytest, seed, N_samples, N_importance)
    with torch.no_grad(), Recorder() as rec:
        rgb, disp, acc, ex = call()
    out.update(rgb=rgb, disp=disp, acc=acc, rgb0=ex["rgb0"], disp0=ex["disp0"], acc0=ex["acc0"], z_std=ex["z_std"])
    if intermediates:
        nch = len(rec.spdf)
        out["z_coarse"] = torch.cat([rec.r2o[2 * i]["z"] for i in range(nch)])
        out["raw_coarse"] = torch.cat([rec.r2o[2 * i]["raw"] for i in range(nch)])
        out["weights_coarse"] = torch.cat([rec.r2o[2 * i]["weights"] for i in range(nch)])
        out["z_fine"] = torch.cat([rec.r2o[2 * i + 1]["z"] for i in range(nch)])
        out["raw_fine"] = torch.cat([rec.r2o[2 * i + 1]["raw"] for i in range(nch)])
        out["weights_fine"] = torch.cat([rec.r2o[2 * i + 1]["weights"] for i in range(nch)])
        out["z_samples"] = torch.cat([s["samples"] for s in rec.spdf])
    save(name, out)


def _envelope(call, n_pert, eps, verbose=True):
    """The reference's outputs under ``n_pert`` seeded relative perturbations ``w * (1 + U(-eps, eps))`` of its OWN coarse
    weights (class PerturbCoarse), next to its unperturbed sample positions: the per-ray envelope a second correct fp32
    implementation must land in.  ``moved``: samples per ray that move by more than 1e-3; ``agree``: every new sample of the
    ray stays within a few ulp (6e-6) of the unperturbed one."""
    import time
    with torch.no_grad(), Recorder() as rec:
        rgb, disp, acc, ex = call()
    zs0 = torch.cat([s["samples"] for s in rec.spdf])
    P = {k: [] for k in ("rgb", "acc", "disp", "z_std", "moved", "agree")}
    for k in range(n_pert):
        t0 = time.time()
        with torch.no_grad(), PerturbCoarse(1000 + k, eps) as pc:
            rgb_k, disp_k, acc_k, ex_k = call()
        zs = torch.cat(pc.spdf)
        P["rgb"].append(rgb_k.reshape(-1, 3)); P["acc"].append(acc_k.reshape(-1)); P["disp"].append(disp_k.reshape(-1))
        P["z_std"].append(ex_k["z_std"].reshape(-1))
        P["moved"].append(((zs - zs0).abs() > 1e-3).sum(-1).to(torch.uint8))
        P["agree"].append(((zs - zs0).abs() <= 6e-6).all(-1))
        if verbose:
            d = (rgb_k - rgb).abs().reshape(-1, 3).max(-1)[0]
            print(f"  perturbation {k}: {time.time() - t0:.1f} s; rays agreeing within ulps {P['agree'][-1].float().mean():.3f}; with a "
                  f"moved sample {(P['moved'][-1] > 0).float().mean():.3f}; rgb diff max {d.max():.2e} mean {d.mean():.2e}; "
                  f"rays > 1e-4: {(d > 1e-4).float().mean():.3f}", flush=True)
    out = {"pert_" + k: torch.stack(v, 0) for k, v in P.items()}
    out["pert_eps"] = eps
    return (rgb, disp, acc, ex, rec), out


def g2_small():
    K16 = np.array([[37.5, 0, 8.0], [0, 37.5, 8.0], [0, 0, 1]])       # 16x16 image, focal 1200/32
    _render_case("e2e_small.npz", 16, K16, 0.0, 8, 64, 10, 128, chunk=96, netchunk=4096)
    _render_case("e2e_small_stoch.npz", 16, K16, -60.0, 8, 64, 10, 64, chunk=256, netchunk=100000, perturb=1.0,
                 noise=0.5, white=True, pytest=True)


def g3_true():
    K8 = np.array([[18.75, 0, 4.0], [0, 18.75, 4.0], [0, 0, 1]])       # 8x8 image = 64 rays
    _render_case("e2e_true.npz", 8, K8, 60.0, 8, 256, 10, 1024, chunk=64, netchunk=196608, intermediates=True)


def g4_grads():
    """d loss / d{bm, tex, exp, rays_o, rays_d} for loss = mean|rgb - 0.5| + mean(rgb0^2), 64 rays."""
    r = mk_renderer(4096, 0)
    coarse, fine = mk_nerf(8, 64, 0, "coarse"), mk_nerf(10, 64, 0, "fine")
    kw = kwargs_for(r, coarse, fine)
    bm, tex, exp = [t.clone().requires_grad_(True) for t in synth.codes(0)]
    K8 = np.array([[18.75, 0, 4.0], [0, 18.75, 4.0], [0, 0, 1]])
    ro, rd = get_rays(8, 8, K8, pose_spherical(20.0, 0.0, 16.0)[:3, :4])
    ro = ro.reshape(-1, 3).clone().requires_grad_(True)
    rd = rd.reshape(-1, 3).clone().requires_grad_(True)
    rgb, disp, acc, ex = r.render_fitting(8, 8, K8, chunk=64, rays=torch.stack([ro, rd], 0), shapeCodes=bm.expand(64, 50),
                                          uvCodes=tex, expType=20, expCodes=exp, **kw)
    loss = (rgb - 0.5).abs().mean() + (ex["rgb0"] ** 2).mean()
    loss.backward()
    save("grads_small.npz", dict(bm=bm, tex=tex, exp=exp, rays_o=ro, rays_d=rd, rgb=rgb, rgb0=ex["rgb0"], loss=loss,
                                 g_bm=bm.grad, g_tex=tex.grad, g_exp=exp.grad, g_rays_o=ro.grad, g_rays_d=rd.grad,
                                 g_w_rgb=fine.rgb_linear.weight.grad, g_b_alpha=fine.alpha_linear[0].bias.grad,
                                 g_w_xyz0_c=coarse.xyzEncode.linears1.Linear0.weight.grad,
                                 g_style_scale_w=r.idSpecificMod.linears_scale.weight.grad))


def g5_render_tex():
    """``render()``: texture encoder on a seeded UV map, then the same path (training-style entry)."""
    r = mk_renderer(4096, 0, with_tex=True)
    coarse, fine = mk_nerf(8, 64, 0, "coarse"), mk_nerf(10, 64, 0, "fine")
    kw = kwargs_for(r, coarse, fine)
    rng = np.random.default_rng(5)
    uv = torch.from_numpy(rng.uniform(0, 1, (512, 512, 3)).astype(np.float32))
    bm = synth.codes(0)[0]
    K8 = np.array([[18.75, 0, 4.0], [0, 18.75, 4.0], [0, 0, 1]])
    ro, rd = get_rays(8, 8, K8, pose_spherical(-35.0, 0.0, 16.0)[:3, :4])
    rays = torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0)
    with torch.no_grad():
        rgb, disp, acc, ex = r.render(8, 8, K8, chunk=64, rays=rays, shapeCodes=bm.expand(64, 50), uvMap=uv, expType=7,
                                      retraw=True, **kw)
        code = r.decoding_texCodes
    # the UV map is regenerated from rng seed 5 by the test; store only a checksum of it
    save("render_tex.npz", dict(uv_sum=uv.double().sum(), bm=bm, rays=rays, tex_code=code, rgb=rgb, disp=disp, acc=acc,
                                rgb0=ex["rgb0"], z_std=ex["z_std"], raw=ex["raw"], losses=ex["losses"]))


def g6_schema():
    """Checkpoint schema as the reference's own modules report it: state-dict key -> shape for both shipped networks, the
    StyleModule and the texture encoder, the optimizer's param-group layout over ``grad_vars`` (create_model_condition.py:
    25-63) and the top-level keys run_train.py:371-379 writes.  Shapes only - no weights."""
    import json
    r = mk_renderer(196608, 0)
    coarse = NeRF(D=8, W=256, input_ch_shapeCodes=50, input_ch_textureCodes=256, input_ch=93, output_ch=5, skips=[4],
                  input_ch_views=27, use_viewdirs=True)
    fine = NeRF(D=10, W=1024, input_ch_shapeCodes=50, input_ch_textureCodes=256, input_ch=93, output_ch=5, skips=[4],
                input_ch_views=27, use_viewdirs=True)
    grad_vars = list(coarse.parameters()) + list(fine.parameters()) + list(r.grad_parameter())
    opt = torch.optim.Adam(params=grad_vars, lr=5e-5, betas=(0.9, 0.999))
    sd = lambda m: {k: list(v.shape) for k, v in m.state_dict().items()}
    out = {
        "network_fn_state_dict": sd(coarse), "network_fine_state_dict": sd(fine),
        "network_render_textureEncoder": sd(r.texEncoder), "network_render_idSpecific": sd(r.idSpecificMod),
        "expression_latent_codes_sigma": [list(t.shape) for t in r.expCodes_Sigma],
        "grad_vars_shapes": [list(p.shape) for p in grad_vars],
        "optimizer_param_groups": [{k: (len(v) if k == "params" else v) for k, v in g.items()
                                    if k in ("params", "lr", "betas", "eps", "weight_decay", "amsgrad")}
                                   for g in opt.state_dict()["param_groups"]],
        "top_level_keys": ["global_step", "network_fn_state_dict", "network_fine_state_dict",
                           "network_render_textureEncoder", "network_render_idSpecific", "optimizer_state_dict",
                           "expression_latent_codes_sigma"],
        "n_params": {"coarse": sum(p.numel() for p in coarse.parameters()), "fine": sum(p.numel() for p in fine.parameters()),
                     "texEncoder": sum(p.numel() for p in r.texEncoder.parameters()),
                     "idSpecific": sum(p.numel() for p in r.idSpecificMod.parameters())},
    }
    p = os.path.join(HERE, "schema.json")
    with open(p, "w") as f:
        json.dump(out, f, indent=0)
    print(f"wrote schema.json: {os.path.getsize(p) / 1024:.1f} KiB")


class PerturbCoarse:
    """Perturb the COARSE compositing weights the way a second correct fp32 implementation would (relative noise of a
    few ulp: ``w * (1 + U(-eps, eps))``) before the reference resamples from them, and record the new sample positions.
    The reference's outputs under this perturbation are its OWN envelope: how far a pixel legitimately moves when the
    importance resampling (run_nerf_helpers.py:203-247, the ``denom < 1e-5`` branch at :243) sees weights that differ
    only by rounding."""

    def __init__(self, seed, eps):
        self.rng, self.eps, self.n, self.spdf = np.random.default_rng(seed), eps, 0, []
        self._r2o, self._spdf = render_class.raw2outputs, render_class.sample_pdf

    def __enter__(self):
        def r2o(raw, z, d, *a, **k):
            out = list(self._r2o(raw, z, d, *a, **k))
            if self.n % 2 == 0:
                w = out[3]
                out[3] = w * torch.from_numpy(1.0 + self.rng.uniform(-self.eps, self.eps, tuple(w.shape))).float()
            self.n += 1
            return tuple(out)

        def spdf(bins, w, n, **k):
            out = self._spdf(bins, w, n, **k)
            self.spdf.append(out.detach().clone())
            return out

        render_class.raw2outputs, render_class.sample_pdf = r2o, spdf
        return self

    def __exit__(self, *a):
        render_class.raw2outputs, render_class.sample_pdf = self._r2o, self._spdf


def g7_config1(n_pert=4, eps=1e-6):
    """BASELINE config 1: the 64x64 ``--renderType rendering`` frame (run_fit.py:357-362 geometry scaled: focal 150,
    centre 32), chunk 4096, SHIPPED network sizes (coarse 256x8, fine 1024x10) — 4,096 rays of the reference itself,
    with the intermediates a teacher-forced test needs, plus the reference's outputs under ``n_pert`` seeded ulp-level
    perturbations of its own coarse weights (the per-ray envelope)."""
    H, seed = 64, 0
    K = synth.intrinsics(H, H)
    arch = (8, 256, 10, 1024)
    r = mk_renderer(196608, seed)
    coarse, fine = mk_nerf(arch[0], arch[1], seed, "coarse"), mk_nerf(arch[2], arch[3], seed, "fine")
    kw = kwargs_for(r, coarse, fine)
    bm, tex, exp = synth.codes(seed)
    c2w = pose_spherical(0.0, 0.0, 16.0)[:3, :4]
    call = lambda: r.render_fitting(H, H, K, chunk=4096, c2w=c2w, shapeCodes=bm, uvCodes=tex, expType=20, expCodes=exp,
                                    retraw=True, **kw)
    out = dict(c2w=c2w, K=K, bm=bm, tex=tex, exp=exp, H=H, chunk=4096, netchunk=196608, arch=np.array(arch), seed=seed,
               perturb=0.0, noise=0.0, white=0)
    (rgb, disp, acc, ex, rec), env = _envelope(call, n_pert, eps)
    assert len(rec.spdf) == 1
    sub = np.arange(0, H * H, 16)                                    # 256 rays keep the bulky per-sample arrays
    out.update(rgb=rgb, disp=disp, acc=acc, rgb0=ex["rgb0"], disp0=ex["disp0"], acc0=ex["acc0"], z_std=ex["z_std"],
               z_coarse_row=rec.r2o[0]["z"][:1], weights_coarse=rec.r2o[0]["weights"], z_samples=rec.spdf[0]["samples"],
               sub=sub, raw_coarse_sub=rec.r2o[0]["raw"][sub], raw_fine_sub=rec.r2o[1]["raw"][sub],
               weights_fine_sub=rec.r2o[1]["weights"][sub])
    assert torch.equal(rec.r2o[0]["z"], rec.r2o[0]["z"][:1].expand(H * H, -1))
    assert torch.equal(rec.r2o[1]["z"], torch.sort(torch.cat([rec.r2o[0]["z"], rec.spdf[0]["samples"]], -1), -1)[0])
    out.update(env)
    save("e2e_c1.npz", out)


def g11_envelopes(n_pert=8, eps=1e-6):
    """Per-ray envelopes (see _envelope) for the small end-to-end fixtures g2 / g3, in separate files so that the original
    fixtures stay byte-identical."""
    K16 = np.array([[37.5, 0, 8.0], [0, 37.5, 8.0], [0, 0, 1]])
    K8 = np.array([[18.75, 0, 4.0], [0, 18.75, 4.0], [0, 0, 1]])
    for name, args, kws in (("e2e_small", (16, K16, 0.0, 8, 64, 10, 128, 96, 4096), {}),
                            ("e2e_small_stoch", (16, K16, -60.0, 8, 64, 10, 64, 256, 100000),
                             dict(perturb=1.0, noise=0.5, white=True, pytest=True)),
                            ("e2e_true", (8, K8, 60.0, 8, 256, 10, 1024, 64, 196608), {})):
        call, _ = _case_setup(*args, **kws)
        (rgb, disp, acc, ex, rec), env = _envelope(call, n_pert, eps, verbose=False)
        base = np.load(os.path.join(HERE, name + ".npz"))
        assert np.array_equal(base["rgb"], rgb.numpy()), name        # same run as the committed fixture
        save(name + "_env.npz", env)


def g12_pose_grads():
    """run_fit.py's pose path: rays from a camera pose that REQUIRES GRAD (get_rays on a tensor c2w = get_rays_withGrad,
    run_fit.py:116-127), N_rand of them gathered at sampled pixels (:281-293), render_fitting, L1 loss, backward to the pose.
    Stores the pixel list, the per-ray gradients and d loss / d c2w."""
    r = mk_renderer(4096, 0)
    coarse, fine = mk_nerf(8, 64, 0, "coarse"), mk_nerf(10, 64, 0, "fine")
    kw = kwargs_for(r, coarse, fine)
    bm, tex, exp = synth.codes(0)
    H = 32
    K = synth.intrinsics(H, H)
    c2w = pose_spherical(25.0, 0.0, 16.0)[:3, :4].clone().requires_grad_(True)
    ro, rd = get_rays(H, H, torch.from_numpy(K).float(), c2w)
    ro.retain_grad(); rd.retain_grad()
    rng = np.random.default_rng(12)
    pix = np.sort(rng.choice(H * H, 96, replace=False))
    rows, cols = torch.from_numpy(pix // H), torch.from_numpy(pix % H)
    rays_o, rays_d = ro[rows, cols], rd[rows, cols]
    rays_o.retain_grad(); rays_d.retain_grad()
    target = torch.from_numpy(rng.uniform(0, 1, (96, 3)).astype(np.float32))
    rgb, disp, acc, ex = r.render_fitting(H, H, K, chunk=96, rays=torch.stack([rays_o, rays_d], 0), shapeCodes=bm.expand(96, 50),
                                          uvCodes=tex, expType=20, expCodes=exp, **kw)
    loss = torch.nn.functional.l1_loss(rgb, target) + (ex["rgb0"] ** 2).mean()
    loss.backward()
    save("grads_pose.npz", dict(c2w=c2w, K=K, H=H, pix=pix.astype(np.int32), target=target, bm=bm, tex=tex, exp=exp, rays_o=rays_o,
                                rays_d=rays_d, rgb=rgb, loss=loss, g_rays_o=rays_o.grad, g_rays_d=rays_d.grad, g_c2w=c2w.grad))


def g13_ndc():
    """ndc_rays (tools/run_nerf_helpers.py:182-200) on a 16x16 view, as render(..., ndc=True) calls it (near = 1)."""
    from tools.run_nerf_helpers import ndc_rays
    K = synth.intrinsics(16, 16)
    c2w = pose_spherical(10.0, -15.0, 4.0)[:3, :4]
    ro, rd = get_rays(16, 16, K, c2w)
    no, nd = ndc_rays(16, 16, K[0][0], 1., ro, rd)
    save("kat_ndc.npz", dict(K=K, c2w=c2w, rays_o=ro, rays_d=rd, ndc_o=no, ndc_d=nd))


def _sampled(t, key, n=256):
    """``n`` entries of a gradient tensor at seeded positions (the 27.5 M fine-network weight gradients are not stored
    whole) + its L2 norm."""
    flat = t.detach().reshape(-1)
    idx = np.random.default_rng(zlib_crc(key)).integers(0, flat.numel(), size=min(n, flat.numel()))
    return flat[torch.from_numpy(idx)], flat.double().norm().float()


def zlib_crc(key):
    import zlib
    return zlib.crc32(key.encode())


def g8_true_grads():
    """Forward + backward through the reference's ``run_network`` (render_class.py:69-94) at the SHIPPED sizes with
    explicit sample positions: raw = run_network(o + d z, d/|d|, net); loss = sum(raw * G).  Gradients w.r.t. rays,
    codes, every weight and bias of the network (sampled entries + norms) and the StyleModule.  This is what pins the
    width-1024 backward kernels (K = 2048 skip layers included) and the 64-bit tape offsets are pinned by size tests."""
    out = {}
    for tag, (D, W), R, S in (("fine", (10, 1024), 40, 48), ("coarse", (8, 256), 64, 64)):
        rng = np.random.default_rng(D * W)
        r = mk_renderer(196608, 0)
        net = mk_nerf(D, W, 0, tag).train()
        bm, tex, exp = [t.clone().requires_grad_(True) for t in synth.codes(0)]
        o = torch.from_numpy(rng.uniform(-2, 2, (R, 3)).astype(np.float32)).requires_grad_(True)
        d = torch.from_numpy(rng.normal(0, 0.3, (R, 3)).astype(np.float32)).requires_grad_(True)
        z = torch.from_numpy(np.sort(rng.uniform(8, 26, (R, S)).astype(np.float32), -1))
        G = torch.from_numpy(rng.normal(size=(R, S, 4)).astype(np.float32))
        r.shapeCodes, r.expType, r.decoding_texCodes = bm.expand(R, 50), 20, tex
        r.expCodes_Sigma.append(exp)
        pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
        vd = d / torch.norm(d, dim=-1, keepdim=True)
        raw = r.run_network(pts, vd, net)
        (raw * G).sum().backward()
        out.update({f"{tag}_{k}": v for k, v in dict(o=o, d=d, z=z, G=G, bm=bm, tex=tex, exp=exp, raw=raw, g_o=o.grad,
                                                      g_d=d.grad, g_bm=bm.grad, g_tex=tex.grad, g_exp=exp.grad,
                                                      arch=np.array([D, W])).items()})
        for key, p in list(net.named_parameters()) + [("style." + k, v) for k, v in r.idSpecificMod.named_parameters()]:
            out[f"{tag}_gs/{key}"], out[f"{tag}_gn/{key}"] = _sampled(p.grad, f"{tag}/{key}")
        # The same modules and the same fp32 point set in DOUBLE: the yardstick.  The reference's fp32 gradients are themselves
        # 1e-3 .. 1e-2 away from it (ReLU masks of units whose pre-activation is within fp32 noise of 0 flip, and every first-layer
        # gradient passes through d/dx sin(2^9 x)), so an fp32 implementation is judged by ITS distance to this truth relative
        # to the reference's own distance — not by its distance to another fp32 result.
        r64 = mk_renderer(196608, 0)
        r64.idSpecificMod.double()
        net64 = mk_nerf(D, W, 0, tag).double().train()
        bm64, tex64, exp64 = [t.detach().double().requires_grad_(True) for t in (bm, tex, exp)]
        o64, d64 = o.detach().clone().requires_grad_(True), d.detach().clone().requires_grad_(True)     # fp32 leaves
        r64.shapeCodes, r64.expType, r64.decoding_texCodes = bm64.expand(R, 50), 20, tex64
        r64.expCodes_Sigma = [e.detach().double() for e in r64.expCodes_Sigma] + [exp64]
        pts64 = (o64[:, None, :] + d64[:, None, :] * z[:, :, None]).double()
        vd64 = (d64 / torch.norm(d64, dim=-1, keepdim=True)).double()
        raw64 = r64.run_network(pts64, vd64, net64)
        (raw64 * G.double()).sum().backward()
        out.update({f"{tag}_t_{k}": v.detach().double() for k, v in dict(raw=raw64, g_o=o64.grad, g_d=d64.grad, g_bm=bm64.grad,
                                                                           g_tex=tex64.grad, g_exp=exp64.grad).items()})
        for key, p in list(net64.named_parameters()) + [("style." + k, v) for k, v in r64.idSpecificMod.named_parameters()]:
            s64, n64 = _sampled(p.grad, f"{tag}/{key}")
            out[f"{tag}_ts/{key}"], out[f"{tag}_tn/{key}"] = s64.double(), p.grad.double().norm()
    save("grads_true.npz", out)


def g9_run_network_kat():
    """``run_network`` (render_class.py:69-94: Embedder, expression modulation, code expansion, batchify, NeRF.forward)
    from EXPLICIT points / view directions / codes at small widths — the input form the fused HIP network takes."""
    out = {}
    rng = np.random.default_rng(9)
    for D, W, netchunk in ((8, 64, 1000), (10, 64, 4096), (8, 96, 777), (10, 128, 100000)):
        r = mk_renderer(netchunk, 0)
        net = mk_nerf(D, W, 3, "kat")
        R, S = 23, 37
        pts = torch.from_numpy(rng.uniform(-9, 9, (R, S, 3)).astype(np.float32))
        vd = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(R, 3)).astype(np.float32)), dim=-1)
        bm, tex, _ = synth.codes(D + W)
        r.shapeCodes, r.expType, r.decoding_texCodes = bm.expand(R, 50), 5, tex
        with torch.no_grad():
            raw = r.run_network(pts, vd, net)
        t = f"rn{D}x{W}"
        out.update({t + "_pts": pts, t + "_vd": vd, t + "_bm": bm, t + "_tex": tex, t + "_raw": raw,
                    t + "_meta": np.array([D, W, netchunk, 3, 5])})     # D, W, netchunk, weight seed, expType
    save("kat_run_network.npz", out)


def g10_checkpoint():
    """A checkpoint WRITTEN BY THE REFERENCE'S OWN MODULES in run_train.py:369-379's format (small nets 8x64 + 10x64 so the
    file stays small; the texture encoder, whose size is fixed at 3.3 M parameters, gets weights on a 5-level grid (17 for its small tensors) so the
    archive compresses) and the reference's renders from it: ``render_fitting`` (codes given) and ``render`` (texture
    encoder on a seeded UV map).  The GPU test reloads it through ``factory.create_nerf``'s reload path
    (create_model_condition.py:72-89)."""
    import gzip, io
    r = mk_renderer(4096, 0, with_tex=True)
    tex_sd = r.texEncoder.state_dict()
    for k, v in tex_sd.items():
        step = v.abs().max() / (2 if v.numel() > 10000 else 8)
        v.copy_(torch.round(v / step) * step)
    coarse, fine = mk_nerf(8, 64, 11, "ckc"), mk_nerf(10, 64, 11, "ckf")
    grad_vars = list(coarse.parameters()) + list(fine.parameters()) + list(r.grad_parameter())
    opt = torch.optim.Adam(params=grad_vars, lr=5e-5, betas=(0.9, 0.999))
    blob = {'global_step': 100, 'network_fn_state_dict': coarse.state_dict(), 'network_fine_state_dict': fine.state_dict(),
            'network_render_textureEncoder': r.texEncoder.state_dict(), 'network_render_idSpecific': r.idSpecificMod.state_dict(),
            'optimizer_state_dict': opt.state_dict(), 'expression_latent_codes_sigma': r.expCodes_Sigma}
    buf = io.BytesIO()
    torch.save(blob, buf)
    p = os.path.join(HERE, "ref_ckpt_000100.tar.gz")
    with gzip.GzipFile(p, "wb", compresslevel=9, mtime=0) as f:
        f.write(buf.getvalue())
    print(f"wrote ref_ckpt_000100.tar.gz: {os.path.getsize(p) / 1024:.1f} KiB (raw {len(buf.getvalue()) / 1024:.1f} KiB)")
    kw = kwargs_for(r, coarse, fine)
    bm, tex, exp = synth.codes(4)
    K8 = synth.intrinsics(8, 8)
    c2w = pose_spherical(35.0, 0.0, 16.0)[:3, :4]
    rng = np.random.default_rng(5)
    uv = torch.from_numpy(rng.uniform(0, 1, (512, 512, 3)).astype(np.float32))
    with torch.no_grad(), Recorder() as rec:
        rgb, disp, acc, ex = r.render_fitting(8, 8, K8, chunk=64, c2w=c2w, shapeCodes=bm, uvCodes=tex, expType=20,
                                              expCodes=exp, retraw=True, **kw)
    out = dict(c2w=c2w, K=K8, bm=bm, tex=tex, exp=exp, H=8, chunk=64, netchunk=4096, arch=np.array([8, 64, 10, 64]),
               rgb=rgb, disp=disp, acc=acc, rgb0=ex["rgb0"], disp0=ex["disp0"], acc0=ex["acc0"], z_std=ex["z_std"],
               z_coarse=rec.r2o[0]["z"], weights_coarse=rec.r2o[0]["weights"], z_samples=rec.spdf[0]["samples"],
               z_fine=rec.r2o[1]["z"], raw_coarse=rec.r2o[0]["raw"], raw_fine=rec.r2o[1]["raw"],
               weights_fine=rec.r2o[1]["weights"])
    ro, rd = get_rays(8, 8, K8, c2w)
    rays = torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0)
    r.expCodes_Sigma = r.expCodes_Sigma[:20]
    with torch.no_grad():
        rgb, disp, acc, ex = r.render(8, 8, K8, chunk=64, rays=rays, shapeCodes=bm.expand(64, 50), uvMap=uv, expType=7,
                                      retraw=True, **kw)
    out.update(t_tex_code=r.decoding_texCodes, t_rgb=rgb, t_acc=acc, t_rgb0=ex["rgb0"], t_acc0=ex["acc0"], t_raw=ex["raw"])
    save("ckpt_render.npz", out)


def g15_long_rays():
    """More than 256 samples per ray (the reference has no limit: N_samples / N_importance are free flags): 36 rays with 300 coarse +
    212 importance samples — pins the oracle, and through it the multi-pass compositing / resampling kernels, at that size."""
    K6 = np.array([[14.0625, 0, 3.0], [0, 14.0625, 3.0], [0, 0, 1]])       # 6x6 image, focal 1200 * 6 / 512
    _render_case("e2e_long.npz", 6, K6, 35.0, 8, 64, 10, 64, chunk=20, netchunk=4096, N_samples=300, N_importance=212)


class _DrawLog:
    """Record (or replace) what ``np.random.randn`` / ``np.random.choice`` hand to the reference's samplers, so that the device
    samplers can be fed the SAME draws (``draws=`` of mofanerf_amd.rays.train_pixels / fit_pixels)."""

    def __init__(self, zero_randn=False):
        self.randn, self.choice, self.zero = [], [], zero_randn

    def __enter__(self):
        self._r, self._c = np.random.randn, np.random.choice

        def randn(*shape):
            v = self._r(*shape)
            if self.zero:
                v = np.zeros_like(v)
            self.randn.append(v.copy())
            return v

        def choice(a, size=None, replace=True, p=None):
            v = self._c(a, size=size, replace=replace, p=p)
            self.choice.append(np.asarray(v).copy())
            return v

        np.random.randn, np.random.choice = randn, choice
        return self

    def __exit__(self, *a):
        np.random.randn, np.random.choice = self._r, self._c


def g14_samplers():
    """The scripts' landmark-biased pixel samplers, run as they are: ``run_train.LMModule.sample_point`` (3D landmarks projected with
    K and the pose, run_train.py:119-148) and ``run_fit.LMModule.sample_point`` (run_fit.py:35-82), under seeded ``np.random`` with
    the draws recorded.  The scripts only import with stand-ins for packages this image lacks (cv2, imageio, configargparse, dlib: none
    is touched by the samplers) and for the two numpy aliases numpy >= 1.24 removed (``np.long`` = the platform C long = int64,
    ``np.int`` = int, what they were when the scripts were written)."""
    for m in ("configargparse", "dlib"):
        sys.modules.setdefault(m, types.ModuleType(m))
    np.long, np.int = np.int64, int
    cwd = os.getcwd()
    os.chdir("/root/reference")          # run_fit's imports read ./configs/*.npy relative paths lazily; be where the scripts expect
    try:
        import run_train  # noqa: E402  (reference)
        import run_fit  # noqa: E402  (reference)
    finally:
        os.chdir(cwd)
    torch.autograd.set_detect_anomaly(False)
    out = {}
    # ---- training sampler: 68 synthetic 3D landmarks (stored x50, as the FaceScape table is), two identities x three expressions
    rng = np.random.default_rng(14)
    H = W = 512
    K = np.array([[1200.0, 0, 256], [0, 1200.0, 256], [0, 0, 1]])
    face = np.stack([rng.uniform(-1.6, 1.6, 68), rng.uniform(-2.0, 2.0, 68), rng.uniform(-0.3, 1.2, 68)], -1)
    table = (face[None, None] + rng.normal(0, 0.05, (2, 3, 68, 3))) * 50.0
    LM = object.__new__(run_train.LMModule)          # (its __init__ only loads ../data/1_975_landmarks.npy)
    LM.landmark, LM.H = table, H
    full = torch.reshape(torch.stack(torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W)), -1), [-1, 2])
    dH = dW = int(H // 2 * 0.5)
    crop = torch.reshape(torch.stack(torch.meshgrid(torch.linspace(H // 2 - dH, H // 2 + dH - 1, 2 * dH),
                                                    torch.linspace(W // 2 - dW, W // 2 + dW - 1, 2 * dW)), -1), [-1, 2])
    out.update(train_K=K, train_table=table, train_H=H)
    for tag, angle, ident, exp, coords, n in (("a", 25.0, 1, 2, full, 4096), ("b", -40.0, 0, 1, crop, 1024)):
        pose = pose_spherical(angle, 0.0, 16.0)[:3, :4]
        np.random.seed(140 + ord(tag))
        with _DrawLog() as log:
            sel = LM.sample_point(numOfPoint=n, K=K, pose=pose, id=torch.Tensor([ident]), exp=exp, coords=coords)
        with _DrawLog(zero_randn=True) as log0:          # zero offsets: the landmark part IS the projected table, repeated
            sel0 = LM.sample_point(numOfPoint=n, K=K, pose=pose, id=torch.Tensor([ident]), exp=exp, coords=coords)
        p = int(n / 5 * 3 // 68)
        lm2d = sel0[n - 68 * p:].reshape(68, p, 2)[:, 0]
        out.update({f"train_{tag}_pose": pose, f"train_{tag}_id": ident, f"train_{tag}_exp": exp, f"train_{tag}_n": n,
                    f"train_{tag}_precrop": 0.5 if coords is crop else 0.0, f"train_{tag}_rand": log.randn[0] * (H * 0.025),
                    f"train_{tag}_choice": log.choice[0], f"train_{tag}_lm2d": lm2d, f"train_{tag}_select": sel})
        print(f"train sampler {tag}: n={n} p={p} lm2d rows {int(lm2d[:, 0].min())}..{int(lm2d[:, 0].max())} cols {int(lm2d[:, 1].min())}..{int(lm2d[:, 1].max())}")
    # ---- fitting sampler: integer 2D landmarks on a 512 grid, a half-resolution target with an empty background
    lm512 = np.stack([np.linspace(150, 400, 68).round(), np.linspace(140, 380, 68)[::-1].round()], -1).astype(np.int64)
    lm512 = lm512[rng.permutation(68)]
    tgt = np.zeros((256, 256, 3), np.float32)
    tgt[60:215, 55:205] = rng.uniform(0.05, 1.0, (155, 150, 3)).astype(np.float32)
    small = np.zeros((256, 256, 3), np.float32)                      # almost empty target: fewer candidates than N_rand survive
    small[70:112, 60:190] = 0.5
    LMf = run_fit.LMModule(lm512, H=512)
    out.update(fit_lm=lm512, fit_target=tgt, fit_target_small=small)
    for tag, img, n in (("a", tgt, 1024), ("b", small, 1024)):
        np.random.seed(150 + ord(tag))
        with _DrawLog() as log:
            sel = LMf.sample_point(numOfPoint=n, coords=None, tar_img=img, scale=2)
        wid = 512 * 0.025 / 2
        out.update({f"fit_{tag}_n": n, f"fit_{tag}_rand": log.randn[0] * wid, f"fit_{tag}_rand_outline": log.randn[1] * wid,
                    f"fit_{tag}_choice": log.choice[0] if log.choice else np.zeros(0, np.int64), f"fit_{tag}_select": sel})
        print(f"fit sampler {tag}: n={n}, outline draws {log.randn[1].shape[0]}, branch = {'choice' if log.choice else 'repeat'}")
    save("kat_samplers.npz", out)


def g16_flags():
    """Networks built from NON-SHIPPED flags (tools/config_parser.py:51-56,113-118; tools/create_model_condition.py:16-34).

    ``kat_flags.npz`` — ``run_network``'s own arithmetic (embed_fn, cat the expression code, expand the codes, embeddirs_fn,
    ``NeRF.forward``; models/render_class.py:69-94) on explicit points for three flag sets, with the reference's autograd gradients
    of ``sum(raw * G)`` w.r.t. the points, the view directions and the three codes:
      * ``parser``: the parser's own defaults the shipped config overrides — input_ch_shapeCodes=80, input_ch_expCodes=6
        (multires=10, multires_views=4, texture 256).  (``myRenderer`` itself cannot run these two widths: its StyleModule is a fixed
        50 -> 30 map, models/render_class.py:51, models/model.py:175 — so this level, below the StyleModule, is where they can be pinned.)
      * ``low``: multires=6, multires_views=2, shape 80, exp 6, texture 256.
      * ``noembed``: i_embed=-1 (nn.Identity: 3 raw coordinates for points and directions), shape 50, exp 30, texture 64.
    ``flags_e2e.npz`` — the whole ``render_fitting`` path (StyleModule, coarse + fine, resampling, compositing) with multires=6,
    multires_views=2, texture code 128 (shape 50 / exp 30 as the StyleModule fixes them), all intermediates, and the gradients of the
    g4 loss w.r.t. the codes."""
    out = {}
    rng = np.random.default_rng(16)
    for tag, mr, mv, i_embed, ce, cs, ct, D, W in (("parser", 10, 4, 0, 6, 80, 256, 8, 64), ("low", 6, 2, 0, 6, 80, 256, 10, 64),
                                                   ("noembed", 10, 4, -1, 30, 50, 64, 8, 128)):
        embed_fn, ch = get_embedder(mr, i_embed)
        embeddirs_fn, chv = get_embedder(mv, i_embed)
        net = NeRF(D=D, W=W, input_ch_shapeCodes=cs, input_ch_textureCodes=ct, input_ch=ch + ce, output_ch=5, skips=[4],
                   input_ch_views=chv, use_viewdirs=True)
        net.load_state_dict(synth.nerf_state(D, W, 16, "flags", ch_pts=ch + ce, ch_shape=cs, ch_tex=ct, ch_views=chv))
        R, S = 9, 24
        pts = torch.from_numpy(rng.uniform(-8, 8, (R, S, 3)).astype(np.float32)).requires_grad_(True)
        vd = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(R, 3)).astype(np.float32)), dim=-1).requires_grad_(True)
        e = torch.from_numpy(rng.uniform(-1, 1, (1, ce)).astype(np.float32)).requires_grad_(True)
        bm = torch.from_numpy(rng.normal(0, 0.05, (1, cs)).astype(np.float32)).requires_grad_(True)
        tex = torch.from_numpy(rng.normal(0.2, 0.3, (1, ct)).astype(np.float32)).requires_grad_(True)
        G = torch.from_numpy(rng.normal(size=(R, S, 4)).astype(np.float32))
        n = R * S
        flat = pts.reshape(-1, 3)
        emb = torch.cat([embed_fn(flat), e.expand(n, -1)], -1)                                   # render_class.py:77-83
        dirs = embeddirs_fn(vd[:, None].expand(pts.shape).reshape(-1, 3))                       # :88-90
        raw = net(emb, bm.expand(n, -1), dirs, tex.expand(n, -1)).reshape(R, S, 4)              # :74,104 + model.py:121-137
        (raw * G).sum().backward()
        out.update({f"{tag}_flags": np.array([mr if i_embed != -1 else 0, mv if i_embed != -1 else 0, ce, cs, ct, D, W]),
                    f"{tag}_pts": pts, f"{tag}_vd": vd, f"{tag}_e": e, f"{tag}_bm": bm, f"{tag}_tex": tex, f"{tag}_G": G, f"{tag}_raw": raw,
                    f"{tag}_g_pts": pts.grad, f"{tag}_g_vd": vd.grad, f"{tag}_g_e": e.grad, f"{tag}_g_bm": bm.grad, f"{tag}_g_tex": tex.grad})
        print(f"kat_flags {tag}: input_ch={ch + ce} views={chv} raw absmax {float(raw.abs().max()):.3f}")
    save("kat_flags.npz", out)

    # ---- end to end
    mr, mv, ct = 6, 2, 128
    embed_fn, ch = get_embedder(mr, 0)
    embeddirs_fn, chv = get_embedder(mv, 0)
    r = render_class.myRenderer(embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=4096, uvCodesLen=ct, expCodesLen=30)
    r.idSpecificMod.load_state_dict(synth.style_state(0))
    for dst, src in zip(r.expCodes_Sigma, synth.exp_sigma(0)):
        dst.data[:] = src
    r.eval()
    nets = []
    for D, W, tg in ((8, 64, "coarse"), (10, 64, "fine")):
        m = NeRF(D=D, W=W, input_ch_shapeCodes=50, input_ch_textureCodes=ct, input_ch=ch + 30, output_ch=5, skips=[4],
                 input_ch_views=chv, use_viewdirs=True)
        m.load_state_dict(synth.nerf_state(D, W, 16, tg, ch_pts=ch + 30, ch_shape=50, ch_tex=ct, ch_views=chv))
        nets.append(m.eval())
    kw = kwargs_for(r, nets[0], nets[1])
    bm, tex, exp = synth.codes(0)
    tex = tex[:ct].clone()
    bm, tex, exp = [t.clone().requires_grad_(True) for t in (bm, tex, exp)]
    K16 = np.array([[37.5, 0, 8.0], [0, 37.5, 8.0], [0, 0, 1]])
    c2w = pose_spherical(35.0, 0.0, 16.0)[:3, :4]
    with Recorder() as rec:
        rgb, disp, acc, ex = r.render_fitting(16, 16, K16, chunk=96, c2w=c2w, shapeCodes=bm, uvCodes=tex, expType=20, expCodes=exp,
                                              retraw=True, **kw)
    loss = (rgb - 0.5).abs().mean() + (ex["rgb0"] ** 2).mean()
    loss.backward()
    o = dict(c2w=c2w, K=K16, bm=bm, tex=tex, exp=exp, H=16, chunk=96, netchunk=4096, multires=mr, multires_views=mv, ch_tex=ct,
             arch=np.array([8, 64, 10, 64]), seed=16, rgb=rgb, disp=disp, acc=acc, rgb0=ex["rgb0"], disp0=ex["disp0"], acc0=ex["acc0"],
             z_std=ex["z_std"], loss=loss, g_bm=bm.grad, g_tex=tex.grad, g_exp=exp.grad)
    nch = len(rec.spdf)
    o["z_coarse"] = torch.cat([rec.r2o[2 * i]["z"] for i in range(nch)])
    o["raw_coarse"] = torch.cat([rec.r2o[2 * i]["raw"] for i in range(nch)])
    o["weights_coarse"] = torch.cat([rec.r2o[2 * i]["weights"] for i in range(nch)])
    o["z_fine"] = torch.cat([rec.r2o[2 * i + 1]["z"] for i in range(nch)])
    o["raw_fine"] = torch.cat([rec.r2o[2 * i + 1]["raw"] for i in range(nch)])
    o["z_samples"] = torch.cat([s_["samples"] for s_ in rec.spdf])
    save("flags_e2e.npz", o)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16"]
    for w in which:
        {"g1": g1_kats, "g2": g2_small, "g3": g3_true, "g4": g4_grads, "g5": g5_render_tex, "g6": g6_schema, "g7": g7_config1, "g8": g8_true_grads, "g9": g9_run_network_kat,
         "g10": g10_checkpoint, "g11": g11_envelopes, "g12": g12_pose_grads, "g13": g13_ndc, "g14": g14_samplers, "g15": g15_long_rays, "g16": g16_flags}[w]()
