"""Shared helpers for the GPU parity tests, ``__graft_entry__.smoke()`` and ``bench.py``'s checker leg:
build the product renderer and the oracle from the SAME seeded weights and run them on the SAME rays."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mofanerf_amd import factory, synth  # noqa: E402
from oracle import mofa_oracle as orc  # noqa: E402


def make_product(arch, seed=0, netchunk=4096, device="cuda", with_tex=False, N_samples=64, N_importance=64):
    """Product renderer + ``render_kwargs_test`` exactly as ``create_nerf`` hands them to the scripts."""
    Dc, Wc, Df, Wf = arch
    args = factory.default_args(netdepth=Dc, netwidth=Wc, netdepth_fine=Df, netwidth_fine=Wf, netchunk=netchunk,
                                no_reload=True, device=device, basedir="/nonexistent", N_samples=N_samples,
                                N_importance=N_importance)
    kw_train, kw_test, _, _, _, _, render = factory.create_nerf(args)
    kw_test["network_fn"].load_state_dict(synth.nerf_state(Dc, Wc, seed, "coarse"))
    kw_test["network_fine"].load_state_dict(synth.nerf_state(Df, Wf, seed, "fine"))
    render.idSpecificMod.load_state_dict(synth.style_state(seed))
    if with_tex:
        render.texEncoder.load_state_dict(synth.tex_encoder_state(seed))
    for dst, src in zip(render.expCodes_Sigma, synth.exp_sigma(seed)):
        dst.data[:] = src.to(dst.device)
    kw_test.update(near=8.0, far=26.0)
    kw_train.update(near=8.0, far=26.0)
    return render.eval(), kw_test, kw_train


def make_oracle(arch, seed=0, netchunk=4096, with_tex=False):
    Dc, Wc, Df, Wf = arch
    return orc.OracleRenderer(synth.nerf_state(Dc, Wc, seed, "coarse"), synth.nerf_state(Df, Wf, seed, "fine"),
                              synth.style_state(seed), synth.exp_sigma(seed),
                              synth.tex_encoder_state(seed) if with_tex else None, netchunk=netchunk)


def to_np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in d.items()}


def render_pair(H, K, angle, arch, chunk, netchunk, device="cuda", seed=0, n_rays=None):
    """Render an HxH view (or its first ``n_rays`` rays) with the HIP path and with the oracle on identical rays.
    Returns two dicts of numpy arrays (rgb, disp, acc, rgb0, disp0, acc0, z_std)."""
    render, kw, _ = make_product(arch, seed, netchunk, device)
    bm, tex, exp = synth.codes(seed)
    c2w = orc.pose_spherical(angle, 0.0, 16.0)[:3, :4]
    ro, rd = orc.get_rays(H, H, K, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    if n_rays is not None:
        ro, rd = ro[:n_rays].contiguous(), rd[:n_rays].contiguous()
    rays = torch.stack([ro, rd], 0).to(device)
    rgb, disp, acc, ex = render.render_fitting(H, H, K, chunk=chunk, rays=rays, shapeCodes=bm.to(device),
                                               uvCodes=tex.to(device), expType=20, expCodes=exp.to(device), **kw)
    torch.cuda.synchronize()
    hip = to_np(dict(rgb=rgb, disp=disp, acc=acc, rgb0=ex["rgb0"], disp0=ex["disp0"], acc0=ex["acc0"], z_std=ex["z_std"]))
    o = make_oracle(arch, seed, netchunk)
    with torch.no_grad():
        rgb, disp, acc, ex = o.render(ro, rd, chunk, bm, 20, 8.0, 26.0, tex_code=tex, exp_codes=exp, N_samples=64,
                                      N_importance=64)
    ref = to_np(dict(rgb=rgb, disp=disp, acc=acc, rgb0=ex["rgb0"], disp0=ex["disp0"], acc0=ex["acc0"], z_std=ex["z_std"]))
    return hip, ref
