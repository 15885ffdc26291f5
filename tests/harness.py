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
    if kw_test["network_fine"] is not None:          # N_importance == 0: create_nerf builds no fine network
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


def classify_samples(z_coarse, weights_coarse, u, zs_hip, zs_ref, ulp_tol=6e-6, w_err=0.0, bins=None, bin_weights=None):
    """Decide which resampled positions legitimately differ between two fp32 implementations.

    ``sample_pdf`` (tools/run_nerf_helpers.py:242-245) divides by ``denom = cdf[i+1]-cdf[i]`` and REPLACES it by 1
    when ``denom < 1e-5``.  The cdf is a float near 1 (rounding noise ~1.2e-7 on a difference), so (a) a bin holding
    little probability mass makes ``t`` ill-conditioned (|dz| <= binwidth * noise / denom) and (b) a bin whose mass
    is within noise of 1e-5 — every empty bin of a near-opaque ray: 1e-5/(acc+62e-5) — flips the branch and moves the
    sample by up to a bin width.  The reference itself does this under a 1e-7 relative perturbation of its own
    coarse weights (measured: ~10 % of rays), so such samples cannot be held to 1e-4.

    ``w_err``: measured max |coarse weights(A) - coarse weights(B)| — the resampling INPUT differs by that much between
    the two implementations, which moves every cdf entry by up to a few times w_err on top of the rounding noise.

    Returns ``(agree [R,Ni] bool, explained [R,Ni] bool)`` from the REFERENCE's coarse weights: ``agree`` = within a
    few ulp of z; ``explained`` = the disagreement is within the conditioning bound or sits at the threshold."""
    zh, zr = torch.as_tensor(zs_hip).float(), torch.as_tensor(zs_ref).float()
    R, Ni = zr.shape
    u = torch.as_tensor(u).float().expand(R, Ni).contiguous()
    if bins is not None:       # plain sample_pdf(bins, weights) form (``z_coarse`` / ``weights_coarse`` unused)
        bins, ww = torch.as_tensor(bins).float(), torch.as_tensor(bin_weights).float() + 1e-5
    else:
        z, w = torch.as_tensor(z_coarse).float(), torch.as_tensor(weights_coarse).float()
        bins = .5 * (z[:, 1:] + z[:, :-1])
        ww = w[:, 1:-1] + 1e-5
    pdf = ww / ww.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros(R, 1), torch.cumsum(pdf, -1)], -1)
    B = cdf.shape[-1]
    inds = torch.searchsorted(cdf, u, right=True)
    noise = 2.5e-7 + 4.0 * float(w_err)
    expl = torch.zeros(R, Ni, dtype=torch.bool)
    bound = torch.zeros(R, Ni)
    for off in (-2, -1, 0, 1):                       # the bin the sample falls in and its neighbours
        lo = (inds + off).clamp(0, B - 2)
        den = torch.gather(cdf, -1, lo + 1) - torch.gather(cdf, -1, lo)
        width = torch.gather(bins, -1, lo + 1) - torch.gather(bins, -1, lo)
        expl |= (den - 1e-5).abs() <= 4e-7 + 2.0 * float(w_err)   # (b) threshold flip
        if off in (-1, 0):
            bound = torch.maximum(bound, width * noise / den.clamp_min(1e-5) * 4 + ulp_tol)
    # (c) searchsorted tie: u within noise of a cdf entry whose neighbouring bins are below the threshold — the sample
    #     is snapped to the left edge of whichever bin wins the tie (e.g. u = 1.0 against cdf[-1] = 1 -/+ 1 ulp)
    dens = cdf[:, 1:] - cdf[:, :-1]
    small = torch.nn.functional.pad(dens < 1e-5 + 4e-7, (1, 1), value=True)      # [R, B+1]: bin k-1 | bin k around entry k
    for off in (-2, -1, 0, 1):
        k = (inds + off).clamp(0, B - 1)
        tie = (u - torch.gather(cdf, -1, k)).abs() <= noise
        expl |= tie & (torch.gather(small, -1, k) | torch.gather(small, -1, k + 1))
    err = (zh - zr).abs()
    agree = err <= ulp_tol
    expl |= err <= bound                               # (a) ill-conditioned interpolation
    return agree, expl


PERT_EPS = 1e-6     # relative size of the coarse-weight perturbation behind every envelope (HIP's own coarse weights differ
                    # from the reference's by 5e-7..8e-7 absolute on weights <= 1: measured)


def oracle_envelope(o, ro, rd, chunk, bm, base_z_samples, exp_type=20, n_pert=8, eps=PERT_EPS, **kw):
    """The ORACLE's outputs under ``n_pert`` seeded relative perturbations of its own coarse weights (the same recipe
    tests/golden/make_golden.py::_envelope applies to the reference for the committed ``*_env.npz`` fixtures).  ``kw``: what
    ``OracleRenderer.render`` takes (tex_code / uv_map, exp_codes, N_samples, ...).  Returns the ``pert_*`` dict
    :func:`compare_render` takes."""
    R, S = ro.reshape(-1, 3).shape[0], kw.get("N_samples", 64)
    out = {"pert_rgb": [], "pert_acc": [], "pert_agree": []}
    for k in range(n_pert):
        m = torch.from_numpy(1.0 + np.random.default_rng(1000 + k).uniform(-eps, eps, (R, S))).float()
        with torch.no_grad():
            rgb, _, acc, ex = o.render(ro, rd, chunk, bm, exp_type, 8.0, 26.0, keep=True, w0_perturb=m, **kw)
        out["pert_rgb"].append(rgb.reshape(-1, 3).numpy()), out["pert_acc"].append(acc.reshape(-1).numpy())
        out["pert_agree"].append(((ex["_dbg"]["z_samples"] - torch.as_tensor(base_z_samples)).abs() <= 6e-6).all(-1).numpy())
    return {k: np.stack(v, 0) for k, v in out.items()}


def render_pair(H, K, angle, arch, chunk, netchunk, device="cuda", seed=0, n_rays=None, n_pert=8):
    """Render an HxH view (or its first ``n_rays`` rays) with the HIP path and with the oracle on identical rays.
    Returns two dicts of numpy arrays (rgb, disp, acc, rgb0, disp0, acc0, z_std) and the oracle's envelope."""
    render, kw, _ = make_product(arch, seed, netchunk, device)
    bm, tex, exp = synth.codes(seed)
    c2w = orc.pose_spherical(angle, 0.0, 16.0)[:3, :4]
    ro, rd = orc.get_rays(H, H, K, c2w)
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    if n_rays is not None:
        ro, rd = ro[:n_rays].contiguous(), rd[:n_rays].contiguous()
    rays = torch.stack([ro, rd], 0).to(device)
    with torch.no_grad():
        rgb, disp, acc, ex = render.render_fitting(H, H, K, chunk=chunk, rays=rays, shapeCodes=bm.to(device),
                                                   uvCodes=tex.to(device), expType=20, expCodes=exp.to(device),
                                                   verbose=True, **kw)
    torch.cuda.synchronize()
    hip = to_np(dict(rgb=rgb, disp=disp, acc=acc, rgb0=ex["rgb0"], disp0=ex["disp0"], acc0=ex["acc0"], z_std=ex["z_std"],
                     z_samples=ex["_z_samples"], z_fine=ex["_z_fine"], weights_coarse=ex["_weights0"]))
    o = make_oracle(arch, seed, netchunk)
    with torch.no_grad():
        rgb, disp, acc, ex = o.render(ro, rd, chunk, bm, 20, 8.0, 26.0, tex_code=tex, exp_codes=exp, N_samples=64,
                                      N_importance=64, keep=True)
    d = ex["_dbg"]
    ref = to_np(dict(rgb=rgb, disp=disp, acc=acc, rgb0=ex["rgb0"], disp0=ex["disp0"], acc0=ex["acc0"], z_std=ex["z_std"],
                     z_samples=d["z_samples"], z_coarse=d["z_coarse"], weights_coarse=d["weights_coarse"]))
    env = oracle_envelope(o, ro, rd, chunk, bm, ref["z_samples"], n_pert=n_pert, tex_code=tex, exp_codes=exp, N_samples=64,
                          N_importance=64)
    return hip, ref, env


ABS_CAP = 0.05     # absolute per-ray sanity cap on |rgb| / |acc| differences, envelope or not (largest legitimate move measured: 3.1e-2)


def envelope_stats(err, env, tol=1e-4, what="rgb"):
    """Is an implementation's per-ray error INSIDE the reference's own envelope?

    ``err [R]``: max-abs difference to the reference per ray.  ``env [K,R]``: the reference's (or oracle's) own per-ray
    change under K seeded ulp-level perturbations of its coarse weights (``*_env.npz`` / :func:`oracle_envelope`).  The
    resampling is discontinuous (tools/run_nerf_helpers.py:243) and the positional encoding amplifies a 1-ulp move of a sample
    by 2^9, so a small set of rays moves by up to 1e-2 in the REFERENCE ITSELF; which rays, and by how much, is what the
    envelope records.  Asserted, with R rays (slack = 3 rays):
      * frame level — fraction of rays over ``tol`` <= 1.4x the worst draw's; mean error <= 1.3x the worst draw's mean;
        max error <= 1.25x the envelope's max (+ tol) and, whatever the envelope says, <= ``ABS_CAP`` for EVERY ray;
      * per ray — err[r] <= max(tol, 1.5 * max_k env[k, r]) for >= 98.5 % of the rays (a flip the K draws did not sample is an
        outlier, and outliers are budgeted, not waved through: each is still below 1.25x the envelope's max);
      * rays that no draw moves by more than 1e-5 ("stable", the large majority) exceed ``tol`` in <= 2 % of the cases.
    The multipliers sit just above what the shipped kernels measure on MI355X against the committed reference fixtures
    (config 1, 4,096 rays: 1.26x / 1.18x / 1.00x / 98.9 %; every other fixture: <= 1.13x / <= 1.05x / <= 1.0x / >= 98.8 %), so that a
    regression which doubled the resampler's flip rate fails (round-2 gates were 2x / 2x / 1.5x / 97 %).
    Returns the measured numbers."""
    err, env = np.asarray(err, np.float64), np.asarray(env, np.float64)
    R = err.shape[0]
    slack = 3.0 / R
    env_ray = env.max(0)
    stable = env_ray <= 1e-5
    inside = err <= np.maximum(tol, 1.5 * env_ray)
    st = {"frac_over": float((err > tol).mean()), "ref_frac_over": float((env > tol).mean(1).max()),
          "mean": float(err.mean()), "ref_mean": float(env.mean(1).max()), "max": float(err.max()),
          "ref_max": float(env.max()), "inside": float(inside.mean()), "stable": float(stable.mean()),
          "stable_over": float((err[stable] > tol).mean()) if stable.any() else 0.0}
    msg = f"{what}: {st}"
    assert st["frac_over"] <= 1.4 * st["ref_frac_over"] + slack, msg
    assert st["mean"] <= 1.3 * st["ref_mean"] + 2e-6, msg
    assert st["max"] <= 1.25 * st["ref_max"] + tol and st["max"] <= ABS_CAP, msg
    assert st["inside"] >= 0.985 - slack, msg
    assert st["stable_over"] <= 0.02 + slack, msg
    return st


def compare_render(hip, ref, env, u=None, tol=1e-4, verbose=True, expect_ab=None):
    """Parity of a coarse+fine render (dicts of numpy arrays, rays flat).

    * coarse outputs (rgb0/acc0/disp0, coarse weights): every ray, ``tol``;
    * resampled positions: every disagreement must be EXPLAINED (see :func:`classify_samples`), and the fraction of rays
      whose 64 new positions all agree within a few ulp (tiers A + B) must be at least 0.6x what the reference shows against
      ITSELF under ulp-level noise (``env["pert_agree"]``; measured 0.65x - 1.0x) and, for the committed reference fixtures, must
      not drop below the number measured on MI355X with the shipped kernels (``expect_ab``, deterministic across boxes) by more
      than 0.03;
    * fine outputs (rgb/acc/disp/z_std), by how the ray's 64 new sample positions compare:
        A  bit-identical positions ......... ``tol`` (1e-4; only MLP/composite rounding is left)
        B  all within a few ulp of z ....... 1e-3: the reference amplifies a 1-ulp position change by 2^9*|d| in the
                                             positional encoding (measured on the oracle by tests/test_oracle_sensitivity.py)
      and for ALL rays together: inside the reference's own envelope, :func:`envelope_stats` (rgb and acc);
      the strict fine-pass gate for ALL rays is the teacher-forced test (reference sample positions fed to the HIP
      network + compositing), tests/test_gpu_render.py::test_fine_pass_teacher_forced_*, tests/test_gpu_config1.py.
    ``env``: dict with ``pert_rgb [K,R,3]``, ``pert_acc [K,R]``, ``pert_agree [K,R]`` (fixture or :func:`oracle_envelope`).
    Returns a dict of measured errors."""
    from conftest import nan_equal_close
    flat = lambda a, n: np.asarray(a).reshape(-1, *np.asarray(a).shape[-n:]) if n else np.asarray(a).reshape(-1)
    out = {}
    out["rgb0"] = nan_equal_close(flat(hip["rgb0"], 1), flat(ref["rgb0"], 1), tol)
    out["acc0"] = nan_equal_close(flat(hip["acc0"], 0), flat(ref["acc0"], 0), tol)
    out["disp0"] = nan_equal_close(flat(hip["disp0"], 0), flat(ref["disp0"], 0), 1e-6, 1e-4)
    if "weights_coarse" in hip:
        out["weights0"] = nan_equal_close(flat(hip["weights_coarse"], 1), flat(ref["weights_coarse"], 1), 2e-5)
    zh, zr = flat(hip["z_samples"], 1), flat(ref["z_samples"], 1)
    if u is None:
        u = torch.linspace(0., 1., zr.shape[-1])
    zc = flat(ref["z_coarse"], 1) if "z_coarse" in ref else np.broadcast_to(flat(ref["z_coarse_row"], 1), (zr.shape[0], flat(ref["z_coarse_row"], 1).shape[-1]))
    agree, expl = classify_samples(zc, flat(ref["weights_coarse"], 1), u, zh, zr, w_err=out.get("weights0", 0.0))
    bad = ~(agree | expl)
    assert not bad.any(), f"{int(bad.sum())} resampled positions differ without being ill-conditioned/at the threshold"
    tier_a = (zh == zr).all(-1)
    tier_b = agree.all(-1).numpy() & ~tier_a
    out["frac_A_B_rest"] = [round(float(t.mean()), 3) for t in (tier_a, tier_b, ~(tier_a | tier_b))]
    out["ref_self_agree"] = round(float(np.asarray(env["pert_agree"]).mean(1).min()), 3)
    assert (tier_a | tier_b).mean() >= 0.6 * out["ref_self_agree"] - 3.0 / zr.shape[0], out
    if expect_ab is not None:
        assert (tier_a | tier_b).mean() >= expect_ab - 0.03, (out, expect_ab)
    for name, mask, t in (("A", tier_a, tol), ("B", tier_b, 1e-3)):
        if not mask.any():
            continue
        out["rgb_" + name] = nan_equal_close(flat(hip["rgb"], 1)[mask], flat(ref["rgb"], 1)[mask], t)
        out["acc_" + name] = nan_equal_close(flat(hip["acc"], 0)[mask], flat(ref["acc"], 0)[mask], t)
        nan_equal_close(flat(hip["disp"], 0)[mask], flat(ref["disp"], 0)[mask], 10 * t * 1e-2, 10 * t)
        out["z_std_" + name] = nan_equal_close(flat(hip["z_std"], 0)[mask], flat(ref["z_std"], 0)[mask], 1e-5)
    rgb_r, acc_r = flat(ref["rgb"], 1), flat(ref["acc"], 0)
    out["env_rgb"] = envelope_stats(np.abs(flat(hip["rgb"], 1) - rgb_r).max(-1),
                                    np.abs(np.asarray(env["pert_rgb"]) - rgb_r[None]).max(-1), tol, "rgb")
    out["env_acc"] = envelope_stats(np.abs(flat(hip["acc"], 0) - acc_r), np.abs(np.asarray(env["pert_acc"]) - acc_r[None]), tol, "acc")
    mse = float(((flat(hip["rgb"], 1).astype(np.float64) - rgb_r) ** 2).mean())
    out["psnr_db"] = round(-10 * np.log10(max(mse, 1e-30)), 1)
    ref_mse = float(((np.asarray(env["pert_rgb"], np.float64) - rgb_r[None]) ** 2).mean((1, 2)).max())
    out["ref_psnr_db"] = round(-10 * np.log10(max(ref_mse, 1e-30)), 1)
    assert out["psnr_db"] >= out["ref_psnr_db"] - 3.0 and out["psnr_db"] >= 60.0, out      # frame level: >= 60 dB and within 3 dB of the reference vs itself
    if verbose:
        fmt = lambda v: f"{v:.2e}" if isinstance(v, float) else ({k: (f"{x:.2e}" if isinstance(x, float) else x) for k, x in v.items()} if isinstance(v, dict) else v)
        print({k: fmt(v) for k, v in out.items()})
    return out
