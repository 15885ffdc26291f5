"""Pin oracle/mofa_oracle.py against fixtures produced by the reference itself (CPU only).

The reference has no tests of its own; tests/golden/make_golden.py ran the reference
in the build container and these tests replay its inputs through the oracle.  Same torch CPU ops in
the same order => tolerances are at the fp32 rounding floor (mostly bit-exact).
"""
import os

import numpy as np
import torch

from conftest import nan_equal_close
from mofanerf_amd import synth
from oracle import mofa_oracle as orc

T = torch.from_numpy


def test_embedder(golden):
    g = golden("kat.npz")
    x = T(g["embed_x"])
    assert np.array_equal(orc.positional_encode(x, 10).numpy(), g["embed_L10"])
    assert np.array_equal(orc.positional_encode(x, 4).numpy(), g["embed_L4"])
    e = orc.positional_encode(torch.tensor([[0.5, -1.25, 2.0]]), 10)[0]
    assert abs(float(e.sum()) - 4.0605088) < 1e-5          # SURVEY.md §8c sanity anchor


def test_raw2outputs(golden):
    g = golden("kat.npz")
    for S in (64, 128):
        raw, z, d = T(g[f"r2o{S}_raw"]), T(g[f"r2o{S}_z"]), T(g[f"r2o{S}_d"])
        for wb in (0, 1):
            out = orc.raw2outputs(raw, z, d, None, bool(wb))
            for n, v in zip(("rgb", "disp", "acc", "weights", "depth"), out):
                nan_equal_close(v.numpy(), g[f"r2o{S}_{wb}_{n}"], 0.0)
        assert np.isnan(g[f"r2o{S}_0_disp"][0])            # the zero-opacity ray
        np.random.seed(0)
        noise = T(np.random.rand(*raw.shape[:2]).astype(np.float64) * 0.7).float()
        out = orc.raw2outputs(raw, z, d, noise, False)
        for n, v in zip(("rgb", "disp", "acc", "weights", "depth"), out):
            nan_equal_close(v.numpy(), g[f"r2o{S}_noise_{n}"], 1e-7, 1e-6)
    np.testing.assert_allclose(g["r2o_anchor_rgb"][0], [0.52549005, 0.45028955, 0.57368523], rtol=1e-6)


def test_sample_pdf(golden):
    g = golden("kat.npz")
    bins, w = T(g["spdf_bins"]), T(g["spdf_w"])
    assert np.array_equal(orc.sample_pdf(bins, w, torch.linspace(0, 1, 64)).numpy(), g["spdf_det"])
    np.random.seed(0)
    u = torch.Tensor(np.random.rand(bins.shape[0], 64))
    assert np.array_equal(orc.sample_pdf(bins, w, u).numpy(), g["spdf_rand"])
    np.testing.assert_allclose(g["spdf_anchor"][0], [8.0, 14.178512, 14.857115, 15.535717, 16.214321, 16.892923,
                                                     18.714521, 26.0], rtol=1e-6)


def test_rays_and_pose(golden):
    g = golden("kat.npz")
    K = np.array([[600., 0, 128], [0, 600., 128], [0, 0, 1]])
    for ang in (-60, 0, 60):
        c2w = orc.pose_spherical(float(ang), 0.0, 16.0)
        assert np.array_equal(c2w.numpy(), g[f"rays{ang}_c2w"])
        ro, rd = orc.get_rays(256, 256, K, c2w[:3, :4])
        assert np.array_equal(ro[0, 0].numpy(), g[f"rays{ang}_o"])
        assert np.array_equal(rd[::37, ::41].numpy(), g[f"rays{ang}_d_sub"])
    assert np.array_equal(orc.pose_spherical(-17.0, 23.0, 16.0).numpy(), g["pose_m17_23_16"])


def test_nerf_forward_and_style(golden):
    g = golden("kat.npz")
    for D, W in ((8, 64), (10, 64), (8, 96)):
        st = synth.nerf_state(D, W)
        n = g[f"nerf{D}x{W}_pts"].shape[0]
        out = orc.nerf_forward(st, T(g[f"nerf{D}x{W}_pts"]), T(g[f"nerf{D}x{W}_bm"]).expand(n, -1),
                               T(g[f"nerf{D}x{W}_views"]), T(g[f"nerf{D}x{W}_tex"]).expand(n, -1))
        nan_equal_close(out.numpy(), g[f"nerf{D}x{W}_out"], 1e-6)
    s, b = orc.style_module(synth.style_state(0), T(g["style_bm"]))
    nan_equal_close(s.numpy(), g["style_scale"], 1e-7)
    nan_equal_close(b.numpy(), g["style_bias"], 1e-7)


def _oracle_for(g, with_tex=False):
    Dc, Wc, Df, Wf = [int(v) for v in g["arch"]] if "arch" in g else (8, 64, 10, 64)
    seed = int(g["seed"]) if "seed" in g else 0
    return orc.OracleRenderer(synth.nerf_state(Dc, Wc, seed, "coarse"), synth.nerf_state(Df, Wf, seed, "fine"),
                              synth.style_state(seed), synth.exp_sigma(seed),
                              synth.tex_encoder_state(seed) if with_tex else None,
                              netchunk=int(g["netchunk"]) if "netchunk" in g else 4096)


def _check_e2e(g, kw, tol):
    r = _oracle_for(g)
    H = int(g["H"])
    ro, rd = orc.get_rays(H, H, g["K"], T(g["c2w"]))
    with torch.no_grad():
        Ns, Ni = (int(g["N_samples"]), int(g["N_importance"])) if "N_samples" in g else (64, 64)
        rgb, disp, acc, ex = r.render(ro, rd, int(g["chunk"]), T(g["bm"]), 20, 8.0, 26.0, tex_code=T(g["tex"]),
                                      exp_codes=T(g["exp"]), N_samples=Ns, N_importance=Ni, retraw=True, **kw)
    errs = {}
    for n, v in (("rgb", rgb), ("disp", disp), ("acc", acc), ("rgb0", ex["rgb0"]), ("disp0", ex["disp0"]),
                 ("acc0", ex["acc0"]), ("z_std", ex["z_std"])):
        errs[n] = nan_equal_close(v.numpy(), g[n], tol, tol)
    nan_equal_close(ex["raw"].reshape(-1, Ns + Ni, 4).numpy(), g["raw_fine"], 10 * tol, tol)
    assert ex["losses"] == 0
    return errs


def test_e2e_small(golden):
    _check_e2e(golden("e2e_small.npz"), {}, 2e-6)


def test_e2e_small_stochastic(golden):
    g = golden("e2e_small_stoch.npz")
    R = int(g["H"]) ** 2
    np.random.seed(0); t_rand = torch.Tensor(np.random.rand(R, 64))
    np.random.seed(0); u = torch.Tensor(np.random.rand(R, 64))
    np.random.seed(0); n0 = torch.Tensor(np.random.rand(R, 64) * float(g["noise"]))
    np.random.seed(0); n1 = torch.Tensor(np.random.rand(R, 128) * float(g["noise"]))
    _check_e2e(g, dict(perturb=1.0, white_bkgd=True, t_rand=t_rand, u=u, noise0=n0, noise1=n1), 2e-6)


def test_e2e_long_rays(golden):
    """300 coarse + 212 importance samples per ray (beyond the 256 one wavefront pass holds on the device)."""
    _check_e2e(golden("e2e_long.npz"), {}, 2e-6)


def test_e2e_true_size(golden):
    """64 rays through coarse 256x8 + fine 1024x10 (the shipped sizes), recipe weights."""
    _check_e2e(golden("e2e_true.npz"), {}, 5e-6)


def test_render_with_tex_encoder(golden):
    g = golden("render_tex.npz")
    uv = T(np.random.default_rng(5).uniform(0, 1, (512, 512, 3)).astype(np.float32))
    assert abs(float(uv.double().sum()) - float(g["uv_sum"])) < 1e-6
    r = _oracle_for(g, with_tex=True)
    rays = T(g["rays"])
    with torch.no_grad():
        code = orc.tex_encoder(r.tex_enc, uv)
        rgb, disp, acc, ex = r.render(rays[0], rays[1], 64, T(g["bm"]), 7, 8.0, 26.0, uv_map=uv, N_samples=64,
                                      N_importance=64)
    nan_equal_close(code.numpy(), g["tex_code"], 1e-6, 1e-5)
    nan_equal_close(rgb.numpy(), g["rgb"], 2e-6)
    nan_equal_close(acc.numpy(), g["acc"], 2e-6)
    nan_equal_close(disp.numpy(), g["disp"], 2e-6, 2e-6)


# ---- round 2 fixtures ------------------------------------------------------------------------------------------------------
def test_run_network_kat(golden):
    """run_network (render_class.py:69-94) from explicit points / view directions / codes — the reference's outputs."""
    g = golden("kat_run_network.npz")
    for D, W in ((8, 64), (10, 64), (8, 96), (10, 128)):
        t = f"rn{D}x{W}"
        _, _, netchunk, wseed, exp_type = [int(v) for v in g[t + "_meta"]]
        r = orc.OracleRenderer(synth.nerf_state(D, W, wseed, "kat"), None, synth.style_state(0), synth.exp_sigma(0),
                               netchunk=netchunk)
        with torch.no_grad():
            raw = r.run_network(T(g[t + "_pts"]), T(g[t + "_vd"]), r.coarse, T(g[t + "_bm"]), T(g[t + "_tex"]), exp_type)
        nan_equal_close(raw.numpy(), g[t + "_raw"], 2e-6, 2e-6)


def _sampled_idx(key, numel, n=256):
    import zlib
    return np.random.default_rng(zlib.crc32(key.encode())).integers(0, numel, size=min(n, numel))


def test_true_size_gradients(golden):
    """Backward through run_network at the SHIPPED widths (fixture g8): the oracle's fp32 autograd reproduces the reference's
    gradients w.r.t. rays, codes and (sampled entries + norms of) every weight and bias."""
    g = golden("grads_true.npz")
    for tag in ("coarse", "fine"):
        D, W = [int(v) for v in g[f"{tag}_arch"]]
        st = {k: v.clone().requires_grad_(True) for k, v in synth.nerf_state(D, W, 0, tag).items()}
        style = {k: v.clone().requires_grad_(True) for k, v in synth.style_state(0).items()}
        bm, tex, exp = [T(g[f"{tag}_{k}"]).clone().requires_grad_(True) for k in ("bm", "tex", "exp")]
        o, d = [T(g[f"{tag}_{k}"]).clone().requires_grad_(True) for k in ("o", "d")]
        z, G = T(g[f"{tag}_z"]), T(g[f"{tag}_G"])
        r = orc.OracleRenderer(st, None, style, synth.exp_sigma(0) + [exp], netchunk=196608)
        pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
        vd = d / torch.norm(d, dim=-1, keepdim=True)
        raw = r.run_network(pts, vd, st, bm.expand(o.shape[0], 50), tex, 20)
        nan_equal_close(raw.detach().numpy(), g[f"{tag}_raw"], 5e-6, 5e-6)
        (raw * G).sum().backward()
        rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))
        for name, t in (("g_o", o), ("g_d", d), ("g_bm", bm), ("g_tex", tex), ("g_exp", exp)):
            assert rel(t.grad.numpy(), g[f"{tag}_{name}"]) < 2e-5, (tag, name)
        for key, p in list(st.items()) + [("style." + k, v) for k, v in style.items()]:
            ref_s, ref_n = g[f"{tag}_gs/{key}"], float(g[f"{tag}_gn/{key}"])
            got = p.grad.reshape(-1)[_sampled_idx(f"{tag}/{key}", p.numel())].numpy()
            assert np.abs(got - ref_s).max() <= 2e-5 * (np.abs(ref_s).max() + 1e-30) + 1e-6 * ref_n / np.sqrt(p.numel()) + 1e-12, (tag, key)
            assert abs(float(p.grad.double().norm()) - ref_n) <= 1e-5 * ref_n + 1e-12, (tag, key)
        # the fixture's fp64 yardstick (the reference's modules in double) against the oracle in double
        st64 = {k: v.detach().double().requires_grad_(True) for k, v in st.items()}
        sty64 = {k: v.detach().double().requires_grad_(True) for k, v in style.items()}
        bm64, tex64, exp64 = [t.detach().double().requires_grad_(True) for t in (bm, tex, exp)]
        o64, d64 = o.detach().clone().requires_grad_(True), d.detach().clone().requires_grad_(True)
        r64 = orc.OracleRenderer(st64, None, sty64, [e.double() for e in synth.exp_sigma(0)] + [exp64], netchunk=196608)
        raw64 = r64.run_network((o64[:, None, :] + d64[:, None, :] * z[:, :, None]).double(),
                                (d64 / torch.norm(d64, dim=-1, keepdim=True)).double(), st64, bm64.expand(o.shape[0], 50), tex64, 20)
        (raw64 * G.double()).sum().backward()
        for name, t in (("raw", raw64.detach()), ("g_o", o64.grad), ("g_d", d64.grad), ("g_bm", bm64.grad), ("g_tex", tex64.grad), ("g_exp", exp64.grad)):
            assert rel(t.numpy(), g[f"{tag}_t_{name}"]) < 1e-9, (tag, name)
        key = "linear_BiM_xyz.linears1.Linear1.weight"
        assert abs(float(st64[key].grad.norm()) - float(g[f"{tag}_tn/{key}"])) <= 1e-9 * float(g[f"{tag}_tn/{key}"])


def test_config1_fixture_teacher_forced_subset(golden):
    """BASELINE config 1 (64x64, chunk 4096, shipped sizes): the oracle's FINE pass on the fixture's own sample positions
    reproduces the reference's raw / weights on the 256 rays the fixture keeps per-sample arrays for, and the fixture is
    self-consistent (its z_fine is the sort of z_coarse and z_samples; its envelope entries are non-trivial)."""
    g = golden("e2e_c1.npz")
    sub = g["sub"]
    H = int(g["H"])
    r = _oracle_for(g)
    r.exp_sigma.append(T(g["exp"]))
    ro, rd = orc.get_rays(H, H, g["K"], T(g["c2w"]))
    ro, rd = ro.reshape(-1, 3)[sub], rd.reshape(-1, 3)[sub]
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    zc = T(g["z_coarse_row"]).expand(len(sub), -1)
    zf = torch.sort(torch.cat([zc, T(g["z_samples"])[sub]], -1), -1)[0]
    with torch.no_grad():
        raw = r.run_network(ro[:, None, :] + rd[:, None, :] * zf[:, :, None], vd, r.fine, T(g["bm"]), T(g["tex"]), 20)
        rgb, disp, acc, w, _ = orc.raw2outputs(raw, zf, rd)
    nan_equal_close(raw.numpy(), g["raw_fine_sub"], 2e-5, 2e-5)
    nan_equal_close(w.numpy(), g["weights_fine_sub"], 2e-6)
    nan_equal_close(rgb.numpy(), g["rgb"].reshape(-1, 3)[sub], 2e-6)
    nan_equal_close(acc.numpy(), g["acc"].reshape(-1)[sub], 2e-6)
    env = np.abs(g["pert_rgb"] - g["rgb"].reshape(1, -1, 3)).max(-1)           # [n_pert, 4096]
    assert env.shape[1] == H * H and (env.max(0) > 1e-4).mean() > 0.005       # the reference itself moves under ulp noise


def test_reference_checkpoint_renders(golden, tmp_path):
    """The checkpoint written by the reference's modules (g10) loads into the oracle's state dicts and reproduces the
    reference's renders from it (render_fitting and the texture-encoder entry)."""
    import gzip
    raw = gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_ckpt_000100.tar.gz")).read()
    p = tmp_path / "000100.tar"
    p.write_bytes(raw)
    ck = torch.load(str(p), map_location="cpu", weights_only=False)
    assert ck["global_step"] == 100 and len(ck["expression_latent_codes_sigma"]) == 20
    g = golden("ckpt_render.npz")
    r = orc.OracleRenderer(ck["network_fn_state_dict"], ck["network_fine_state_dict"], ck["network_render_idSpecific"],
                           [t.detach() for t in ck["expression_latent_codes_sigma"]], ck["network_render_textureEncoder"],
                           netchunk=4096)
    ro, rd = orc.get_rays(8, 8, g["K"], T(g["c2w"]))
    with torch.no_grad():
        rgb, disp, acc, ex = r.render(ro, rd, 64, T(g["bm"]), 20, 8.0, 26.0, tex_code=T(g["tex"]), exp_codes=T(g["exp"]),
                                      N_samples=64, N_importance=64)
    nan_equal_close(rgb.numpy(), g["rgb"], 2e-6)
    nan_equal_close(acc.numpy(), g["acc"], 2e-6)
    r.exp_sigma = r.exp_sigma[:20]
    uv = T(np.random.default_rng(5).uniform(0, 1, (512, 512, 3)).astype(np.float32))
    with torch.no_grad():
        rgb, disp, acc, ex = r.render(ro.reshape(-1, 3), rd.reshape(-1, 3), 64, T(g["bm"]), 7, 8.0, 26.0, uv_map=uv,
                                      N_samples=64, N_importance=64)
        code = orc.tex_encoder(r.tex_enc, uv)
    nan_equal_close(code.numpy(), g["t_tex_code"], 1e-6, 1e-5)
    nan_equal_close(rgb.numpy(), g["t_rgb"], 2e-6)
    nan_equal_close(acc.numpy(), g["t_acc"], 2e-6)


def test_ndc_rays(golden):
    g = golden("kat_ndc.npz")
    no, nd = orc.ndc_rays(16, 16, g["K"][0][0], 1., T(g["rays_o"]), T(g["rays_d"]))
    assert np.array_equal(no.numpy(), g["ndc_o"]) and np.array_equal(nd.numpy(), g["ndc_d"])
    from mofanerf_amd import rays as mrays                       # the product's torch expression, on CPU tensors
    po, pd = mrays.ndc_rays(16, 16, float(g["K"][0][0]), 1., T(g["rays_o"]), T(g["rays_d"]))
    nan_equal_close(po.numpy(), g["ndc_o"], 1e-6, 1e-6)
    nan_equal_close(pd.numpy(), g["ndc_d"], 1e-6, 1e-6)


FLAG_TAGS = ("parser", "low", "noembed")


def flags_net_state(g, tag):
    """The seeded weights g16_flags loaded into the reference's NeRF for flag set ``tag`` (tests/golden/make_golden.py)."""
    mr, mv, ce, cs, ct, D, W = [int(v) for v in g[f"{tag}_flags"]]
    return (mr, mv, ce, cs, ct, D, W), synth.nerf_state(D, W, 16, "flags", ch_pts=3 + 6 * mr + ce, ch_shape=cs, ch_tex=ct,
                                                       ch_views=3 + 6 * mv)


def test_non_shipped_flags_run_network_level(golden):
    """multires / multires_views / i_embed / input_ch_*Codes off their shipped values (incl. the parser's own 80 / 6 defaults):
    the oracle's encoding + network on explicit points against the reference's, forward and every input gradient."""
    g = golden("kat_flags.npz")
    for tag in FLAG_TAGS:
        (mr, mv, ce, cs, ct, D, W), st = flags_net_state(g, tag)
        leaf = lambda k: T(g[f"{tag}_{k}"]).clone().requires_grad_(True)
        pts, vd, e, bm, tex = leaf("pts"), leaf("vd"), leaf("e"), leaf("bm"), leaf("tex")
        R, S = pts.shape[:2]
        n = R * S
        emb = torch.cat([orc.positional_encode(pts.reshape(-1, 3), mr), e.expand(n, -1)], -1)
        dirs = orc.positional_encode(vd[:, None].expand(R, S, 3).reshape(-1, 3), mv)
        raw = orc.nerf_forward(st, emb, bm.expand(n, -1), dirs, tex.expand(n, -1)).reshape(R, S, 4)
        (raw * T(g[f"{tag}_G"])).sum().backward()
        nan_equal_close(raw.detach().numpy(), g[f"{tag}_raw"], 1e-6, 1e-6)
        for k, v in (("pts", pts), ("vd", vd), ("e", e), ("bm", bm), ("tex", tex)):
            nan_equal_close(v.grad.numpy(), g[f"{tag}_g_{k}"], 1e-5 * float(np.abs(g[f"{tag}_g_{k}"]).max()), 1e-5)


def flags_e2e_oracle(g):
    mr, mv, ct = int(g["multires"]), int(g["multires_views"]), int(g["ch_tex"])
    Dc, Wc, Df, Wf = [int(v) for v in g["arch"]]
    w = dict(ch_pts=3 + 6 * mr + 30, ch_shape=50, ch_tex=ct, ch_views=3 + 6 * mv)
    return orc.OracleRenderer(synth.nerf_state(Dc, Wc, 16, "coarse", **w), synth.nerf_state(Df, Wf, 16, "fine", **w), synth.style_state(0),
                              synth.exp_sigma(0), netchunk=int(g["netchunk"]), multires=mr, multires_views=mv)


def test_non_shipped_flags_end_to_end(golden):
    """render_fitting with multires=6, multires_views=2 and a 128-wide texture code: outputs, intermediates and code gradients."""
    g = golden("flags_e2e.npz")
    r = flags_e2e_oracle(g)
    H = int(g["H"])
    ro, rd = orc.get_rays(H, H, g["K"], T(g["c2w"]))
    bm, tex, exp = [T(g[k]).clone().requires_grad_(True) for k in ("bm", "tex", "exp")]
    rgb, disp, acc, ex = r.render(ro, rd, int(g["chunk"]), bm, 20, 8.0, 26.0, tex_code=tex, exp_codes=exp, N_samples=64,
                                  N_importance=64, retraw=True, keep=True)
    for n, v in (("rgb", rgb), ("disp", disp), ("acc", acc), ("rgb0", ex["rgb0"]), ("disp0", ex["disp0"]), ("acc0", ex["acc0"]),
                 ("z_std", ex["z_std"])):
        nan_equal_close(v.detach().numpy(), g[n], 2e-6, 2e-6)
    nan_equal_close(ex["_dbg"]["z_samples"].detach().numpy(), g["z_samples"], 2e-6)
    nan_equal_close(ex["raw"].detach().reshape(-1, 128, 4).numpy(), g["raw_fine"], 2e-5, 2e-6)
    loss = (rgb - 0.5).abs().mean() + (ex["rgb0"] ** 2).mean()
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    for k, v in (("bm", bm), ("tex", tex), ("exp", exp)):
        nan_equal_close(v.grad.numpy(), g["g_" + k], 2e-5 * float(np.abs(g["g_" + k]).max()), 1e-4)
