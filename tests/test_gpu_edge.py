"""Edge cases of the drop-in boundary on the GPU: ragged sample counts, coarse-only rendering, lindisp, static-camera view
directions, tiny ray counts, netchunk smaller than one ray, DataParallel-wrapped networks, stochastic sampling with the
device RNG, render_path output/resume, and the loud failures for unsupported flags."""
import os

import numpy as np
import pytest
import torch

from conftest import nan_equal_close
from harness import classify_samples, make_oracle, make_product, to_np
from mofanerf_amd import factory, lib, rays, synth
from oracle import mofa_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
ARCH = (8, 64, 10, 64)


def _rays(H, angle=15.0):
    K = synth.intrinsics(H, H)
    ro, rd = orc.get_rays(H, H, K, orc.pose_spherical(angle, 0.0, 16.0)[:3, :4])
    return K, ro.reshape(-1, 3), rd.reshape(-1, 3)


def _codes():
    return [t.to(DEV) for t in synth.codes(0)]


@pytest.mark.parametrize("Ns,Ni", [(32, 16), (64, 0), (48, 80), (16, 128), (256, 256), (300, 212), (520, 0)])
def test_ragged_sample_counts_and_coarse_only(Ns, Ni):
    render, kw, _ = make_product(ARCH, 0, 4096, DEV, N_samples=Ns, N_importance=Ni)
    K, ro, rd = _rays(8)
    bm, tex, exp = _codes()
    with torch.no_grad():
        rgb, disp, acc, ex = render.render_fitting(8, 8, K, chunk=40, rays=torch.stack([ro, rd], 0).to(DEV), shapeCodes=bm,
                                                   uvCodes=tex, expType=20, expCodes=exp, verbose=True, **kw)
    o = make_oracle(ARCH, 0, 4096)
    with torch.no_grad():
        r_rgb, r_disp, r_acc, r_ex = o.render(ro, rd, 40, synth.codes(0)[0], 20, 8.0, 26.0, tex_code=synth.codes(0)[1],
                                              exp_codes=synth.codes(0)[2], N_samples=Ns, N_importance=Ni, keep=True)
    if Ni == 0:
        assert "rgb0" not in ex and set(ex) >= {"losses"}
        nan_equal_close(rgb.cpu().numpy(), r_rgb.numpy(), 1e-4)
        nan_equal_close(acc.cpu().numpy(), r_acc.numpy(), 1e-4)
        return
    nan_equal_close(ex["rgb0"].cpu().numpy(), r_ex["rgb0"].numpy(), 1e-4)
    d = r_ex["_dbg"]
    w_err = float((ex["_weights0"].cpu() - d["weights_coarse"]).abs().max())
    assert w_err < 2e-5
    agree, expl = classify_samples(d["z_coarse"], d["weights_coarse"], torch.linspace(0., 1., Ni), ex["_z_samples"].cpu(),
                                   d["z_samples"], w_err=w_err)
    assert (agree | expl).all()
    clean = agree.all(-1).numpy()
    assert clean.any()
    nan_equal_close(rgb.cpu().numpy()[clean], r_rgb.numpy()[clean], 1e-3)
    assert ex["_z_fine"].shape[-1] == Ns + Ni and (np.diff(ex["_z_fine"].cpu().numpy(), axis=-1) >= 0).all()


def test_lindisp_and_white_background():
    render, kw, _ = make_product(ARCH, 0, 4096, DEV)
    K, ro, rd = _rays(8)
    bm, tex, exp = _codes()
    kw = dict(kw, lindisp=True, white_bkgd=True)
    with torch.no_grad():
        rgb, disp, acc, ex = render.render_fitting(8, 8, K, chunk=64, rays=torch.stack([ro, rd], 0).to(DEV), shapeCodes=bm,
                                                   uvCodes=tex, expType=20, expCodes=exp, **kw)
    o = make_oracle(ARCH, 0, 4096)
    with torch.no_grad():
        _, _, _, r_ex = o.render(ro, rd, 64, synth.codes(0)[0], 20, 8.0, 26.0, tex_code=synth.codes(0)[1],
                                 exp_codes=synth.codes(0)[2], N_samples=64, N_importance=64, lindisp=True, white_bkgd=True)
    nan_equal_close(ex["rgb0"].cpu().numpy(), r_ex["rgb0"].numpy(), 1e-4)          # coarse pass: identical sample positions
    nan_equal_close(ex["acc0"].cpu().numpy(), r_ex["acc0"].numpy(), 1e-4)


def test_static_camera_view_directions():
    """c2w_staticcam: rays from the static camera, view directions from c2w (render_class.py:161-163)."""
    render, kw, _ = make_product(ARCH, 0, 4096, DEV)
    bm, tex, exp = _codes()
    K = synth.intrinsics(8, 8)
    a, b = orc.pose_spherical(40.0, 0.0, 16.0)[:3, :4], orc.pose_spherical(-10.0, 0.0, 16.0)[:3, :4]
    with torch.no_grad():
        out = render.render_fitting(8, 8, K, chunk=64, c2w=a, c2w_staticcam=b, shapeCodes=bm, uvCodes=tex, expType=20,
                                    expCodes=exp, **kw)
    ro, rd = orc.get_rays(8, 8, K, b)
    _, vd_src = orc.get_rays(8, 8, K, a)
    vd = (vd_src / torch.norm(vd_src, dim=-1, keepdim=True)).reshape(-1, 3)
    o = make_oracle(ARCH, 0, 4096)
    o.exp_sigma.append(synth.codes(0)[2])
    rays = torch.cat([ro.reshape(-1, 3), rd.reshape(-1, 3), 8 * torch.ones(64, 1), 26 * torch.ones(64, 1), vd], -1)
    with torch.no_grad():
        r = o.render_rays(rays, synth.codes(0)[0], synth.codes(0)[1], 20, 64, 64)
    nan_equal_close(out[3]["rgb0"].reshape(-1, 3).cpu().numpy(), r["rgb0"].numpy(), 1e-4)
    assert out[0].shape == (8, 8, 3)


@pytest.mark.parametrize("n_rays,netchunk", [(1, 4096), (7, 4096), (5, 50), (130, 64)])
def test_tiny_ray_counts_and_netchunk_below_one_ray(n_rays, netchunk):
    render, kw, _ = make_product(ARCH, 0, netchunk, DEV)
    K, ro, rd = _rays(16)
    bm, tex, exp = _codes()
    rays = torch.stack([ro[:n_rays], rd[:n_rays]], 0).to(DEV)
    with torch.no_grad():
        rgb, disp, acc, ex = render.render_fitting(16, 16, K, chunk=64, rays=rays, shapeCodes=bm, uvCodes=tex, expType=20,
                                                   expCodes=exp, **kw)
        ref = make_product(ARCH, 0, 1 << 20, DEV)
        rgb2, _, acc2, ex2 = ref[0].render_fitting(16, 16, K, chunk=1 << 20, rays=rays, shapeCodes=bm, uvCodes=tex, expType=20,
                                                   expCodes=exp, **ref[1])
    assert rgb.shape == (n_rays, 3) and disp.shape == (n_rays,)
    assert torch.equal(rgb, rgb2) and torch.equal(acc, acc2) and torch.equal(ex["rgb0"], ex2["rgb0"])   # split-invariant


def test_dataparallel_wrapped_networks_and_tuple_rays():
    render, kw, _ = make_product(ARCH, 0, 4096, DEV)
    K, ro, rd = _rays(8)
    bm, tex, exp = _codes()
    with torch.no_grad():
        a = render.render_fitting(8, 8, K, chunk=64, rays=(ro.to(DEV), rd.to(DEV)), shapeCodes=bm, uvCodes=tex, expType=20,
                                  expCodes=exp, **kw)
        kw2 = dict(kw, network_fn=torch.nn.DataParallel(kw["network_fn"]), network_fine=torch.nn.DataParallel(kw["network_fine"]))
        render.idSpecificMod = torch.nn.DataParallel(render.idSpecificMod)          # run_fit.py:166-168
        b = render.render_fitting(8, 8, K, chunk=64, rays=torch.stack([ro, rd], 0).to(DEV), shapeCodes=bm, uvCodes=tex,
                                  expType=20, expCodes=exp, **kw2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])


def test_stochastic_sampling_with_device_rng():
    """perturb=1 / raw_noise_std>0 without the pytest hook: draws come from torch's device RNG — outputs are finite, differ
    between calls, are reproducible under manual_seed, and the merged positions stay sorted inside [near, far]."""
    render, kw_test, kw_train = make_product(ARCH, 0, 4096, DEV)
    K, ro, rd = _rays(8)
    bm, tex, exp = _codes()
    kw = dict(kw_train, raw_noise_std=1.0)
    rays = torch.stack([ro, rd], 0).to(DEV)

    def go(seed):
        torch.manual_seed(seed)
        with torch.no_grad():
            return render.render_fitting(8, 8, K, chunk=64, rays=rays, shapeCodes=bm, uvCodes=tex, expType=20, expCodes=exp,
                                         verbose=True, **kw)
    a, b, c = go(1), go(2), go(1)
    assert torch.equal(a[0], c[0]) and not torch.equal(a[0], b[0])
    zf = a[3]["_z_fine"].cpu().numpy()
    assert np.isfinite(a[0].cpu().numpy()).all() and (np.diff(zf, axis=-1) >= 0).all() and zf.min() >= 8.0 and zf.max() <= 26.0


def test_render_path_writes_png_and_resumes(tmp_path):
    render, kw, _ = make_product(ARCH, 0, 4096, DEV, with_tex=True)
    rng = np.random.default_rng(0)
    uv = torch.from_numpy(rng.uniform(0, 1, (1, 512, 512, 3)).astype(np.float32)).to(DEV)
    poses = torch.stack([orc.pose_spherical(a, 0.0, 16.0) for a in (-20.0,)], 0)
    K = synth.intrinsics(16, 16)
    bm = synth.codes(0)[0].to(DEV)
    with torch.no_grad():
        rgbs, disps = render.render_path(poses, [16, 16, float(K[0][0])], K, 4096, kw, uvMap=uv,
                                         expType=torch.tensor([5]), savedir=str(tmp_path), shapeCodes=bm, name="000_05_0")
        assert rgbs.shape == (1, 16, 16, 3) and disps.shape == (1, 16, 16)
        f = tmp_path / "000_05_0.png"
        data = f.read_bytes()
        assert data[:8] == b"\x89PNG\r\n\x1a\n" and len(data) > 100
        again = render.render_path(poses, [16, 16, float(K[0][0])], K, 4096, kw, uvMap=uv, expType=torch.tensor([5]),
                                   savedir=str(tmp_path), shapeCodes=bm, name="000_05_0")
    assert again == (0, 0)                                                        # existing output is skipped (resume)
    # the texture code is cached per UV map for render-only calls
    assert render._tex_cache is not None


def test_render_path_multi_pose_render_factor_and_shared_sink(tmp_path):
    """Several poses, numbered outputs, `render_factor` (render_class.py:205-209: H, W, focal divided, K untouched), and one
    asynchronous sink shared across calls: the files hold exactly to8b of the returned frames."""
    from PIL import Image
    from mofanerf_amd.io import PngSink
    render, kw, _ = make_product(ARCH, 0, 4096, DEV, with_tex=True)
    rng = np.random.default_rng(1)
    uv = torch.from_numpy(rng.uniform(0, 1, (1, 512, 512, 3)).astype(np.float32)).to(DEV).expand(3, -1, -1, -1)
    poses = torch.stack([orc.pose_spherical(a, 0.0, 16.0) for a in (-30.0, 0.0, 30.0)], 0)
    K = synth.intrinsics(16, 16)
    bm = synth.codes(0)[0].to(DEV).expand(3, -1)
    render.png_sink = PngSink(workers=2)
    with torch.no_grad():
        rgbs, disps = render.render_path(poses, [32, 32, 2 * float(K[0][0])], K, 4096, kw, uvMap=uv,
                                         expType=torch.tensor([1, 5, 9]), savedir=str(tmp_path), shapeCodes=bm, render_factor=2)
    render.png_sink.close()
    render.png_sink = None
    assert rgbs.shape == (3, 16, 16, 3) and disps.shape == (3, 16, 16)
    for i in range(3):
        got = np.asarray(Image.open(tmp_path / f"{i:03d}.png").convert("RGB"))
        assert np.array_equal(got, (255 * np.clip(rgbs[i], 0, 1)).astype(np.uint8))
    assert not np.array_equal(rgbs[0], rgbs[2])


def test_unsupported_flags_fail_loudly():
    render, kw, _ = make_product(ARCH, 0, 4096, DEV)
    K, ro, rd = _rays(8)
    bm, tex, exp = _codes()
    rays = torch.stack([ro, rd], 0).to(DEV)
    # use_viewdirs=False: the reference's own branch cannot run (NeRF.forward needs alpha_linear / rgb_linear) -> rejected with the reason
    with pytest.raises(NotImplementedError, match="alpha_linear"), torch.no_grad():
        render.render_fitting(8, 8, K, chunk=64, rays=rays, shapeCodes=bm, uvCodes=tex, expType=20, expCodes=exp,
                              **dict(kw, use_viewdirs=False))
    with pytest.raises(RuntimeError, match="inference-only"):      # NeRF.forward (embedded inputs) refuses to drop gradients silently
        kw["network_fn"](torch.zeros(1, 93, device=DEV), torch.zeros(1, 50, device=DEV), torch.zeros(1, 27, device=DEV),
                         torch.zeros(1, 256, device=DEV))
    with pytest.raises(lib.MofaError, match="S >= 2"):            # argument checks return before anything is launched
        lib.check(lib.load().mofa_composite_forward(1, 1, 0, 1, None, 4, 1, 0, 1, 1, 1, 1, 1, None), "composite S=1")
    with pytest.raises(lib.MofaError, match="3 S \\+ Ni"):        # one ray's positions, bins and cdf must fit 64 KiB of LDS
        lib.check(lib.load().mofa_sample_pdf_merge(1, 0, 1, 1, 0, 4, 5000, 5000, 1, 1, 1, None), "sample_pdf_merge S=Ni=5000")


def test_bulk_render_tool_and_ray_helpers(tmp_path):
    """Config-4 job shape at toy size (2 identities x 1 expression x 2 views, 16x16, small nets) + resume; rays helpers."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bulk_render", os.path.join(root, "tools", "bulk_render.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = ["--out", str(tmp_path), "--identities", "2", "--expressions", "1", "--views", "2", "--size", "16", "--arch", "8", "64",
            "10", "64"]
    assert mod.main(argv) == 4
    assert sorted(os.listdir(tmp_path / "000")) == ["00_0.png", "00_1.png"]
    assert mod.main(argv) == 0                                        # everything already on disk: nothing re-rendered
    from mofanerf_amd import rays
    K = synth.intrinsics(32, 32)
    c2w = rays.pose_spherical(25.0, 0.0, 16.0)
    assert torch.equal(c2w, orc.pose_spherical(25.0, 0.0, 16.0))
    ro, rd = rays.get_rays(32, 32, K, c2w[:3, :4], device=DEV)
    ro_ref, rd_ref = orc.get_rays(32, 32, K, c2w[:3, :4])
    assert torch.equal(rd.cpu(), rd_ref) and torch.equal(ro.cpu(), ro_ref.contiguous())
    pose = c2w[:3, :4].clone().to(DEV).requires_grad_(True)
    rows, cols = torch.tensor([0, 5, 31], device=DEV), torch.tensor([3, 17, 31], device=DEV)
    r = rays.rays_at_pixels(K, pose, rows, cols)
    assert torch.allclose(r[1].detach().cpu(), rd_ref[rows.cpu(), cols.cpu()], atol=1e-6)
    r[1].sum().backward()
    assert pose.grad is not None and float(pose.grad.abs().sum()) > 0


def test_no_reduced_precision_mode_is_reachable(monkeypatch):
    """The library is exact fp32 and nothing else: a leftover MOFA_GEMM setting (the removed split-product experiment) is refused
    loudly instead of being ignored, and the ABI carries no split entry point."""
    L = lib.load()
    assert not any(hasattr(L, n) for n in ("mofa_layer_forward_split", "mofa_pack_split", "mofa_net_pack_split", "mofa_net_packed_split_elems"))
    for mode in ("bf16x3", "bf16x6", "fp16x3"):
        monkeypatch.setenv("MOFA_GEMM", mode)
        with pytest.raises(lib.MofaError, match="exact fp32 only"):
            lib.load()
        with pytest.raises(lib.MofaError, match="exact fp32 only"):
            lib.reload_env()
    monkeypatch.setenv("MOFA_GEMM", "fp32")
    lib.load()


def test_nan_input_propagates_like_the_reference():
    """A NaN ray (bad pose / bad data) yields NaN outputs for that ray in the reference (torch's relu, sigmoid, cumprod all
    propagate NaN); the HIP path must not turn it into a plausible pixel, and must leave the other rays untouched."""
    render, kw, _ = make_product(ARCH, 0, 4096, DEV)
    K, ro, rd = _rays(4)
    ro = ro.clone()
    ro[5, 1] = float("nan")
    bm, tex, exp = _codes()
    with torch.no_grad():
        rgb, disp, acc, ex = render.render_fitting(4, 4, K, chunk=16, rays=torch.stack([ro, rd], 0).to(DEV), shapeCodes=bm,
                                                   uvCodes=tex, expType=20, expCodes=exp, **kw)
    o = make_oracle(ARCH, 0, 4096)
    with torch.no_grad():
        r_rgb, r_disp, r_acc, r_ex = o.render(ro, rd, 16, synth.codes(0)[0], 20, 8.0, 26.0, tex_code=synth.codes(0)[1],
                                              exp_codes=synth.codes(0)[2], N_samples=64, N_importance=64)
    assert torch.isnan(r_rgb[5]).all() and torch.isnan(r_ex["rgb0"][5]).all()       # what the reference does
    for got, want in ((ex["rgb0"], r_ex["rgb0"]), (ex["acc0"], r_ex["acc0"])):
        nan_equal_close(got.cpu().numpy(), want.numpy(), 1e-4)                      # same NaN pattern, same finite values
    for got, want in ((rgb, r_rgb), (acc, r_acc)):
        g, w = got.cpu().numpy(), want.numpy()
        assert (np.isnan(g) == np.isnan(w)).all()
    assert torch.isnan(rgb[5]).all() and torch.isnan(acc[5]) and torch.isfinite(rgb[:5]).all() and torch.isfinite(rgb[6:]).all()


def test_ndc_rays_forward_facing_flag(golden):
    """`render(..., ndc=True)` (models/render_class.py:166-169; the signature's default): rays are mapped to normalised device
    coordinates by ndc_rays (tools/run_nerf_helpers.py:182-200) after the view directions were taken, near / far = 0 / 1.
    The mapped rays equal the reference's KAT; the render equals the oracle on the same NDC rays."""
    from mofanerf_amd import rays as mrays
    from harness import make_oracle
    g = golden("kat_ndc.npz")
    no, nd = mrays.ndc_rays(16, 16, float(g["K"][0][0]), 1., torch.from_numpy(g["rays_o"]).to(DEV), torch.from_numpy(g["rays_d"]).to(DEV))
    nan_equal_close(no.cpu().numpy(), g["ndc_o"], 1e-6, 1e-6)
    nan_equal_close(nd.cpu().numpy(), g["ndc_d"], 1e-6, 1e-6)
    render, kw, _ = make_product(ARCH, 0, 4096, DEV)
    bm, tex, exp = _codes()
    kw = dict(kw, ndc=True, near=0.0, far=1.0)
    with torch.no_grad():
        rgb, disp, acc, ex = render.render_fitting(16, 16, g["K"], chunk=100, c2w=torch.from_numpy(g["c2w"]), shapeCodes=bm, uvCodes=tex,
                                                   expType=20, expCodes=exp, **kw)
    assert rgb.shape == (16, 16, 3)
    o = make_oracle(ARCH, 0, 4096)
    o.exp_sigma.append(synth.codes(0)[2])
    ro, rd = torch.from_numpy(g["rays_o"]).reshape(-1, 3), torch.from_numpy(g["rays_d"]).reshape(-1, 3)
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    r11 = torch.cat([torch.from_numpy(g["ndc_o"]).reshape(-1, 3), torch.from_numpy(g["ndc_d"]).reshape(-1, 3), torch.zeros(256, 1),
                     torch.ones(256, 1), vd], -1)
    with torch.no_grad():
        ref = o.render_rays(r11, synth.codes(0)[0], synth.codes(0)[1], 20, 64, 64)
    nan_equal_close(ex["rgb0"].reshape(-1, 3).cpu().numpy(), ref["rgb0"].numpy(), 1e-4)
    nan_equal_close(ex["acc0"].reshape(-1).cpu().numpy(), ref["acc0"].numpy(), 1e-4)
