"""End-to-end GPU parity of the drop-in renderer (render_fitting / render / run_network) against the committed golden
fixtures (outputs of the reference itself) and the CPU oracle.  Tolerance: the north star's fp32 budget of
1e-4 max-abs on RGB / acc; disp (= 1/depth-like, up to 0.125) relative 1e-4 and NaN-pattern-exact."""
import numpy as np
import pytest
import torch

from conftest import nan_equal_close
from harness import make_oracle, make_product, render_pair, to_np
from mofanerf_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = torch.from_numpy


def _fit(g, **over):
    arch = tuple(int(v) for v in g["arch"])
    render, kw, kw_train = make_product(arch, int(g["seed"]), int(g["netchunk"]), DEV)
    kw = dict(kw)
    kw.update(over)
    H = int(g["H"])
    out = render.render_fitting(H, H, g["K"], chunk=int(g["chunk"]), c2w=T(g["c2w"]), shapeCodes=T(g["bm"]).to(DEV),
                                uvCodes=T(g["tex"]).to(DEV), expType=20, expCodes=T(g["exp"]).to(DEV), retraw=True,
                                verbose=True, **kw)
    torch.cuda.synchronize()
    return out


def _check(g, out, tol=1e-4):
    rgb, disp, acc, ex = out
    H = int(g["H"])
    assert rgb.shape == (H, H, 3) and disp.shape == (H, H) and ex["rgb0"].shape == (H, H, 3)
    errs = {}
    errs["rgb"] = nan_equal_close(rgb.cpu().numpy(), g["rgb"], tol)
    errs["acc"] = nan_equal_close(acc.cpu().numpy(), g["acc"], tol)
    errs["rgb0"] = nan_equal_close(ex["rgb0"].cpu().numpy(), g["rgb0"], tol)
    errs["acc0"] = nan_equal_close(ex["acc0"].cpu().numpy(), g["acc0"], tol)
    errs["disp"] = nan_equal_close(disp.cpu().numpy(), g["disp"], 1e-6, 1e-4)
    errs["disp0"] = nan_equal_close(ex["disp0"].cpu().numpy(), g["disp0"], 1e-6, 1e-4)
    errs["z_std"] = nan_equal_close(ex["z_std"].cpu().numpy(), g["z_std"], 1e-3)
    assert ex["losses"] == 0
    print({k: f"{v:.2e}" for k, v in errs.items()})
    return errs


def test_render_fitting_small_golden(golden):
    """256 rays, coarse 8x64 + fine 10x128, chunk 96 (3 ragged chunks), rays generated on the device."""
    g = golden("e2e_small.npz")
    out = _fit(g)
    _check(g, out)
    ex = out[3]
    R = int(g["H"]) ** 2
    # intermediates: coarse weights, new samples (conditioning-aware), merged z
    nan_equal_close(ex["_weights0"].reshape(R, 64).cpu().numpy(), g["weights_coarse"], 1e-5)
    zs, zs_ref = ex["_z_samples"].reshape(R, 64).cpu().numpy(), g["z_samples"]
    assert np.median(np.abs(zs - zs_ref)) <= 4e-6
    assert (np.abs(zs - zs_ref) <= 5e-2).all()
    zf = ex["_z_fine"].reshape(R, 128).cpu().numpy()
    assert (np.diff(zf, axis=-1) >= 0).all()


def test_render_fitting_true_size_golden(golden):
    """64 rays through the SHIPPED network sizes (coarse 256x8, fine 1024x10)."""
    g = golden("e2e_true.npz")
    _check(g, _fit(g))


def test_render_fitting_stochastic_golden(golden):
    """perturb=1, raw_noise_std=0.5, white_bkgd, pytest=True (seed-0 numpy randoms, as the reference's hook)."""
    g = golden("e2e_small_stoch.npz")
    _check(g, _fit(g, perturb=1.0, raw_noise_std=float(g["noise"]), white_bkgd=True, pytest=True))


def test_render_with_texture_encoder_golden(golden):
    g = golden("render_tex.npz")
    render, kw, _ = make_product((8, 64, 10, 64), 0, 4096, DEV, with_tex=True)
    uv = T(np.random.default_rng(5).uniform(0, 1, (512, 512, 3)).astype(np.float32)).to(DEV)
    rays = T(g["rays"]).to(DEV)
    with torch.no_grad():
        rgb, disp, acc, ex = render.render(8, 8, None, chunk=64, rays=rays, shapeCodes=T(g["bm"]).expand(64, 50).to(DEV),
                                           uvMap=uv, expType=7, retraw=True, **kw)
    nan_equal_close(render.decoding_texCodes.cpu().numpy(), g["tex_code"], 2e-5, 1e-4)
    nan_equal_close(rgb.cpu().numpy(), g["rgb"], 1e-4)
    nan_equal_close(acc.cpu().numpy(), g["acc"], 1e-4)
    nan_equal_close(ex["raw"].cpu().numpy(), g["raw"], 2e-3, 1e-3)
    assert ex["losses"] == 0 and rgb.shape == (64, 3)


def test_run_network_api():
    """run_network(inputs[R,S,3], viewdirs[R,3], fn) — the reference's network_query_fn."""
    from oracle import mofa_oracle as orc
    render, kw, _ = make_product((8, 64, 10, 64), 0, 1000, DEV)
    o = make_oracle((8, 64, 10, 64), 0, 1000)
    rng = np.random.default_rng(8)
    pts = T(rng.uniform(-8, 8, (19, 33, 3)).astype(np.float32))
    vd = torch.nn.functional.normalize(T(rng.normal(size=(19, 3)).astype(np.float32)), dim=-1)
    bm, tex, exp = synth.codes(0)
    render.shapeCodes, render.expType, render.decoding_texCodes = bm.to(DEV), 3, tex.to(DEV)
    raw = kw["network_query_fn"](pts.to(DEV), vd.to(DEV), kw["network_fine"])
    ref = o.run_network(pts, vd, o.fine, bm, tex, 3)
    nan_equal_close(raw.cpu().numpy(), ref.numpy(), 2e-5, 1e-5)


def test_chunk_and_netchunk_invariance_shipped_sizes():
    """'chunk ... Does not affect final results' (render_class.py:136-137): 300 rays through the shipped sizes with
    different chunk/netchunk splits must agree BIT-exactly (every ray's arithmetic is independent of its tile)."""
    arch = (8, 256, 10, 1024)
    K = synth.intrinsics(32, 32)
    outs = []
    for chunk, netchunk in ((300, 196608), (128, 8192), (77, 5000)):
        render, kw, _ = make_product(arch, 0, netchunk, DEV)
        bm, tex, exp = synth.codes(0)
        from oracle import mofa_oracle as orc
        ro, rd = orc.get_rays(32, 32, K, orc.pose_spherical(-60.0, 0.0, 16.0)[:3, :4])
        rays = torch.stack([ro.reshape(-1, 3)[:300], rd.reshape(-1, 3)[:300]], 0).to(DEV)
        rgb, disp, acc, ex = render.render_fitting(32, 32, K, chunk=chunk, rays=rays, shapeCodes=bm.to(DEV),
                                                   uvCodes=tex.to(DEV), expType=20, expCodes=exp.to(DEV), **kw)
        outs.append(to_np(dict(rgb=rgb, acc=acc, rgb0=ex["rgb0"], z_std=ex["z_std"])))
    for o in outs[1:]:
        for k in o:
            assert np.array_equal(o[k], outs[0][k], equal_nan=True), k
    a = outs[0]["acc"]
    assert (a >= 0).all() and (a <= 1 + 1e-5).all() and np.isfinite(outs[0]["rgb"]).all()


def test_device_pair_vs_oracle_medium():
    """1024 rays (32x32 view), coarse 8x128 + fine 10x256: HIP vs oracle on identical rays."""
    hip, ref = render_pair(32, synth.intrinsics(32, 32), 60.0, (8, 128, 10, 256), chunk=400, netchunk=20000, device=DEV)
    for k, tol in (("rgb", 1e-4), ("acc", 1e-4), ("rgb0", 1e-4), ("acc0", 1e-4)):
        print(k, nan_equal_close(hip[k], ref[k], tol))
    nan_equal_close(hip["disp"], ref["disp"], 1e-6, 1e-4)


def test_cpu_tensors_fail_loudly():
    from mofanerf_amd import lib
    render, kw, _ = make_product((8, 64, 10, 64), 0, 4096, "cpu")
    bm, tex, exp = synth.codes(0)
    with pytest.raises(lib.MofaError):
        render.render_fitting(8, 8, synth.intrinsics(8, 8), chunk=64, c2w=torch.eye(4)[:3], shapeCodes=bm, uvCodes=tex,
                              expType=20, expCodes=exp, **kw)
