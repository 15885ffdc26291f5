"""(f2) Ray generation folded into the path: pixel-list ray generation, the camera-pose gradient and layer 0's camera mode,
against the reference's own fixtures (get_rays KAT; pose gradients of a run_fit.py-style step, tests/golden/grads_pose.npz)."""
import numpy as np
import pytest
import torch

from harness import make_product
from mofanerf_amd import lib, rays, synth
from oracle import mofa_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = torch.from_numpy


def test_get_rays_at_pixel_list_is_bit_exact(golden):
    """`mofa_get_rays_at` on the pixel list (::37, ::41) of a 256x256 view reproduces the reference's directions bit for bit
    (fixture rays*_d_sub), and equals the full-frame kernel at those pixels."""
    g = golden("kat.npz")
    L = lib.load()
    rows, cols = np.meshgrid(np.arange(0, 256, 37), np.arange(0, 256, 41), indexing="ij")
    pix = torch.from_numpy((rows * 256 + cols).reshape(-1).astype(np.int32)).to(DEV)
    n = pix.numel()
    for ang in (-60, 0, 60):
        c2w = T(g[f"rays{ang}_c2w"][:3, :4]).contiguous().to(DEV)
        o, d, v = (torch.empty(n, 3, device=DEV) for _ in range(3))
        lib.check(L.mofa_get_rays_at(256, 256, 600., 600., 128., 128., lib.ptr(c2w), pix.data_ptr(), n, lib.ptr(o), lib.ptr(d),
                                     lib.ptr(v), lib.stream()), "get_rays_at")
        assert np.array_equal(d.cpu().numpy().reshape(rows.shape + (3,)), g[f"rays{ang}_d_sub"])
        assert np.array_equal(o[0].cpu().numpy(), g[f"rays{ang}_o"])
        of, df, vf = (torch.empty(256 * 256, 3, device=DEV) for _ in range(3))
        lib.check(L.mofa_get_rays(256, 256, 600., 600., 128., 128., lib.ptr(c2w), 0, 256 * 256, lib.ptr(of), lib.ptr(df),
                                  lib.ptr(vf), lib.stream()), "get_rays")
        assert torch.equal(df[pix.long()], d) and torch.equal(vf[pix.long()], v) and torch.equal(of[pix.long()], o)


def test_layer0_camera_mode_is_bit_identical_to_materialised_rays():
    """Layer 0 with the ray built in the prologue from (K, c2w, pixel) == get_rays_at + layer 0 on the materialised rays: pixel
    list and contiguous pixel range, row z and per-ray z."""
    L = lib.load()
    rng = np.random.default_rng(31)
    W_, S, Wn = 64, 48, 128
    c2w = orc.pose_spherical(-35.0, 0.0, 16.0)[:3, :4].contiguous().to(DEV)
    K = synth.intrinsics(W_, W_)
    fx, fy, cx, cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    w = T((rng.normal(size=(Wn, 64)) / 8).astype(np.float32)).to(DEV)
    wp = torch.empty(Wn * 64, device=DEV)
    lib.check(L.mofa_pack_panels(lib.ptr(w), Wn, 64, 0, 63, lib.ptr(wp), Wn, 0, 64, lib.stream()), "pack")
    bias = T(rng.normal(size=(Wn,)).astype(np.float32)).to(DEV)
    for pix, pix0, n in ((torch.from_numpy(rng.choice(W_ * W_, 300, replace=False).astype(np.int32)).to(DEV), 0, 300), (None, 1000, 517)):
        for zs in (0, S):
            z = T(np.sort(rng.uniform(8, 26, (n if zs else 1, S)).astype(np.float32), -1)).contiguous().to(DEV)
            M = n * S
            Mp = (M + 255) // 256 * 256
            o, d = torch.empty(n, 3, device=DEV), torch.empty(n, 3, device=DEV)
            if pix is not None:
                lib.check(L.mofa_get_rays_at(W_, W_, fx, fy, cx, cy, lib.ptr(c2w), pix.data_ptr(), n, lib.ptr(o), lib.ptr(d), None,
                                             lib.stream()), "get_rays_at")
            else:
                lib.check(L.mofa_get_rays(W_, W_, fx, fy, cx, cy, lib.ptr(c2w), pix0, n, lib.ptr(o), lib.ptr(d), None, lib.stream()),
                          "get_rays")
            ya, yb = torch.full((Mp * Wn,), float("nan"), device=DEV), torch.full((Mp * Wn,), float("nan"), device=DEV)
            lib.check(L.mofa_layer0_forward(lib.ptr(o), lib.ptr(d), lib.ptr(z), zs, None, M, S, 10, lib.ptr(wp), lib.ptr(bias), lib.ptr(ya),
                                            Mp, Wn, None, lib.stream()), "layer0")
            lib.check(L.mofa_layer0_forward_cam(W_, fx, fy, cx, cy, lib.ptr(c2w), None if pix is None else pix.data_ptr(), pix0,
                                                lib.ptr(z), zs, M, S, 10, lib.ptr(wp), lib.ptr(bias), lib.ptr(yb), Mp, Wn, lib.stream()),
                      "layer0_cam")
            torch.cuda.synchronize()
            a, b = torch.empty(M, Wn, device=DEV), torch.empty(M, Wn, device=DEV)
            lib.check(L.mofa_from_panels(lib.ptr(ya), Mp, M, Wn, lib.ptr(a), lib.stream()), "from_panels")
            lib.check(L.mofa_from_panels(lib.ptr(yb), Mp, M, Wn, lib.ptr(b), lib.stream()), "from_panels")
            assert torch.equal(a, b) and bool(torch.isfinite(a).all()) and float(a.abs().sum()) > 0


def test_pose_gradient_vs_reference_fixture(golden):
    """The camera-pose gradient of a run_fit.py-style step (fixture g12: get_rays on a pose that requires grad, 96 gathered
    pixels, render_fitting, L1 + coarse loss).  (i) teacher-forced: the reference's per-ray gradients through
    `mofa_rays_pose_backward` give the reference's d loss / d c2w to fp32 rounding; (ii) end to end through the HIP path
    (`rays_at_pixels` -> render_fitting -> backward): same direction (cosine > 0.99; sample positions legitimately differ)."""
    g = golden("grads_pose.npz")
    H = int(g["H"])
    K = g["K"]
    L = lib.load()
    pix = T(g["pix"]).to(DEV)
    n = pix.numel()
    d_pose = torch.empty(3, 4, device=DEV)
    go, gd = T(g["g_rays_o"]).contiguous().to(DEV), T(g["g_rays_d"]).contiguous().to(DEV)
    lib.check(L.mofa_rays_pose_backward(H, float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]), pix.data_ptr(), 0, n,
                                        lib.ptr(go), lib.ptr(gd), lib.ptr(d_pose), lib.stream()), "rays_pose_backward")
    ref = g["g_c2w"].astype(np.float64)
    err = np.abs(d_pose.cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err
    # end to end
    render, kw, _ = make_product((8, 64, 10, 64), 0, 4096, DEV)
    c2w = T(g["c2w"]).to(DEV).requires_grad_(True)
    batch = rays.rays_at_pixels(K, c2w, (pix // H).long(), (pix % H).long(), H, H)
    assert np.array_equal(batch[1].detach().cpu().numpy(), g["rays_d"]) and np.array_equal(batch[0].detach().cpu().numpy(), g["rays_o"])
    bm, tex, exp = [T(g[k]).to(DEV) for k in ("bm", "tex", "exp")]
    rgb, _, _, ex = render.render_fitting(H, H, K, chunk=96, rays=batch, shapeCodes=bm.expand(n, 50), uvCodes=tex, expType=20,
                                          expCodes=exp, **kw)
    loss = torch.nn.functional.l1_loss(rgb, T(g["target"]).to(DEV)) + (ex["rgb0"] ** 2).mean()
    loss.backward()
    torch.cuda.synchronize()
    a = c2w.grad.cpu().numpy().ravel().astype(np.float64)
    cos = float(a @ ref.ravel() / (np.linalg.norm(a) * np.linalg.norm(ref) + 1e-30))
    print(f"pose gradient: teacher-forced rel err {err:.1e}; end-to-end cosine {cos:.5f}, loss {float(loss.detach()):.6f} vs {float(g['loss']):.6f}")
    assert cos > 0.99 and abs(float(loss.detach()) - float(g["loss"])) < 1e-4
