"""Networks built from NON-SHIPPED reference flags — multires, multires_views, i_embed, input_ch_shapeCodes / textureCodes / expCodes
(tools/config_parser.py:51-56,113-118; tools/create_model_condition.py:16-34) — on the HIP path, against fixtures written by the
reference itself (tests/golden/make_golden.py::g16_flags).  The C plan carries these widths in MofaNetShape (ABI 2); before that
they were compile-time constants and a module built from other flags was silently mis-read.

Tolerance: 1e-4 max-abs on raw / RGB / acc (the north star's fp32 budget), gradients relative to the reference's fp32 autograd."""
import numpy as np
import pytest
import torch

from conftest import nan_equal_close
from harness import classify_samples
from mofanerf_amd import factory, lib, synth
from mofanerf_amd.autograd import NetFn, fold_torch, view_bias_torch
from mofanerf_amd.hipnet import HipNet
from mofanerf_amd.model import NeRF
from test_oracle_golden import FLAG_TAGS, flags_net_state

pytestmark = pytest.mark.gpu
DEV = "cuda"
T = torch.from_numpy


def _net(g, tag):
    (mr, mv, ce, cs, ct, D, W), st = flags_net_state(g, tag)
    net = NeRF(D=D, W=W, input_ch=3 + 6 * mr + ce, input_ch_views=3 + 6 * mv, input_ch_textureCodes=ct, input_ch_shapeCodes=cs,
               use_viewdirs=True)
    net.load_state_dict(st)
    return (mr, mv, ce, cs, ct, D, W), net.to(DEV)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("fused", ["0", "1"])
@pytest.mark.parametrize("tag", FLAG_TAGS)
def test_folded_path_forward_at_non_shipped_flags(golden, knob, tag, fused):
    """mofa_net_pack + mofa_net_fold + mofa_net_forward (explicit points) with the plan built from the module's own widths — per-layer
    launches and the persistent kernel — against the reference's run_network arithmetic."""
    g = golden("kat_flags.npz")
    (mr, mv, ce, cs, ct, D, W), net = _net(g, tag)
    knob("MOFA_FUSED", fused)
    h = HipNet(net, point_freqs=mr)
    assert (h.shape.pe_point_freqs, h.shape.pe_view_freqs, h.shape.ch_exp, h.shape.ch_shape, h.shape.ch_tex) == (mr, mv, ce, cs, ct)
    pts, vd = T(g[f"{tag}_pts"]).to(DEV), T(g[f"{tag}_vd"]).to(DEV)
    R, S = pts.shape[:2]
    folded = h.fold(T(g[f"{tag}_e"]).to(DEV), T(g[f"{tag}_bm"]).to(DEV), T(g[f"{tag}_tex"]).to(DEV))
    raw = torch.full((R, S, 4), float("nan"), device=DEV)
    h.forward_points(pts.reshape(-1, 3).contiguous(), vd.contiguous(), S, raw, folded)
    torch.cuda.synchronize()
    err = nan_equal_close(raw.cpu().numpy(), g[f"{tag}_raw"], 1e-4)
    print(f"{tag} fused={fused}: raw max abs err {err:.2e}")
    assert err < 2e-5


@pytest.mark.parametrize("tag", FLAG_TAGS)
def test_embedded_module_call_at_non_shipped_flags(golden, tag):
    """``NeRF.forward`` on already-embedded inputs (the reference's eager batchify form) for the same modules."""
    from oracle import mofa_oracle as orc
    g = golden("kat_flags.npz")
    (mr, mv, ce, cs, ct, D, W), net = _net(g, tag)
    pts, vd = T(g[f"{tag}_pts"]), T(g[f"{tag}_vd"])
    R, S = pts.shape[:2]
    n = R * S
    emb = torch.cat([orc.positional_encode(pts.reshape(-1, 3), mr), T(g[f"{tag}_e"]).expand(n, -1)], -1)
    dirs = orc.positional_encode(vd[:, None].expand(R, S, 3).reshape(-1, 3), mv)
    with torch.no_grad():
        raw = net(emb.to(DEV), T(g[f"{tag}_bm"]).expand(n, -1).to(DEV), dirs.to(DEV), T(g[f"{tag}_tex"]).expand(n, -1).to(DEV))
    nan_equal_close(raw.reshape(R, S, 4).cpu().numpy(), g[f"{tag}_raw"], 2e-5)
    with pytest.raises(lib.MofaError, match="do not match"):
        with torch.no_grad():
            net(emb[:, :-1].to(DEV), T(g[f"{tag}_bm"]).expand(n, -1).to(DEV), dirs.to(DEV), T(g[f"{tag}_tex"]).expand(n, -1).to(DEV))


@pytest.mark.parametrize("mode", ["mask", "tape", "train"])
@pytest.mark.parametrize("tag", FLAG_TAGS)
def test_backward_to_points_dirs_and_codes_at_non_shipped_flags(golden, tag, mode):
    """The explicit-point backward (mofa_net_backward with pts / d_pts — what a differentiable run_network needs) at the same flag
    sets: d(sum raw*G) / d{pts, viewdirs, exp, shape, tex} against the reference's autograd, with the mask-only tape (fitting
    default), the fp32 tape, and the training form (weight gradients requested)."""
    g = golden("kat_flags.npz")
    (mr, mv, ce, cs, ct, D, W), net = _net(g, tag)
    h = HipNet(net, point_freqs=mr)
    h.force_fp32_tape = mode == "tape"
    leaf = lambda k: T(g[f"{tag}_{k}"]).to(DEV).requires_grad_(True)
    pts, vd, e, bm, tex = leaf("pts"), leaf("vd"), leaf("e"), leaf("bm"), leaf("tex")
    R, S = pts.shape[:2]
    wts = [l.weight for l in h._linears] if mode == "train" else []
    raw = NetFn.apply(h, None, None, None, 0, S, fold_torch(h, e, bm, tex, detach_params=mode != "train"),
                      view_bias_torch(h, vd, detach_params=mode != "train"), pts.reshape(-1, 3), *wts)
    (raw * T(g[f"{tag}_G"]).to(DEV)).sum().backward()
    torch.cuda.synchronize()
    nan_equal_close(raw.detach().cpu().numpy(), g[f"{tag}_raw"], 2e-5)
    errs = {k: rel(v.grad.cpu().numpy(), g[f"{tag}_g_{k}"]) for k, v in (("pts", pts), ("vd", vd), ("e", e), ("bm", bm), ("tex", tex))}
    print(tag, mode, {k: f"{v:.1e}" for k, v in errs.items()})
    for k, v in errs.items():       # d/dx sin(2^(L-1) x) amplifies fp32 activation rounding in the point gradient
        assert v < (5e-3 if k == "pts" else 5e-4), (k, v)
    if mode == "train":
        assert all(l.weight.grad is not None and torch.isfinite(l.weight.grad).all() for l in h._linears)


def test_mask_tape_gradients_are_bit_identical_to_the_fp32_tape(golden):
    """One bit per activation is ALL the fitting backward reads of the forward: every gradient equals the fp32-tape run bit for bit."""
    g = golden("kat_flags.npz")
    for tag in FLAG_TAGS:
        (mr, mv, ce, cs, ct, D, W), net = _net(g, tag)
        outs = {}
        for mode in ("mask", "tape"):
            h = HipNet(net, point_freqs=mr)
            h.force_fp32_tape = mode == "tape"
            leaf = lambda k: T(g[f"{tag}_{k}"]).to(DEV).requires_grad_(True)
            pts, vd, e, bm, tex = leaf("pts"), leaf("vd"), leaf("e"), leaf("bm"), leaf("tex")
            S = pts.shape[1]
            raw = NetFn.apply(h, None, None, None, 0, S, fold_torch(h, e, bm, tex, detach_params=True),
                              view_bias_torch(h, vd, detach_params=True), pts.reshape(-1, 3))
            (raw * T(g[f"{tag}_G"]).to(DEV)).sum().backward()
            outs[mode] = [raw.detach().clone()] + [t.grad.clone() for t in (pts, vd, e, bm, tex)]
        for a, b in zip(outs["mask"], outs["tape"]):
            assert torch.equal(a, b)


def _flags_product(g, netchunk=None):
    mr, mv, ct = int(g["multires"]), int(g["multires_views"]), int(g["ch_tex"])
    Dc, Wc, Df, Wf = [int(v) for v in g["arch"]]
    args = factory.default_args(netdepth=Dc, netwidth=Wc, netdepth_fine=Df, netwidth_fine=Wf, multires=mr, multires_views=mv,
                                input_ch_textureCodes=ct, netchunk=netchunk or int(g["netchunk"]), no_reload=True, device=DEV,
                                basedir="/nonexistent")
    _, kw, _, _, _, _, render = factory.create_nerf(args)
    w = dict(ch_pts=3 + 6 * mr + 30, ch_shape=50, ch_tex=ct, ch_views=3 + 6 * mv)
    kw["network_fn"].load_state_dict(synth.nerf_state(Dc, Wc, 16, "coarse", **w))
    kw["network_fine"].load_state_dict(synth.nerf_state(Df, Wf, 16, "fine", **w))
    render.idSpecificMod.load_state_dict(synth.style_state(0))
    for dst, src in zip(render.expCodes_Sigma, synth.exp_sigma(0)):
        dst.data[:] = src.to(dst.device)
    kw.update(near=8.0, far=26.0)
    return render.eval(), kw


def test_render_fitting_end_to_end_at_non_shipped_flags(golden):
    """create_nerf(args) with multires=6, multires_views=2, input_ch_textureCodes=128 -> render_fitting: coarse pass 1e-4 on every ray;
    every resampled position agreeing with the reference's or explained by its 1e-5 branch (tests/harness.py); RGB / acc 1e-4 on the
    rays whose positions agree; code gradients of the run_fit-style loss against the reference's autograd."""
    g = golden("flags_e2e.npz")
    render, kw = _flags_product(g)
    H = int(g["H"])
    R = H * H
    bm, tex, exp = [T(g[k]).to(DEV).requires_grad_(True) for k in ("bm", "tex", "exp")]
    rgb, disp, acc, ex = render.render_fitting(H, H, g["K"], chunk=int(g["chunk"]), c2w=T(g["c2w"]), shapeCodes=bm, uvCodes=tex, expType=20,
                                               expCodes=exp, verbose=True, **kw)
    loss = (rgb - 0.5).abs().mean() + (ex["rgb0"] ** 2).mean()
    loss.backward()
    torch.cuda.synchronize()
    nan_equal_close(ex["rgb0"].detach().reshape(R, 3).cpu().numpy(), g["rgb0"].reshape(R, 3), 1e-4)
    nan_equal_close(ex["acc0"].detach().reshape(R).cpu().numpy(), g["acc0"].reshape(R), 1e-4)
    w_err = float((ex["_weights0"].detach().reshape(R, 64).cpu() - T(g["weights_coarse"])).abs().max())
    assert w_err < 2e-5
    agree, expl = classify_samples(g["z_coarse"], g["weights_coarse"], torch.linspace(0., 1., 64), ex["_z_samples"].reshape(R, 64).cpu(),
                                   g["z_samples"], w_err=w_err)
    assert (agree | expl).all()
    clean = agree.all(-1).numpy()
    assert clean.mean() > 0.3, clean.mean()
    e_rgb = nan_equal_close(rgb.detach().reshape(R, 3).cpu().numpy()[clean], g["rgb"].reshape(R, 3)[clean], 1e-4)
    e_acc = nan_equal_close(acc.detach().reshape(R).cpu().numpy()[clean], g["acc"].reshape(R)[clean], 1e-4)
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-4
    errs = {k: rel(v.grad.cpu().numpy(), g["g_" + k]) for k, v in (("bm", bm), ("tex", tex), ("exp", exp))}
    print(f"clean rays {clean.mean():.2f}; rgb {e_rgb:.1e} acc {e_acc:.1e}; grads", {k: f"{v:.1e}" for k, v in errs.items()})
    for k, v in errs.items():       # (loose: a few rays' sample positions differ legitimately, which perturbs their contributions)
        assert v < 0.08, (k, v)


def test_fine_pass_teacher_forced_at_non_shipped_flags(golden):
    """The reference's own sample positions through the HIP coarse / fine networks + compositing: 1e-4 on EVERY ray."""
    from oracle import mofa_oracle as orc
    g = golden("flags_e2e.npz")
    render, kw = _flags_product(g)
    H = int(g["H"])
    R = H * H
    ro, rd = orc.get_rays(H, H, g["K"], T(g["c2w"]))
    ro, rd = ro.reshape(-1, 3).contiguous().to(DEV), rd.reshape(-1, 3).contiguous().to(DEV)
    vd = (rd / torch.norm(rd, dim=-1, keepdim=True)).contiguous()
    render.shapeCodes, render.expType = T(g["bm"]).to(DEV), 20
    render.expCodes_Sigma.append(T(g["exp"]).to(DEV))
    for tag, net, S, sfx in (("coarse", kw["network_fn"], 64, "0"), ("fine", kw["network_fine"], 128, "")):
        with torch.no_grad():
            folded = render._fold_codes(net, T(g["tex"]).to(DEV))
        z = T(g[f"z_{tag}"]).contiguous().to(DEV)
        raw = torch.empty(R, S, 4, device=DEV)
        render._hip(net).forward_rays(ro, rd, z, S, vd, S, raw, folded)
        o = {k: torch.empty(R, *sh, device=DEV) for k, sh in (("rgb", (3,)), ("disp", ()), ("acc", ()), ("depth", ()), ("weights", (S,)))}
        lib.check(lib.load().mofa_composite_forward(lib.ptr(raw), lib.ptr(z), S, lib.ptr(rd), None, R, S, 0, lib.ptr(o["rgb"]), lib.ptr(o["disp"]),
                                                    lib.ptr(o["acc"]), lib.ptr(o["depth"]), lib.ptr(o["weights"]), lib.stream()), "composite")
        torch.cuda.synchronize()
        e_raw = nan_equal_close(raw.cpu().numpy(), g[f"raw_{tag}"], 1e-4, 1e-4)
        e_rgb = nan_equal_close(o["rgb"].cpu().numpy(), g["rgb" + sfx].reshape(R, 3), 1e-4)
        e_acc = nan_equal_close(o["acc"].cpu().numpy(), g["acc" + sfx].reshape(R), 1e-4)
        print(f"{tag}: raw {e_raw:.1e} rgb {e_rgb:.1e} acc {e_acc:.1e}")


def test_renderer_refuses_code_widths_the_networks_were_not_built_for(golden):
    """A shape / texture / expression code of the wrong width: the reference dies inside the first Linear (model.py:129-133); here the
    renderer says which width the network expects — nothing is launched on a mis-sized buffer."""
    g = golden("flags_e2e.npz")
    render, kw = _flags_product(g)
    bm, exp = T(g["bm"]).to(DEV), T(g["exp"]).to(DEV)
    call = lambda tex: render.render_fitting(8, 8, synth.intrinsics(8, 8), chunk=64, c2w=T(g["c2w"]), shapeCodes=bm, uvCodes=tex, expType=20,
                                             expCodes=exp, **kw)
    with torch.no_grad():
        call(T(g["tex"]).to(DEV))
        with pytest.raises(lib.MofaError, match="code widths"):
            call(synth.codes(0)[1].to(DEV))                  # the shipped 256-wide texture code into a 128-wide network
    with pytest.raises(lib.MofaError, match="code widths"):
        call(synth.codes(0)[1].to(DEV).requires_grad_(True))


@pytest.mark.parametrize("ch_in,ch_views,cs,ct,D,W", [(40, 10, 7, 33, 8, 64), (93, 27, 50, 256, 6, 128), (5, 3, 0, 16, 7, 64)])
def test_embedded_forward_accepts_any_per_point_widths_like_the_reference_module(ch_in, ch_views, cs, ct, D, W):
    """``NeRF.forward`` on already-embedded inputs (models/model.py:121-137) takes ANY ``input_ch`` / ``input_ch_views`` in the
    reference — there they are plain Linear input widths.  The embedded path no longer inherits the renderer path's
    ``input_ch >= 3 + 6 L`` / ``input_ch_views = 3 + 6 L`` rule; it checks the Linear shapes only and agrees with the oracle's
    restatement of the module on random weights."""
    from mofanerf_amd.model import NeRF
    from oracle import mofa_oracle as orc
    torch.manual_seed(ch_in + ch_views)
    net = NeRF(D=D, W=W, input_ch=ch_in, input_ch_views=ch_views, input_ch_textureCodes=ct, input_ch_shapeCodes=cs, use_viewdirs=True).to(DEV)
    n = 300
    rng = np.random.default_rng(ch_in)
    x, bm, v, tex = [T(rng.normal(0, 0.5, (n, c)).astype(np.float32)) for c in (ch_in, cs, ch_views, ct)]
    with torch.no_grad():
        raw = net(x.to(DEV), bm.to(DEV), v.to(DEV), tex.to(DEV))
    st = {k: p.detach().cpu() for k, p in net.state_dict().items()}
    ref = orc.nerf_forward(st, x, bm, v, tex)
    nan_equal_close(raw.cpu().numpy(), ref.numpy(), 2e-5, 1e-5)
    net2 = NeRF(D=D, W=W, input_ch=ch_in, input_ch_views=ch_views, input_ch_textureCodes=ct, input_ch_shapeCodes=cs, use_viewdirs=True).to(DEV)
    net2.linear_view_xyBMuv[0] = torch.nn.Linear(ch_views + W + 1, W // 2).to(DEV)        # a module the constructor would not build: refused
    with pytest.raises(lib.MofaError, match="refusing to pack"):
        with torch.no_grad():
            net2(x.to(DEV), bm.to(DEV), v.to(DEV), tex.to(DEV))
