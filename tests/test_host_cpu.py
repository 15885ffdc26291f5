"""Host-side logic that needs no GPU: the checkpoint schema against the reference's own modules (tests/golden/schema.json,
written by tests/golden/make_golden.py g6 from the reference) and the save -> create_nerf(ft_path) round trip."""
import json
import os

import numpy as np
import pytest
import torch

from mofanerf_amd import factory, schema, synth
from mofanerf_amd.model import NeRF

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ref_schema():
    with open(os.path.join(HERE, "golden", "schema.json")) as f:
        return json.load(f)


def _shapes(sd):
    return {k: list(v.shape) for k, v in sd.items()}


def test_state_dict_schema_equals_reference(ref_schema):
    """Every key name, its order and its shape, for the shipped sizes (coarse 256x8, fine 1024x10), StyleModule and the
    texture encoder; plus the flat `grad_vars` order the optimizer state is indexed by."""
    a = factory.default_args(device="cpu")
    kw, _, start, grad_vars, opt, _, render = factory.create_nerf(a)
    assert start == 0
    for name, mod in (("network_fn_state_dict", kw["network_fn"]), ("network_fine_state_dict", kw["network_fine"]),
                      ("network_render_textureEncoder", render.texEncoder), ("network_render_idSpecific", render.idSpecificMod)):
        got = _shapes(mod.state_dict())
        assert list(got.keys()) == list(ref_schema[name].keys()), name
        assert got == ref_schema[name], name
    assert [list(p.shape) for p in grad_vars] == ref_schema["grad_vars_shapes"]
    assert [list(t.shape) for t in render.expCodes_Sigma] == ref_schema["expression_latent_codes_sigma"]
    g = opt.state_dict()["param_groups"]
    assert len(g) == 1 and len(g[0]["params"]) == ref_schema["optimizer_param_groups"][0]["params"]
    assert g[0]["lr"] == ref_schema["optimizer_param_groups"][0]["lr"]
    assert list(g[0]["betas"]) == ref_schema["optimizer_param_groups"][0]["betas"]
    n = ref_schema["n_params"]
    assert sum(p.numel() for p in kw["network_fn"].parameters()) == n["coarse"] == 1598852
    assert sum(p.numel() for p in kw["network_fine"].parameters()) == n["fine"] == 27502084
    # schema.py's tables (what the C ABI's weight order is built from) agree with the same fixture
    assert {k: list(v) for k, v in schema.linear_shapes(schema.nerf_layers(10, 1024)).items()} == ref_schema["network_fine_state_dict"]
    assert {k: list(v) for k, v in schema.tex_encoder_shapes().items()} == ref_schema["network_render_textureEncoder"]
    assert {k: list(v) for k, v in schema.linear_shapes(schema.style_layers()).items()} == ref_schema["network_render_idSpecific"]


def test_checkpoint_round_trip(tmp_path, ref_schema):
    """save_checkpoint writes the reference's dictionary; create_nerf(ft_path=...) restores every tensor, the optimizer
    moments and the step counter (run_train.py:369-379 / create_model_condition.py:66-89)."""
    a = factory.default_args(device="cpu", netwidth=64, netwidth_fine=64, basedir=str(tmp_path), expname="rt")
    kw, _, _, grad_vars, opt, _, render = factory.create_nerf(a)
    kw["network_fn"].load_state_dict(synth.nerf_state(8, 64, 3, "coarse"))
    kw["network_fine"].load_state_dict(synth.nerf_state(10, 64, 3, "fine"))
    render.idSpecificMod.load_state_dict(synth.style_state(3))
    for dst, src in zip(render.expCodes_Sigma, synth.exp_sigma(3)):
        dst.data[:] = src
    g = torch.Generator().manual_seed(0)
    for p in grad_vars:                                  # one Adam step so that the optimizer has state to carry
        p.grad = torch.randn(p.shape, generator=g) * 1e-3
    opt.step()
    render.idSpecificMod = torch.nn.DataParallel(render.idSpecificMod)     # the scripts wrap it (run_fit.py:168)
    path = factory.save_checkpoint(os.path.join(str(tmp_path), "rt", "000123.tar"), 123, kw, render, opt)
    render.idSpecificMod = render.idSpecificMod.module

    blob = torch.load(path, map_location="cpu", weights_only=False)
    assert list(blob.keys()) == ref_schema["top_level_keys"]
    assert not any(k.startswith("module.") for k in blob["network_render_idSpecific"])
    assert _shapes(blob["network_render_textureEncoder"]) == ref_schema["network_render_textureEncoder"]

    kw2, _, start2, gv2, opt2, _, render2 = factory.create_nerf(a)          # picks the newest *.tar under basedir/expname
    assert start2 == 123
    for m1, m2 in ((kw["network_fn"], kw2["network_fn"]), (kw["network_fine"], kw2["network_fine"]),
                   (render.texEncoder, render2.texEncoder), (render.idSpecificMod, render2.idSpecificMod)):
        for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
            assert k1 == k2 and torch.equal(v1, v2), k1
    for t1, t2 in zip(render.expCodes_Sigma, render2.expCodes_Sigma):
        assert torch.equal(t1, t2)
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert s1.keys() == s2.keys() and len(s1) == len(grad_vars)
    for k in s1:
        assert torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) and torch.equal(s1[k]["exp_avg_sq"], s2[k]["exp_avg_sq"])
    a.no_reload = True
    assert factory.create_nerf(a)[2] == 0


def test_unsupported_configurations_raise():
    with pytest.raises(NotImplementedError):
        NeRF(D=8, W=64, input_ch=93, input_ch_views=27, use_viewdirs=False)
    with pytest.raises(NotImplementedError):
        NeRF(D=8, W=64, input_ch=93, input_ch_views=27, skips=(3,), use_viewdirs=True)
    net = NeRF(D=8, W=64, input_ch=93, input_ch_views=27, input_ch_shapeCodes=50, input_ch_textureCodes=256, use_viewdirs=True)
    z = [torch.zeros(1, k) for k in (93, 50, 27, 256)]
    with pytest.raises(RuntimeError, match="inference-only"):          # no silent detach when gradients would be needed
        net(*z)
    from mofanerf_amd import lib
    with pytest.raises(lib.MofaError), torch.no_grad():                 # and no CPU path: CPU parameters fail loudly
        net(*z)


def test_mac_counts_match_survey():
    assert schema.mac_per_point(8, 256, folded=False) == 1593600 and schema.mac_per_point(8, 256) == 1425792
    assert schema.mac_per_point(10, 1024, folded=False) == 27476992 and schema.mac_per_point(10, 1024) == 26805760


def test_landmark_pixel_samplers():
    """Device-side twins of the scripts' numpy samplers (run_train.py:119-148, run_fit.py:35-82): sizes, order, window,
    distinctness of the uniform part and the offset statistics."""
    from mofanerf_amd import rays
    g = torch.Generator().manual_seed(5)
    H = W = 512
    lm = torch.stack([torch.linspace(150, 360, 68).round(), torch.linspace(170, 340, 68).flip(0).round()], -1).long()
    n = 4096
    px = rays.train_pixels(lm, n, H, W, generator=g)
    p = int(n / 5 * 3 // 68)
    assert px.shape == (n, 2) and px.dtype == torch.long and p == 36
    uni, near = px[: n - 68 * p], px[n - 68 * p:]
    assert len({(int(a), int(b)) for a, b in uni}) == uni.shape[0]                 # drawn without replacement
    assert px.min() >= 0 and px.max() < H
    off = (near.reshape(68, p, 2) - lm[:, None, :]).float()
    assert torch.equal(off[0], off[37])                                            # one offset table shared by all landmarks
    assert abs(float(off.std()) - H * 0.025) < 3.0 and abs(float(off.mean())) < 3.0
    crop = rays.train_pixels(lm, n, H, W, precrop_frac=0.5, generator=g)[: n - 68 * p]
    assert crop.min() >= 128 and crop.max() < 384

    target = torch.zeros(256, 256, 3)
    target[60:200, 70:190] = 0.5                                                   # "face" region; elsewhere empty
    f = rays.fit_pixels(lm, 1024, target, scale=2, generator=g)
    assert f.shape == (1024, 2) and f.min() >= 0 and f.max() < 256
    inside = target[f[:, 0], f[:, 1]].sum(-1) != 0
    assert float(inside.float().mean()) > 0.9                                      # only the few outline extras may fall outside
    few = rays.fit_pixels(lm[:4], 1024, target, scale=2, generator=g)              # fewer candidates than n: tiled
    assert few.shape == (1024, 2)
    with pytest.raises(ValueError):
        rays.fit_pixels(lm, 64, torch.zeros(256, 256, 3), scale=2, generator=g)
    # rays of sampled pixels == the same pixels of the full grid (the gather the scripts do), differentiable in the pose
    K = np.array([[1200.0, 0, 256], [0, 1200.0, 256], [0, 0, 1]])
    c2w = rays.pose_spherical(30.0, 0.0, 16.0)[:3, :4].clone().requires_grad_(True)
    br = rays.rays_at_pixels(K, c2w, px[:, 0], px[:, 1])
    ii, jj = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="xy")
    dirs = torch.stack([(ii - 256) / 1200, -(jj - 256) / 1200, -torch.ones_like(ii)], -1)
    full_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    assert torch.allclose(br[1], full_d[px[:, 0], px[:, 1]], atol=1e-6)
    br[1].sum().backward()
    assert c2w.grad is not None and float(c2w.grad.abs().sum()) > 0


def test_png_sink_matches_to8b(tmp_path):
    """The asynchronous output stage writes exactly the reference's 8-bit quantisation ((255*clip(x,0,1)) truncated), leaves
    no partial files behind and surfaces write errors at close()."""
    from PIL import Image
    from mofanerf_amd.io import PngSink, write_png
    rng = np.random.default_rng(2)
    imgs = [rng.uniform(-0.2, 1.2, (h, w, 3)).astype(np.float32) for h, w in ((16, 16), (33, 7), (64, 128))]
    with PngSink(workers=2) as sink:
        for i, im in enumerate(imgs):
            sink.submit(str(tmp_path / f"t{i}.png"), torch.from_numpy(im))
            sink.submit(str(tmp_path / f"n{i}.png"), im)
    for i, im in enumerate(imgs):
        want = (255 * np.clip(im, 0, 1)).astype(np.uint8)
        for stem in ("t", "n"):
            got = np.asarray(Image.open(tmp_path / f"{stem}{i}.png").convert("RGB"))
            assert np.array_equal(got, want)
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".part")]
    sink = PngSink()
    sink.submit(str(tmp_path / "no_such_dir" / "x.png"), imgs[0])
    with pytest.raises(OSError):
        sink.close()
    with pytest.raises(ValueError):
        write_png(str(tmp_path / "bad.png"), np.zeros((4, 4), np.uint8))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="build container only: loads the file into the reference's own modules")
def test_checkpoint_written_here_loads_into_the_reference_modules(tmp_path):
    """(f4) `save_checkpoint` output is consumed by the REFERENCE's modules with strict `load_state_dict` and its own
    reload loop (create_model_condition.py:72-89) — run in a subprocess because importing the reference needs process-wide
    shims (stub imageio / cv2, Tensor.cuda = identity)."""
    import subprocess
    import sys
    a = factory.default_args(device="cpu", netwidth=64, netwidth_fine=64, basedir=str(tmp_path), expname="rt", no_reload=True)
    kw, _, _, grad_vars, opt, _, render = factory.create_nerf(a)
    kw["network_fn"].load_state_dict(synth.nerf_state(8, 64, 5, "coarse"))
    kw["network_fine"].load_state_dict(synth.nerf_state(10, 64, 5, "fine"))
    path = factory.save_checkpoint(str(tmp_path / "rt" / "000042.tar"), 42, kw, render, opt)
    code = f"""
import sys, types, torch
sys.path.insert(0, '/root/reference')
for m in ('imageio', 'cv2'):
    sys.modules[m] = types.ModuleType(m)
torch.Tensor.cuda = lambda self, *a, **k: self
from models import render_class
from models.model import NeRF, get_embedder
ck = torch.load({path!r}, map_location='cpu', weights_only=False)
mk = lambda D, W: NeRF(D=D, W=W, input_ch_shapeCodes=50, input_ch_textureCodes=256, input_ch=93, output_ch=5, skips=[4],
                       input_ch_views=27, use_viewdirs=True)
coarse, fine = mk(8, 64), mk(10, 64)
r = render_class.myRenderer(embed_fn=get_embedder(10, 0)[0], embeddirs_fn=get_embedder(4, 0)[0], netchunk=4096, uvCodesLen=256,
                            expCodesLen=30)
opt = torch.optim.Adam(params=list(coarse.parameters()) + list(fine.parameters()) + list(r.grad_parameter()), lr=5e-5)
opt.load_state_dict(ck['optimizer_state_dict'])
coarse.load_state_dict(ck['network_fn_state_dict'])
fine.load_state_dict(ck['network_fine_state_dict'])
r.texEncoder.load_state_dict(ck['network_render_textureEncoder'])
r.idSpecificMod.load_state_dict(ck['network_render_idSpecific'])
for latent, saved in zip(r.expCodes_Sigma, ck['expression_latent_codes_sigma']):
    latent.data[:] = saved[:].detach().clone()
assert ck['global_step'] == 42
print('REFERENCE_LOADED', float(fine.rgb_linear.weight.sum()))
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "REFERENCE_LOADED" in out.stdout, out.stderr[-1500:]
    got = float(out.stdout.split("REFERENCE_LOADED")[1].split()[0])
    assert abs(got - float(kw["network_fine"].rgb_linear.weight.detach().sum())) < 1e-5


def test_embedded_forward_cache_does_not_keep_modules_alive():
    """model._EMBEDDED_CACHE maps NeRF -> HipNet weakly; the HipNet it stores holds only a WEAK back-reference to the module
    (HipNet(net, weak=True)), so dropping the module drops the entry and its packed device panels."""
    import gc
    import weakref
    from mofanerf_amd.hipnet import HipNet
    from mofanerf_amd.model import NeRF, _EMBEDDED_CACHE
    n0 = len(_EMBEDDED_CACHE)
    net = NeRF(D=8, W=64, input_ch=93, input_ch_views=27, input_ch_textureCodes=256, input_ch_shapeCodes=50, use_viewdirs=True)
    h = _EMBEDDED_CACHE[net] = HipNet(net, weak=True)
    assert h.net is net and (h.D, h.W) == (8, 64)
    ref = weakref.ref(net)
    del net, h
    gc.collect()
    assert ref() is None and len(_EMBEDDED_CACHE) == n0
    strong = HipNet(NeRF(D=8, W=64, input_ch=93, input_ch_views=27, input_ch_textureCodes=256, input_ch_shapeCodes=50, use_viewdirs=True))
    gc.collect()
    assert strong.net is not None                     # the renderer's own HipNets keep their module (default: strong)


def test_bench_roofline_of_picks_the_kernel_with_the_largest_summed_time():
    """bench.py's `roofline` object (ours and the variants'): dominant kernel = largest summed HIP-event time; achieved = its algorithmic
    FLOPs / that time; frac against the fp32 matrix peak; everything else listed under `other_mfma_kernels`."""
    import ctypes
    import bench
    from mofanerf_amd import lib
    NK = lib.PROF_KINDS
    ms, launches, pflops = (ctypes.c_double * NK)(), (ctypes.c_int64 * NK)(), (ctypes.c_double * NK)()
    ms[5], launches[5], pflops[5] = 705.0, 10, 10 * 10.52e12            # ten chained launches of 10.52 TFLOP in 70.5 ms each
    ms[1], launches[1], pflops[1] = 40.0, 10, 10 * 0.5604e12            # the coarse network's persistent kernel
    dom, r = bench.roofline_of(ms, launches, pflops, dt=0.75)
    assert dom == 5 and r["kernel"].startswith("mofa::k_net_chain<0>") and r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["achieved"] - 149.22) < 0.01 and abs(r["frac"] - 149.22 / 157.3) < 1e-4 and r["peak"] == 157.3
    assert r["launches"] == 10 and abs(r["avg_launch_ms"] - 70.5) < 1e-9 and abs(r["algorithmic_gflop_per_launch"] - 10520.0) < 1e-6
    assert abs(r["share_of_timed_region"] - 0.94) < 1e-9
    assert [o["kernel"] for o in r["other_mfma_kernels"]] == ["mofa::k_mlp_fused"] and abs(r["other_mfma_kernels"][0]["tflops"] - 140.1) < 0.01


def test_bench_roofline_hbm_prices_the_ray_kernels_against_the_hbm_peak():
    """SURVEY section 8d's second roofline (VERDICT r5 missing 3): compositing / resampling kernels, algorithmic bytes per ray x the rays the
    launches processed / their HIP-event time, against 8 TB/s; PMC traffic quoted only under the same source digest."""
    import ctypes
    import bench
    from mofanerf_amd import lib
    NK = lib.PROF_KINDS
    ms, launches, work = (ctypes.c_double * NK)(), (ctypes.c_int64 * NK)(), (ctypes.c_double * NK)()
    ms[5], launches[5], work[5] = 705.0, 10, 10 * 10.52e12                       # an MFMA kind: not part of the HBM list
    ms[8], launches[8], work[8] = 0.2, 2, 2 * 131072.0                            # two coarse composites of 131,072 rays, 100 us each
    ms[10], launches[10], work[10] = 0.5, 2, 2 * 131072.0
    assert bench.HBM_KINDS == {8: 1568, 9: 2616, 10: 1284}                        # SURVEY 8d's per-ray figures (1,568 / 2,616 / ~1.3 k)
    tj = {"csrc_sha256": "abc", "kernels": {"mofa::k_composite<1>": {"bytes_per_launch": 250e6, "rays_per_launch": 131072, "algorithmic_bytes_per_launch": 205520896}}}
    r = bench.roofline_hbm_of(ms, launches, work, dt=1.0, traffic_json=tj, digest="abc")
    assert [x["kernel"].split(" ")[0] for x in r] == ["mofa::k_composite<1>", "mofa::k_sample_pdf_merge<false>"]
    c = r[0]
    assert c["bound"] == "hbm" and c["unit"] == "GB/s" and c["peak"] == 8000.0 and c["launches"] == 2 and abs(c["avg_launch_us"] - 100.0) < 1e-9
    assert abs(c["achieved"] - 131072 * 1568 / 100e-6 / 1e9) < 0.1 and abs(c["frac"] - c["achieved"] / 8000.0) < 1e-4 and c["traffic"] == 250e6
    assert r[1]["traffic"] is None
    stale = bench.roofline_hbm_of(ms, launches, work, dt=1.0, traffic_json=tj, digest="other")
    assert stale[0]["traffic"] is None and "null" in stale[0]["traffic_source"]
    dom, mf = bench.roofline_of(ms, launches, work, dt=1.0)                       # the MFMA roofline never picks a ray kernel
    assert dom == 5


def test_committed_traffic_profiles_were_taken_on_the_committed_kernel_sources():
    """bench.py quotes `roofline.traffic` / `roofline_hbm[*].traffic` from profiles/hbm_traffic_chain.json / hbm_traffic_rays.json only while the
    kernel sources hash to what the rocprofv3 --pmc passes were taken on (`csrc_sha256` = build.csrc_digest()); otherwise the driver's line says
    `traffic: null`.  The committed pair must therefore describe the committed sources — a source change without re-running
    tools/gpu_profile_round6.sh fails HERE, not silently in the driver's line."""
    import json
    from mofanerf_amd import build
    digest = build.csrc_digest()
    for name in ("hbm_traffic_chain.json", "hbm_traffic_rays.json"):
        tj = json.load(open(os.path.join(os.path.dirname(HERE), "profiles", name)))
        assert tj["csrc_sha256"] == digest, f"profiles/{name} was taken on kernel sources {tj['csrc_sha256'][:16]}, the tree is {digest[:16]}"
    chain = json.load(open(os.path.join(os.path.dirname(HERE), "profiles", "hbm_traffic_chain.json")))
    assert chain["kernel"].startswith("mofa::k_net_chain<0>") and chain["bytes_per_launch"] > chain["algorithmic_bytes_per_launch"] > 0
