"""BASELINE.json configs[3] at FULL size inside the driver's GPU test run: render_refine_trainSet.py's job shape — one identity x one
expression x two views through ``render_path`` at 256 x 256 (the script halves hwf, render_refine_trainSet.py:288-295), the SHIPPED
network widths (coarse 256 x 8, fine 1024 x 10), the texture encoder on a 512 x 512 UV map, PNGs through one shared asynchronous sink
(models/render_class.py:199-237).  Checked: (i) 256 rays of one frame teacher-forced against the CPU oracle at the north star's 1e-4
(coarse pass ray by ray; the device's own resampled positions through the oracle's fine network + raw2outputs), those rays
reproducing the frame's pixels; (ii) the PNG files hold exactly ``to8b`` of the returned frames; (iii) a second call skips the
finished files (the bulk job's resume)."""
import numpy as np
import pytest
import torch

from harness import make_oracle, make_product
from mofanerf_amd import rays, synth
from mofanerf_amd.io import PngSink

pytestmark = pytest.mark.gpu
DEV = "cuda"
ARCH = (8, 256, 10, 1024)
H = 256


def test_config3_render_path_full_size_teacher_forced_and_png(tmp_path):
    from PIL import Image
    from oracle import mofa_oracle as orc
    render, kw, _ = make_product(ARCH, 0, 196608, DEV, with_tex=True)
    K = synth.intrinsics(H, H)                                                  # focal 600 at 256^2 (run_fit.py:357-362 halves 1200 @ 512)
    uv_cpu = torch.from_numpy(np.random.default_rng(3).uniform(0, 1, (512, 512, 3)).astype(np.float32))
    uv = uv_cpu.to(DEV)[None].expand(2, -1, -1, -1)
    bm_cpu = synth.codes(3)[0]
    bm = bm_cpu.to(DEV).expand(2, -1)
    angles = (-45.0, 30.0)
    poses = torch.stack([rays.pose_spherical(a, 0.0, 16.0) for a in angles], 0)
    exp_type = torch.tensor([7, 7])
    render.png_sink = PngSink(workers=2)
    with torch.no_grad():
        rgbs, disps = render.render_path(poses, [H, H, float(K[0][0])], K, 196608, kw, uvMap=uv, expType=exp_type, savedir=str(tmp_path),
                                         shapeCodes=bm)
    render.png_sink.close()
    render.png_sink = None
    assert rgbs.shape == (2, H, H, 3) and disps.shape == (2, H, H) and np.isfinite(rgbs).all()
    assert not np.array_equal(rgbs[0], rgbs[1])
    # ---- (ii) the files are to8b of the frames
    for i in range(2):
        got = np.asarray(Image.open(tmp_path / f"{i:03d}.png").convert("RGB"))
        assert np.array_equal(got, (255 * np.clip(rgbs[i], 0, 1)).astype(np.uint8))
    # ---- (i) 256 rays of frame 1, teacher-forced
    n = 256
    idx = torch.from_numpy(np.random.default_rng(11).choice(H * H, n, replace=False)).sort()[0]
    ro, rd = orc.get_rays(H, H, K, poses[1][:3, :4])
    ro, rd = ro.reshape(-1, 3)[idx].contiguous(), rd.reshape(-1, 3)[idx].contiguous()
    with torch.no_grad():
        rgb, disp, acc, ex = render.render(H, H, K, chunk=196608, rays=torch.stack([ro, rd], 0).to(DEV), shapeCodes=bm[:1], uvMap=uv[1], expType=7,
                                           verbose=True, **kw)
    frame = torch.from_numpy(rgbs[1]).reshape(-1, 3)
    # the same pixels as the frame's (its rays came from the device's ray kernel, viewdirs = d / |d| with correctly rounded sqrt and
    # divide; a caller-built ray batch gets them from torch's norm on the GPU, an ulp away — so close, not bit-equal)
    assert float((rgb.cpu() - frame[idx]).abs().max()) < 2e-5
    o = make_oracle(ARCH, 0, 196608, with_tex=True)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    with torch.no_grad():
        tex = orc.tex_encoder(o.tex_enc, uv_cpu)
        t = torch.linspace(0., 1., 64)
        zc = (8.0 * (1. - t) + 26.0 * t).expand(n, 64)
        raw0 = o.run_network(ro[:, None, :] + rd[:, None, :] * zc[:, :, None], vd, o.coarse, bm_cpu, tex, 7)
        rgb0_r, _, acc0_r, _, _ = orc.raw2outputs(raw0, zc, rd)
        zf = ex["_z_fine"].cpu()
        raw1 = o.run_network(ro[:, None, :] + rd[:, None, :] * zf[:, :, None], vd, o.fine, bm_cpu, tex, 7)
        rgb_r, disp_r, acc_r, _, _ = orc.raw2outputs(raw1, zf, rd)
    err = lambda a, b: float((a.cpu() - b).abs().max())
    errs = dict(tex=err(render.decoding_texCodes.reshape(-1), tex.reshape(-1)), rgb0=err(ex["rgb0"], rgb0_r), acc0=err(ex["acc0"], acc0_r),
                rgb=err(rgb, rgb_r), acc=err(acc, acc_r))
    print({k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["tex"] < 1e-4 and max(errs["rgb0"], errs["acc0"], errs["rgb"], errs["acc"]) <= 1e-4, errs
    assert torch.equal(torch.isnan(disp.cpu()), torch.isnan(disp_r))
    # ---- (iii) resume: a named image that exists is skipped
    with torch.no_grad():
        r1 = render.render_path(poses[:1], [H, H, float(K[0][0])], K, 196608, kw, uvMap=uv[:1], expType=exp_type[:1], savedir=str(tmp_path),
                                shapeCodes=bm[:1], name="000")
    assert r1 == (0, 0)
