#!/usr/bin/env python3
"""Driver for rocprofv3 --pmc passes over the HBM-bound ray kernels at the benchmark's shapes (one render_rays chunk of the 512 x 512 frame:
196,608 rays): k_composite<1> (64 coarse samples, one shared z row), k_sample_pdf_merge<false> (64 + 64, det) and k_composite<2> (128 merged
samples, per-ray z), four launches each.  tools/gpu_profile_rays.sh wraps it, one pass per counter group; tools/make_traffic_rays_json.py turns
the passes into profiles/hbm_traffic_rays.json (what bench.py quotes as roofline_hbm[*].traffic)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import lib

L = lib.load()
R, S, Ni = 196608, 64, 64
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
st = lib.stream()
rays_d = torch.randn(R, 3, device=dev, generator=g) * 0.3
z = torch.linspace(0., 1., S, device=dev) * 18 + 8
raw0 = torch.randn(R, S, 4, device=dev, generator=g)
u = torch.linspace(0., 1., Ni, device=dev)
o = lambda *sh: torch.empty(R, *sh, dtype=torch.float32, device=dev)
rgb, disp, acc, depth, w0 = o(3), o(), o(), o(), o(S)
zs, zf, zstd = o(Ni), o(S + Ni), o()
raw1 = torch.randn(R, S + Ni, 4, device=dev, generator=g)
w1 = o(S + Ni)
for _ in range(4):
    lib.check(L.mofa_composite_forward(lib.ptr(raw0), lib.ptr(z), 0, lib.ptr(rays_d), None, R, S, 0, lib.ptr(rgb), lib.ptr(disp), lib.ptr(acc),
                                       lib.ptr(depth), lib.ptr(w0), st), "composite coarse")
    lib.check(L.mofa_sample_pdf_merge(lib.ptr(z), 0, lib.ptr(w0), lib.ptr(u), 0, R, S, Ni, lib.ptr(zs), lib.ptr(zf), lib.ptr(zstd), st), "sample_pdf_merge")
    lib.check(L.mofa_composite_forward(lib.ptr(raw1), lib.ptr(zf), S + Ni, lib.ptr(rays_d), None, R, S + Ni, 0, lib.ptr(rgb), lib.ptr(disp),
                                       lib.ptr(acc), lib.ptr(depth), lib.ptr(w1), st), "composite fine")
torch.cuda.synchronize()
assert bool(torch.isfinite(rgb).all()) and bool((zf[:, 1:] >= zf[:, :-1]).all())
