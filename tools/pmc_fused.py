#!/usr/bin/env python3
"""Driver for rocprofv3 --pmc passes over the persistent network kernels: the shipped coarse network (256 x 8) forward on 196,608
points (one sub-batch of the 512 x 512 frame: 1,536 tiles of 128 points), four launches per arm.  Arms = the library's bit-identical
launch forms: `pipelined` (k_mlp_fused<false>, the default), `generic` (k_mlp_fused_generic<false> with its plain loops: MOFA_PIPE=0).
(profiles/r04_pmc_resident_experiment_*.csv: the same passes on commit a90d63b, incl. the LDS-resident experiment's k_mlp_resident<4|8>.)
tools/gpu_profile_fused.sh wraps it, one pass per counter group."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import lib, synth
from mofanerf_amd.hipnet import HipNet
from mofanerf_amd.model import NeRF

arms = sys.argv[1:] or ["pipelined", "generic"]
net = NeRF(D=8, W=256, input_ch=93, input_ch_views=27, input_ch_textureCodes=256, input_ch_shapeCodes=50, use_viewdirs=True)
net.load_state_dict(synth.nerf_state(8, 256, 0, "coarse"))
h = HipNet(net.cuda())
R, S = 3072, 64
g = torch.Generator(device="cuda").manual_seed(0)
o = torch.randn(R, 3, device="cuda", generator=g)
d = torch.randn(R, 3, device="cuda", generator=g) * 0.3
z = torch.sort(torch.rand(R, S, device="cuda", generator=g) * 18 + 8, -1)[0].contiguous()
vd = torch.nn.functional.normalize(d, dim=-1).contiguous()
bm, tex, e = synth.codes(0)
folded = h.fold(e.cuda(), bm.cuda(), tex.cuda()).clone()
raw = torch.empty(R, S, 4, device="cuda")
env = {"pipelined": {}, "generic": {"MOFA_PIPE": "0"}}
for arm in arms:
    for k in ("MOFA_PIPE", "MOFA_FUSED"):
        os.environ.pop(k, None)
    os.environ.update(env[arm])
    lib.reload_env()
    for _ in range(4):
        h.forward_rays(o, d, z, S, vd, S, raw, folded)
    torch.cuda.synchronize()
