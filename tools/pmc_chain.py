#!/usr/bin/env python3
"""Driver for rocprofv3 --pmc passes over the chained wide-network kernel k_net_chain: the shipped fine network (1024 x 10) forward on
196,608 points (one sub-batch of the 512 x 512 frame: 768 row tiles x 27 layers), four launches.  tools/gpu_profile_chain.sh wraps it,
one pass per counter group; tools/make_traffic_chain_json.py turns the passes into profiles/hbm_traffic_chain.json."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import synth
from mofanerf_amd.hipnet import HipNet
from mofanerf_amd.model import NeRF

net = NeRF(D=10, W=1024, input_ch=93, input_ch_views=27, input_ch_textureCodes=256, input_ch_shapeCodes=50, use_viewdirs=True)
net.load_state_dict(synth.nerf_state(10, 1024, 0, "fine"))
h = HipNet(net.cuda())
R, S = 1536, 128
g = torch.Generator(device="cuda").manual_seed(0)
o = torch.randn(R, 3, device="cuda", generator=g)
d = torch.randn(R, 3, device="cuda", generator=g) * 0.3
z = torch.sort(torch.rand(R, S, device="cuda", generator=g) * 18 + 8, -1)[0].contiguous()
vd = torch.nn.functional.normalize(d, dim=-1).contiguous()
bm, tex, e = synth.codes(0)
folded = h.fold(e.cuda(), bm.cuda(), tex.cuda()).clone()
raw = torch.empty(R, S, 4, device="cuda")
for _ in range(4):
    h.forward_rays(o, d, z, S, vd, S, raw, folded)
torch.cuda.synchronize()
