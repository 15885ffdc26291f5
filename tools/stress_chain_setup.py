"""Seeded inputs of one network pass (shared by tools/pmc_train.py; the same recipe as tools/stress_chain.py::setup)."""
import numpy as np
import torch
from mofanerf_amd import synth
from mofanerf_amd.autograd import view_bias_torch
from mofanerf_amd.hipnet import HipNet
from mofanerf_amd.model import NeRF

DEV = "cuda"


def setup(D, W, R, S, seed=1):
    rng = np.random.default_rng(D + W + R + S)
    net = NeRF(D=D, W=W, input_ch=93, input_ch_views=27, input_ch_textureCodes=256, input_ch_shapeCodes=50, use_viewdirs=True)
    net.load_state_dict(synth.nerf_state(D, W, seed))
    h = HipNet(net.to(DEV))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    o = t(rng.uniform(-2, 2, (R, 3)).astype(np.float32))
    d = t(rng.normal(0, 0.3, (R, 3)).astype(np.float32))
    z = t(np.sort(rng.uniform(8, 26, (R, S)).astype(np.float32), -1))
    vd = torch.nn.functional.normalize(d, dim=-1).contiguous()
    bm, tex, e = synth.codes(3)
    folded = h.fold(e.to(DEV), bm.to(DEV), tex.to(DEV)).clone()
    vb = view_bias_torch(h, vd).detach().contiguous()
    G = t(rng.normal(size=(R, S, 4)).astype(np.float32))
    return h, o, d, z, vd, folded, vb, G
