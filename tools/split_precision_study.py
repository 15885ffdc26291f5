#!/usr/bin/env python3
"""CPU study (no GPU): how much RGB error would emulating every fp32 product with split-bf16 partial products add?
Every Linear of the oracle network is re-evaluated with operands split into 3 bf16 pieces (a = a1 + a2 + a3) and
  x3 : a1b1 + a1b2 + a2b1                      (16/3 = 5.3x the fp32-MFMA rate on the bf16 matrix pipe)
  x6 : all terms with i + j <= 4               (16/6 = 2.7x)
  x9 : all nine terms                          (16/9 = 1.8x)
accumulated in fp32, on the e2e_small fixture's rays with IDENTICAL sample positions (teacher forced)."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from mofanerf_amd import synth
from oracle import mofa_oracle as orc

T = torch.from_numpy


def split3(a):
    a1 = a.to(torch.bfloat16).float(); r = a - a1
    a2 = r.to(torch.bfloat16).float(); r = r - a2
    a3 = r.to(torch.bfloat16).float()
    return a1, a2, a3


def split_f16(a):
    a1 = a.to(torch.float16).float(); r = a - a1
    a2 = r.to(torch.float16).float(); r = r - a2
    a3 = r.to(torch.float16).float()
    return a1, a2, a3


def make_linear(terms, splitter=None):
    splitter = splitter or split3

    def lin(x, w, b=None):
        xs, ws = splitter(x), splitter(w)
        out = None
        for (i, j) in terms:                      # small terms first, like a careful accumulation order
            p = xs[i] @ ws[j].t()
            out = p if out is None else out + p
        return out if b is None else out + b
    return lin


TERMS = {"x3": [(0, 1), (1, 0), (0, 0)], "x6": [(0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)],
         "x9": [(2, 2), (1, 2), (2, 1), (0, 2), (1, 1), (2, 0), (0, 1), (1, 0), (0, 0)]}

g = dict(np.load(os.path.join(root, "tests", "golden", "e2e_small.npz")))
arch = [int(v) for v in g["arch"]]
o = orc.OracleRenderer(synth.nerf_state(arch[0], arch[1], 0, "coarse"), synth.nerf_state(arch[2], arch[3], 0, "fine"),
                       synth.style_state(0), synth.exp_sigma(0), netchunk=1 << 20)
o.exp_sigma.append(T(g["exp"]))
H = int(g["H"])
ro, rd = orc.get_rays(H, H, g["K"], T(g["c2w"]))
ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
vd = rd / torch.norm(rd, dim=-1, keepdim=True)
zf = T(g["z_fine"])
pts = ro[:, None, :] + rd[:, None, :] * zf[:, :, None]


def render():
    with torch.no_grad():
        raw = o.run_network(pts, vd, o.fine, T(g["bm"]), T(g["tex"]), 20)
        return raw, orc.raw2outputs(raw, zf, rd)[0]


raw0, rgb0 = render()
assert np.array_equal(rgb0.numpy(), g["rgb"].reshape(-1, 3))
print(f"fine net {arch[3]}x{arch[2]}, {pts.shape[0]} rays x {pts.shape[1]} samples, identical sample positions")
orig = F.linear
CASES = [(n, t, None) for n, t in TERMS.items()] + [("fp16x3 (2 fp16 pieces, 3 products)", TERMS["x3"], split_f16),
                                                   ("fp16x1 (plain fp16 inputs)", [(0, 0)], split_f16)]
for name, terms, splitter in CASES:
    orc.F.linear = make_linear(terms, splitter)
    raw, rgb = render()
    orc.F.linear = orig
    print(f"{name}: max |raw - raw_fp32| = {float((raw - raw0).abs().max()):.2e}   max |rgb - rgb_fp32| = {float((rgb - rgb0).abs().max()):.2e}")
