#!/usr/bin/env python3
"""Driver for rocprofv3 passes over a TRAINING step's network pass: the shipped fine network (1024 x 10) on one sub-batch of the benchmark's
training step (1,536 rays x 128 samples = 768 row tiles), tape-keeping forward + backward with weight gradients, three steps.
MOFA_CHAIN_TRAIN=1: k_net_chain<0> + two k_net_chain_train launches per step; default / MOFA_CHAIN=0: the per-layer backward (k_layer<BWD>, k_wgrad).
tools/gpu_profile_train.sh wraps it."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_chain_setup import setup  # noqa: E402
from mofanerf_amd.autograd import NetFn  # noqa: E402

R, S = 1536, 128
h, o, d, z, vd, folded, vb, G = setup(10, 1024, R, S)
for _ in range(int(os.environ.get("PMC_TRAIN_STEPS", "3"))):
    og, dg = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    fo, vbg = folded.clone().requires_grad_(True), vb.clone().requires_grad_(True)
    ws = [l.weight.detach().clone().requires_grad_(True) for l in h._linears]
    raw = NetFn.apply(h, og, dg, z, S, S, fo, vbg, None, *ws)
    (raw * G).sum().backward()
torch.cuda.synchronize()
h.check_verdict(block=True)
