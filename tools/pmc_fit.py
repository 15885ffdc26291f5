#!/usr/bin/env python3
"""Driver for rocprofv3 passes over a FITTING step's fine-network pass: the shipped fine network (1024 x 10) on 1,024 rays x 128 samples (512 row
tiles: BASELINE configs[2]'s step), forward writing the mask tape (k_net_chain<1>) + the chained backward's two launches (k_net_chain<2>), three
steps.  tools/gpu_profile_fit.sh wraps it; tools/pmc_summary.py turns the passes into a table."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_chain_setup import setup  # noqa: E402
from mofanerf_amd.autograd import NetFn  # noqa: E402

R, S = 1024, 128
h, o, d, z, vd, folded, vb, G = setup(10, 1024, R, S)
for _ in range(int(os.environ.get("PMC_FIT_STEPS", "3"))):
    og, dg = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    fo, vbg = folded.clone().requires_grad_(True), vb.clone().requires_grad_(True)
    raw = NetFn.apply(h, og, dg, z, S, S, fo, vbg, None)
    (raw * G).sum().backward()
torch.cuda.synchronize()
h.check_verdict(block=True)
