#!/usr/bin/env python3
"""Same as pmc_layer.py at the coarse-net shape (M=196608, K=N=256) and the skip-layer shape (K=512)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import lib
L = lib.load()
M = 196608
for (K, N) in ((256, 256), (512, 256)):
    x = torch.randn(M * K, device="cuda"); w = torch.randn(N * K, device="cuda") * 0.03
    b = torch.randn(N, device="cuda"); y = torch.empty(M * N, device="cuda")
    for _ in range(4):
        lib.check(L.mofa_layer_forward(lib.ptr(x), K, None, 0, lib.ptr(w), lib.ptr(b), 0, 1, lib.ptr(y), M, N, 1, lib.stream()), "layer")
torch.cuda.synchronize()
