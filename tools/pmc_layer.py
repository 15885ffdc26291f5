#!/usr/bin/env python3
"""Launch the dominant kernel a few times at the shipped fine-network shape (M=196608 points, K=N=1024) so that
rocprofv3 --pmc can attribute counters to individual dispatches — and then at N = 512 / 256 / 128 (4 / 2 / 1 feature tiles per
point tile, weight slabs of 2 / 1 / 0.5 MiB): the activation read is the same 805 MB in all four, so how the past-L2 read
traffic moves with N tells whether the 2x of the full shape is activations fetched twice or the weight slab cycling through
the 4 MiB L2s (tools/make_traffic_json.py separates the launches by grid size)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import lib
L = lib.load()
M, K = 196608, 1024
x = torch.randn(M * K, device="cuda")
for N in (1024, 512, 256, 128):
    w = torch.randn(N * K, device="cuda") * 0.03
    b = torch.randn(N, device="cuda"); y = torch.empty(M * N, device="cuda")
    for _ in range(4):
        lib.check(L.mofa_layer_forward(lib.ptr(x), K, None, 0, lib.ptr(w), lib.ptr(b), 0, 1, lib.ptr(y), M, N, 1, lib.stream()), "layer")
    torch.cuda.synchronize()
print("done")
