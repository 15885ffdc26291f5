#!/usr/bin/env python3
"""Race hunt: launch the hot kernels many times on the same inputs and require bit-identical outputs every time (a data race in
an LDS pipeline shows up as a rare mismatch, not as an error).  Shapes are the shipped ones (M = 196608 points, W = 1024; the
persistent 256-wide network kernel through a whole 128x128 frame).  Prints one line per kernel; exit code 1 on any mismatch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import lib  # noqa: E402

L = lib.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
N_FWD = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0


def report(name, n, mism):
    global bad
    bad += mism
    print(f"{name}: {n} launches, {mism} mismatching", flush=True)


M, K, N = 196608, 1024, 1024
x = torch.randn(M * K, device=dev, generator=g)
w = torch.randn(N * K, device=dev, generator=g) * 0.03
b = torch.randn(N, device=dev, generator=g)
st = lib.stream()
ref = torch.empty(M * N, device=dev)
lib.check(L.mofa_layer_forward(lib.ptr(x), K, None, 0, lib.ptr(w), lib.ptr(b), 0, 1, lib.ptr(ref), M, N, 1, st), "fwd")
y = torch.empty_like(ref)
mism = 0
for _ in range(N_FWD):
    y.fill_(-1.0)
    lib.check(L.mofa_layer_forward(lib.ptr(x), K, None, 0, lib.ptr(w), lib.ptr(b), 0, 1, lib.ptr(y), M, N, 1, st), "fwd")
    mism += int(not torch.equal(y, ref))
report("k_layer forward (pipelined K loop, staged epilogue), M=196608 K=N=1024", N_FWD, mism)

mask = torch.randn(M * N, device=dev, generator=g)
dx0 = torch.randn(M * N, device=dev, generator=g)
ref = dx0.clone()
lib.check(L.mofa_layer_backward_data(lib.ptr(x), K, lib.ptr(w), lib.ptr(mask), 1, lib.ptr(ref), M, N, st), "bwd")
mism = 0
for _ in range(N_FWD):
    y.copy_(dx0)
    lib.check(L.mofa_layer_backward_data(lib.ptr(x), K, lib.ptr(w), lib.ptr(mask), 1, lib.ptr(y), M, N, st), "bwd")
    mism += int(not torch.equal(y, ref))
report("k_layer backward-data (mask + accumulate)", N_FWD, mism)

n_points = M
ws = torch.empty(L.mofa_weight_grad_workspace_floats(n_points, N, K), device=dev)
dw_ref, db_ref = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
lib.check(L.mofa_weight_grad(lib.ptr(mask), N, lib.ptr(x), K, M, n_points, N, K, lib.ptr(dw_ref), K, 0, lib.ptr(db_ref), lib.ptr(ws), st), "wgrad")
mism = 0
for _ in range(max(1, N_FWD // 3)):
    dw, db = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    lib.check(L.mofa_weight_grad(lib.ptr(mask), N, lib.ptr(x), K, M, n_points, N, K, lib.ptr(dw), K, 0, lib.ptr(db), lib.ptr(ws), st), "wgrad")
    mism += int(not (torch.equal(dw, dw_ref) and torch.equal(db, db_ref)))
report("k_wgrad<128,256> (pipelined chunk loop) + bias sums", max(1, N_FWD // 3), mism)
del x, w, mask, dx0, ref, y, ws

# the persistent network kernel: a 128x128 frame with both nets at 256x8, rendered repeatedly
import bench  # noqa: E402
bench.ARCH = (8, 256, 8, 256)
render, kw, _ = bench.build_product(torch.device(dev))
from mofanerf_amd import synth  # noqa: E402
bm, tex, exp = (t.to(dev) for t in synth.codes(0))
Kc = synth.intrinsics(128, 128)
c2w = bench.pose_spherical(20.0, 0.0, 16.0)[:3, :4].contiguous().to(dev)
with torch.no_grad():
    def frame():
        rgb, disp, acc, _ = render.render_fitting(128, 128, Kc, chunk=196608, c2w=c2w, shapeCodes=bm, uvCodes=tex, expType=20,
                                                  expCodes=exp, **kw)
        return torch.cat([rgb.reshape(-1, 3), acc.reshape(-1, 1)], 1)
    ref = frame()
    mism = 0
    n = max(1, N_FWD // 10)
    for _ in range(n):
        mism += int(not torch.equal(frame(), ref))
report("k_mlp_fused (128x128 frame, 256x8 + 256x8)", n, mism)

# the same kernel walking SEVERAL 256-feature blocks per layer (MOFA_FUSED=1 at width 512: the epilogue's LDS windows of one block
# against the next block's first operand fetch — the barrier added in round 3), and the shipped fine net with its first layer
# through encoding panels
os.environ["MOFA_FUSED"] = "1"
lib.reload_env()
bench.ARCH = (8, 256, 10, 512)
render, kw, _ = bench.build_product(torch.device(dev))
with torch.no_grad():
    ref = frame()
    mism = sum(int(not torch.equal(frame(), ref)) for _ in range(n))
report("k_mlp_fused, four/two feature blocks per layer (128x128 frame, 256x8 + 512x10, MOFA_FUSED=1)", n, mism)
del os.environ["MOFA_FUSED"]
lib.reload_env()
bench.ARCH = (8, 256, 10, 1024)
render, kw, _ = bench.build_product(torch.device(dev))
with torch.no_grad():
    ref = frame()
    mism = sum(int(not torch.equal(frame(), ref)) for _ in range(max(1, n // 3)))
report("shipped sizes (128x128 frame, 256x8 + 1024x10: encoding panels + K = 64 first layer, per-layer launches)", max(1, n // 3), mism)
sys.exit(1 if bad else 0)
