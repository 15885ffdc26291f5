#!/usr/bin/env python3
"""Micro-benchmark of the MFMA layer kernel through the C ABI: TFLOP/s of mofa_layer_forward at the shipped shapes (product
library), and of the measurement arms of build_arms/libmofanerf_measure.so (tools/build_measure.py; `run(..., arm="ring3")`,
`--arms`, `--persist`, `--peak`).  Prints one line per case."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import lib  # noqa: E402

L = lib.load()
dev = "cuda"


def measure_lib():
    global _measure
    if _measure is None:
        import build_measure
        build_measure.build(verbose=False)
        _measure = build_measure.load()
    return _measure


def run(M, K, N, k2=0, iters=10, arm=None):
    """`arm` = None: the product's mofa_layer_forward; otherwise the named arm of the measurement library."""
    x = torch.randn(M * K, device=dev)
    x2 = torch.randn(M * k2, device=dev) if k2 else None
    w = torch.randn(N * (K + k2), device=dev) * 0.03
    b = torch.randn(N, device=dev)
    y = torch.empty(M * N, device=dev)
    st = lib.stream()
    args = (lib.ptr(x), K, lib.ptr(x2), k2, lib.ptr(w), lib.ptr(b), 0, 1, lib.ptr(y), M, N, 1, st)
    if arm is None:
        launch = lambda: lib.check(L.mofa_layer_forward(*args), "layer")
    else:
        import build_measure
        Lm = measure_lib()
        launch = lambda: build_measure.check(Lm, Lm.mofa_measure_layer_forward(arm.encode(), *args), arm)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * (K + k2) * N / (ms * 1e-3) / 1e12


def run_bwd(M, K, N, iters=10):
    """Backward-data launch dX[M, N] = G[M, K] @ Wt (+ accumulate, ReLU mask): k_layer<128,..,BWD> on raw panel buffers."""
    g = torch.randn(M * K, device=dev)
    wt = torch.randn(N * K, device=dev) * 0.03
    mask = torch.randn(M * N, device=dev)
    dx = torch.zeros(M * N, device=dev)
    st = lib.stream()
    args = (lib.ptr(g), K, lib.ptr(wt), lib.ptr(mask), 1, lib.ptr(dx), M, N, st)
    for _ in range(3):
        lib.check(L.mofa_layer_backward_data(*args), "bwd")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.check(L.mofa_layer_backward_data(*args), "bwd")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * K * N / (ms * 1e-3) / 1e12


if __name__ == "__main__" and "--peak" in sys.argv:
    f = measure_lib().mofa_measure_mfma_peak_probe
    out = torch.zeros(16, device=dev)
    iters = 4096
    for blocks, rnd, label in ((512, 0, "constant operands, 2 waves/SIMD (512 workgroups = one round)"), (256, 0, "constant operands, 1 wave/SIMD"),
                               (6144, 0, "constant operands, 2 waves/SIMD, 12 rounds of workgroups"),
                               (512, 1, "RANDOM operands re-scrambled every 64 MFMAs, 2 waves/SIMD"), (256, 1, "RANDOM operands, 1 wave/SIMD"),
                               (512, 2, "control: same instruction stream, operands unchanged, 2 waves/SIMD"),
                               (256, 2, "control: same instruction stream, operands unchanged, 1 wave/SIMD")):
        it = iters if blocks <= 512 else iters // 12
        for _ in range(2):
            lib.check(f(lib.ptr(out), blocks, it, rnd, lib.stream()), "probe")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.check(f(lib.ptr(out), blocks, it, rnd, lib.stream()), "probe")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        flops = blocks * 4 * it * 64 * 4096.0
        print(f"pure fp32-MFMA loop, {label}: {ms:7.3f} ms  {flops / (ms * 1e-3) / 1e12:7.2f} TFLOP/s "
              f"({flops / (ms * 1e-3) / 1e12 / 157.3 * 100:5.1f}% of 157.3)", flush=True)
    sys.exit(0)

def ab(arms, cases, rounds=3, iters=20):
    """Interleaved A/B in ONE process: every round runs every arm once, so that clock / thermal drift hits all of them alike."""
    for (M, K, N, k2) in cases:
        res = {a: [] for a in arms}
        for _ in range(rounds):
            for a in arms:
                res[a].append(run(M, K, N, k2, iters=iters, arm=None if a == "product" else a)[1])
        for a in arms:
            v = res[a]
            print(f"M={M:7d} K={K + k2:5d} N={N:5d} {a:18s}: " + " ".join(f"{t:7.2f}" for t in v) +
                  f"  TFLOP/s (best {max(v):7.2f} = {max(v) / 157.3 * 100:5.1f}% of fp32 MFMA peak)", flush=True)


if __name__ == "__main__" and "--persist" in sys.argv:
    ab(["product", "plain", "persist", "persist_dephase"], ((196608, 1024, 1024, 0), (196608, 1024, 1024, 1024), (196608, 256, 256, 0), (131072, 1024, 1024, 0)))
    sys.exit(0)

if __name__ == "__main__" and "--arms" in sys.argv:
    i = sys.argv.index("--arms")
    names = sys.argv[i + 1].split(",") if len(sys.argv) > i + 1 else measure_lib().mofa_measure_arms().decode().split(",")
    ab(["product"] + [n for n in names if n != "timeline"], ((196608, 1024, 1024, 0), (196608, 256, 256, 0), (32768, 1024, 1024, 0)))
    sys.exit(0)

if __name__ == "__main__":
    cases = [(196608, 1024, 1024, 0), (196608, 1024, 1024, 1024), (196608, 1024, 512, 0), (196608, 256, 256, 0),
             (196608, 256, 256, 256), (32768, 1024, 1024, 0), (65536, 64, 64, 0)]
    for (M, K, N, k2) in cases:
        ms, tf = run(M, K, N, k2)
        print(f"M={M:7d} K={K + k2:5d} N={N:5d}: {ms:8.3f} ms  {tf:7.2f} TFLOP/s  ({tf / 157.3 * 100:5.1f}% of fp32 MFMA peak)", flush=True)
    ms, tf = run_bwd(196608, 1024, 1024)
    print(f"backward-data M=196608 K=N=1024: {ms:8.3f} ms  {tf:7.2f} TFLOP/s", flush=True)
