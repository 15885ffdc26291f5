#!/usr/bin/env python3
"""Does the vector pipe issue packed-fp32 FMAs in the shadow of the fp32 MFMAs on MI355X?  (measurement library: k_mfma_valu_probe<V> —
V independent v_pk_fma_f32 behind every v_mfma_f32_32x32x2_f32 of the pure-MFMA stream; no memory traffic.)

    python tools/probe_dual_issue.py            # the sweep: V = 0 .. 16, two and one wave per SIMD
    python tools/probe_dual_issue.py hold V S   # run V for about S seconds (under tools/clock_probe.sh: clock and socket power)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mofanerf_amd import lib  # noqa: E402
import build_measure  # noqa: E402

build_measure.build(verbose=False)
Lm = build_measure.load()
f = Lm.mofa_measure_mfma_valu_probe
out = torch.zeros(16, device="cuda")
PEAK = 157.3


def timed(blocks, iters, V, reps=5):
    for _ in range(2):
        build_measure.check(Lm, f(lib.ptr(out), blocks, iters, V, lib.stream()), "probe")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        build_measure.check(Lm, f(lib.ptr(out), blocks, iters, V, lib.stream()), "probe")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    mfma = blocks * 4 * iters * 64 * 4096.0 / (ms * 1e-3) / 1e12
    valu = blocks * 4 * iters * 64 * V * 256.0 / (ms * 1e-3) / 1e12
    return ms, mfma, valu


if len(sys.argv) > 1 and sys.argv[1] == "hold":
    V, secs = int(sys.argv[2]), float(sys.argv[3])
    ms, mfma, valu = timed(512, 4096, V, reps=2)
    n = max(1, int(secs * 1e3 / ms))
    t0 = time.perf_counter()
    ms, mfma, valu = timed(512, 4096, V, reps=n)
    print(f"hold V={V}: {n} launches of {ms:.2f} ms in {time.perf_counter() - t0:.1f} s: MFMA {mfma:.2f} + vector {valu:.2f} = {mfma + valu:.2f} TFLOP/s "
          f"({(mfma + valu) / PEAK * 100:.1f} % of the fp32 matrix peak)", flush=True)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "kinds":
    # what ONE instruction of each kind costs the matrix pipe when it rides behind every MFMA: cycles added per MFMA (64 = one MFMA)
    g = Lm.mofa_measure_mfma_kind_probe
    KINDS = ["v_pk_fma_f32", "v_fma_f32", "v_add_u32", "v_mov_b32", "v_lshl_add_u32", "ds_read_b128", "s_add_u32", "v_pk_fma_f32, 8 V clustered behind every 8th MFMA"]

    def timed_kind(blocks, iters, V, kind, reps=5):
        for _ in range(2):
            build_measure.check(Lm, g(lib.ptr(out), blocks, iters, V, kind, lib.stream()), "probe")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            build_measure.check(Lm, g(lib.ptr(out), blocks, iters, V, kind, lib.stream()), "probe")
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for blocks, wps in ((512, 2), (256, 1)):
        base, _, _ = timed(blocks, 2048, 0)
        per_mfma0 = base * 1e-3 / (2048 * 64 * (2 if wps == 2 else 1)) * 2.39e9     # cycles per MFMA per SIMD stream (two waves share a SIMD)
        print(f"\n{wps} wave(s) per SIMD; pure MFMA stream: {base:.3f} ms = {per_mfma0:.1f} cycles per MFMA at 2.39 GHz\n")
        print("| instruction behind every MFMA | 1 per MFMA: cycles added per MFMA | 2 per MFMA | 4 per MFMA |\n|---|---|---|---|")
        for kind, name in enumerate(KINDS):
            cells = []
            for V in (1, 2, 4):
                ms = timed_kind(blocks, 2048, V, kind)
                cells.append(f"{(ms / base - 1.0) * per_mfma0:+.1f}")
            print(f"| `{name}` | " + " | ".join(cells) + " |", flush=True)
    sys.exit(0)

print("| V (v_pk_fma_f32 per MFMA) | waves / SIMD | ms | MFMA TFLOP/s | vector TFLOP/s | sum | sum / 157.3 |\n|---|---|---|---|---|---|---|")
for blocks, wps in ((512, 2), (256, 1)):
    for V in (0, 1, 2, 4, 6, 8, 12, 15, 16):
        ms, mfma, valu = timed(blocks, 2048, V)
        print(f"| {V} | {wps} | {ms:.3f} | {mfma:.2f} | {valu:.2f} | {mfma + valu:.2f} | {(mfma + valu) / PEAK:.3f} |", flush=True)
