#!/usr/bin/env python3
"""Per-workgroup TIMELINE of the dominant kernel: the "timeline" arm of the measurement library (the shipped layer kernel source
under a policy whose hooks write time stamps; `python tools/build_measure.py`, then `python tools/timeline_layer.py`).

Every workgroup of one launch of `k_layer<128,false,true>` at the shipped fine-network shape (M = 196608 points, K = N = 1024:
6,144 tiles = 12 rounds of the 512 resident workgroups) records four `wall_clock64()` stamps (100 MHz) — entry, first operand
panel landed (K loop starts), K loop done, epilogue stores issued — and the HW_ID / XCC_ID of its wave 0.  From those:
  * how long prologue / K loop / epilogue take, per round;
  * per CU, the share of the launch during which 2, 1 or 0 of its resident workgroups are INSIDE their K loop (only then does
    the CU's matrix pipe have work) — the direct measure of what the tile boundaries cost;
  * the slot turnaround (a workgroup's last stamp -> the entry stamp of the workgroup that takes its place on that CU);
  * how synchronous the chip is: spread of the K-loop end times within a round.
Prints a markdown summary."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import lib  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_measure  # noqa: E402

build_measure.build(verbose=False)
Lm = build_measure.load()


ARM = sys.argv[sys.argv.index("--arm") + 1].encode() if "--arm" in sys.argv else b"timeline"     # timeline | timeline_sink


def layer(*a):
    """the stamped kernel (stamps are written only while a buffer is set)"""
    return Lm.mofa_measure_layer_forward(ARM, *a)

M, K, N = 196608, 1024, 1024
tiles = (M // 256) * (N // 128)
dev = "cuda"
x = torch.randn(M * K, device=dev)
w = torch.randn(N * K, device=dev) * 0.03
b = torch.randn(N, device=dev)
y = torch.empty(M * N, device=dev)
args = (lib.ptr(x), K, None, 0, lib.ptr(w), lib.ptr(b), 0, 1, lib.ptr(y), M, N, 1, lib.stream())
def timed(n):
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(n):
        build_measure.check(Lm, layer(*args), "timeline layer")
    a1.record()
    torch.cuda.synchronize()
    return a0.elapsed_time(a1) / n


timed(20)                                   # clocks and caches in steady state
ms_plain = timed(20)                        # the same binary with the stamps off (null pointer)
PAN = 64                                    # per-panel stamps of the pipelined loop follow the 8 per-tile words (builds that have them)
tl = torch.zeros(tiles * (8 + PAN), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
# the stamped launch sits in the MIDDLE of a back-to-back stream of launches: an isolated launch after a host synchronisation
# runs at a reduced engine clock (measured with these very stamps: 1.9 GHz instead of 2.3-2.4) while the power state ramps up
for _ in range(20):
    build_measure.check(Lm, layer(*args), "timeline layer")
Lm.mofa_measure_set_timeline(C.c_void_p(tl.data_ptr()))      # host-side switch, read at launch time
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
build_measure.check(Lm, layer(*args), "timeline layer")
e1.record()
Lm.mofa_measure_set_timeline(None)
for _ in range(5):
    build_measure.check(Lm, layer(*args), "timeline layer")
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
raw = tl.cpu().numpy()
t = raw[:tiles * 8].reshape(tiles, 8)
pan = raw[tiles * 8:].reshape(tiles, PAN)
assert (t[:, 0] > 0).all(), "some workgroups left no stamp"
T0, T1 = t[:, 0].min(), t[:, 3].max()
tick_us = ms * 1e3 / float(T1 - T0)            # the stamps span (almost) the whole launch: calibrates the 100 MHz clock
us = lambda d: np.asarray(d, np.float64) * tick_us
ent, k0, k1, end = (t[:, i] - T0 for i in range(4))
cu = (t[:, 5] & 0xF) << 8 | ((t[:, 4] >> 8) & 0xFF)          # (XCC, SE, SH, CU)
order = np.argsort(ent, kind="stable")
print(f"# Timeline of `k_layer<128,false,true>` — M={M}, K=N={K}: {tiles} workgroups, launch {ms * 1e3:.0f} us by HIP events "
      f"({2.0 * M * K * N / (ms * 1e-3) / 1e12:.1f} TFLOP/s with the stamps on), stamp span {float(T1 - T0) * 1e-2:.0f} us at a nominal 100 MHz "
      f"(calibrated tick {tick_us * 1e3:.2f} ns); the same binary with the stamps off: {ms_plain * 1e3:.0f} us = "
      f"{2.0 * M * K * N / (ms_plain * 1e-3) / 1e12:.1f} TFLOP/s\n")
ck = t[:, 6].astype(np.float64) / np.maximum(us(k1 - k0), 1e-9)        # clock64 ticks per microsecond inside the K loop
print(f"clock64() (s_memtime) ticks per us inside the K loops: median {np.median(ck):.1f}, p5 {np.percentile(ck, 5):.1f}, p95 "
      f"{np.percentile(ck, 95):.1f}  (2400 would be the 2.4 GHz the 157.3 TFLOP/s peak assumes, 100 the constant reference clock)\n")
print(f"distinct CUs seen: {len(np.unique(cu))}; workgroups per CU: min {np.bincount(np.unique(cu, return_inverse=True)[1]).min()} "
      f"max {np.bincount(np.unique(cu, return_inverse=True)[1]).max()}\n")
print("| round (512 workgroups by entry time) | entry spread us | first fetch us (median / p95) | K loop us (median / p95) | "
      "epilogue issue us (median / p95) | K-loop END spread within the round us (p5..p95) |\n|---|---|---|---|---|---|")
for r in range(0, tiles, 512):
    g = order[r:r + 512]
    pro, kl, ep = us(k0[g] - ent[g]), us(k1[g] - k0[g]), us(end[g] - k1[g])
    ke = us(k1[g])
    print(f"| {r // 512} | {us(ent[g].max() - ent[g].min()):.1f} | {np.median(pro):.1f} / {np.percentile(pro, 95):.1f} | {np.median(kl):.1f} / "
          f"{np.percentile(kl, 95):.1f} | {np.median(ep):.1f} / {np.percentile(ep, 95):.1f} | {np.percentile(ke, 95) - np.percentile(ke, 5):.1f} |")
# per-CU occupancy of "inside the K loop"
span = float(T1 - T0)
occ = np.zeros(4)
gaps = []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    ev = sorted([(k0[i], 1) for i in idx] + [(k1[i], -1) for i in idx])
    cur, last = 0, 0.0
    for tt, d in ev:
        occ[min(cur, 3)] += tt - last
        last, cur = tt, cur + d
    occ[min(cur, 3)] += span - last
    # slot turnaround: each entry (after the first two) takes the slot of the latest workgroup that ended before it
    ends = sorted(end[idx])
    starts = sorted(ent[idx])
    used = 0
    for s0 in starts[2:]:
        cand = [e for e in ends[used:] if e <= s0]
        if cand:
            gaps.append(us(s0 - cand[0]))
            used += 1
occ /= occ.sum()
# how fast does a workgroup's K loop run as a function of how many OTHER workgroups of its CU are inside theirs?  Regress the
# K-loop duration on the time it spent with 0 / 1 / 2 partners: duration = sum_k share_k * T_k  (T_k = K-loop time if always k partners)
rows, rhs = [], []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    for i in idx:
        others = [j for j in idx if j != i]
        pts = sorted({k0[i], k1[i]} | {k0[j] for j in others if k0[i] < k0[j] < k1[i]} | {k1[j] for j in others if k0[i] < k1[j] < k1[i]})
        share = np.zeros(3)
        for a_, b_ in zip(pts[:-1], pts[1:]):
            mid = 0.5 * (a_ + b_)
            n_oth = sum(1 for j in others if k0[j] <= mid < k1[j])
            share[min(n_oth, 2)] += b_ - a_
        rows.append(share / max(1.0, float(k1[i] - k0[i])))
        rhs.append(1.0)
A = np.array(rows)
# progress rate model: 1 tile = sum_k (time with k partners) * rate_k  ->  solve for rate_k (tiles per tick), report relative to rate_alone
dur = np.array([float(k1[i] - k0[i]) for c in np.unique(cu) for i in np.where(cu == c)[0]])
T = A * dur[:, None]
rate, *_ = np.linalg.lstsq(T, np.ones(len(dur)), rcond=None)
ideal_alone = 2 * 256 * 128 * K / (157.3e12 / 256) * 1e6          # us per tile with the whole CU at peak
with np.errstate(divide="ignore"):
    print("\nK-loop progress rate of ONE workgroup by number of partner workgroups of its CU that are inside their K loop (least squares over all tiles): "
          + ", ".join(f"{k} partners: {tick_us / rate[k]:.0f} us per tile = {ideal_alone / (tick_us / rate[k]) * 100:.0f} % of the CU's peak "
                      f"({T[:, k].sum() / T.sum() * 100:.0f} % of all K-loop time)"
                      for k in range(3) if rate[k] > 0 and T[:, k].sum() > 0.005 * T.sum()))
print(f"\nPer CU, share of the launch with k resident workgroups INSIDE their K loop (average over {len(np.unique(cu))} CUs): "
      f"2: **{occ[2] * 100:.1f} %**, 1: **{occ[1] * 100:.1f} %**, 0: **{occ[0] * 100:.1f} %**" + (f", >2: {occ[3] * 100:.1f} %" if occ[3] > 0 else ""))
kl_all = us(k1 - k0)
print(f"\nK loop: median {np.median(kl_all):.1f} us per tile (two workgroups sharing a CU: ideal 2 x 2*256*128*{K} FLOP / (157.3e12/256 FLOP/s) = "
      f"{2 * 2 * 256 * 128 * K / (157.3e12 / 256) * 1e6:.1f} us); per tile prologue {np.median(us(k0 - ent)):.1f} us, epilogue issue {np.median(us(end - k1)):.1f} us; "
      f"slot turnaround (previous workgroup's last stamp -> successor's entry) median {np.median(gaps):.1f} us, p95 {np.percentile(gaps, 95):.1f} us "
      f"({len(gaps)} hand-overs)")
tile_period = np.median(kl_all) + np.median(us(k0 - ent)) + np.median(us(end - k1)) + np.median(gaps)
print(f"\nTile period = K loop + prologue + epilogue issue + turnaround = {tile_period:.1f} us -> 12 rounds = {12 * tile_period:.0f} us of the {ms * 1e3:.0f} us launch; "
      f"K-loop share {np.median(kl_all) / tile_period * 100:.1f} %.")

if "--panels" in sys.argv and (pan[:, 0] > 0).all():
    # How long does ONE panel of a workgroup's K loop take, by what the OTHER slot of its CU is doing at that moment?
    npan = K // 16 - 1                                              # one stamp per workgroup barrier = per panel (slot 63: panel 0 landed)
    pst = pan[:, :npan] - T0
    cls_names = ["partner in its K loop", "partner fetching its first panels", "partner in its epilogue", "other slot empty"]
    tot = np.zeros(4); cnt = np.zeros(4)
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        for i in idx:
            others = [j for j in idx if j != i]
            edges = np.concatenate([[k0[i]], pst[i]])
            for a_, b_ in zip(edges[:-1], edges[1:]):
                mid = 0.5 * (a_ + b_)
                state = 3
                for j in others:
                    if ent[j] <= mid < end[j]:
                        if mid < pst[j][0]: state = 1
                        elif mid < k1[j]: state = 0
                        else: state = 2
                        break
                tot[state] += b_ - a_
                cnt[state] += 1
    print("\nPer-panel time of a workgroup's K loop (16 of K; two workgroups sharing a CU at 100 %: "
          f"{2 * 2 * 256 * 128 * 16 / (157.3e12 / 256) * 1e6:.2f} us, one alone: {2 * 256 * 128 * 16 / (157.3e12 / 256) * 1e6:.2f} us) by the state of the CU's other slot:\n")
    print("| other slot | panels | share of K-loop time | mean us per panel |\n|---|---|---|---|")
    for k in range(4):
        if cnt[k]:
            print(f"| {cls_names[k]} | {int(cnt[k])} | {tot[k] / tot.sum() * 100:.1f} % | {us(tot[k] / cnt[k]):.2f} |")
    landed = us(pan[:, 63] - T0 - k0)
    print(f"\nFirst fetch: K-loop entry -> panel 0 landed in every wave (barrier passed): median {np.median(landed):.2f} us, p5 {np.percentile(landed, 5):.2f}, "
          f"p95 {np.percentile(landed, 95):.2f}; by round: " + ", ".join(f"{np.median(landed[order[r:r + 512]]):.1f}" for r in range(0, tiles, 512)))
    first = us(pst[:, 0] - k0)
    print(f"\nStart of a tile: K-loop entry (before the first two panel requests) -> first workgroup barrier passed (panel 0's first half computed, panel 1 landed): "
          f"median {np.median(first):.2f} us, p5 {np.percentile(first, 5):.2f}, p95 {np.percentile(first, 95):.2f}; then per panel: "
          + ", ".join(f"{np.median(us(pst[:, i + 1] - pst[:, i])):.2f}" for i in range(6)) + " us (medians of panels 1..6), last three: "
          + ", ".join(f"{np.median(us(pst[:, i + 1] - pst[:, i])):.2f}" for i in range(npan - 4, npan - 1))
          + f"; last barrier -> K loop done: median {np.median(us(k1 - pst[:, npan - 1])):.2f} us")
