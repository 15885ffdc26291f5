"""The measurement library (build_arms/libmofanerf_measure.so, built by tools/build_measure.py from csrc/measure/) — NOT part of
the product.  Its arms instantiate the same layer-kernel source under other policies or are the rejected scheduling twins earlier
rounds measured; every arm that computes results must reproduce the product's `mofa_layer_forward` bit for bit (they only move
instructions around), the time-stamp arm must leave a stamp per workgroup, and the product library must not export any of it.
Skipped when the library has not been built (the driver's test run builds the product only)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mofanerf_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import build_measure  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(build_measure.OUT), reason="measurement library not built (python tools/build_measure.py)")]
DEV = "cuda"


def _setup(M, K, N, k2=0, S=0, seed=0):
    rng = np.random.default_rng(seed)
    t = lambda *sh: torch.from_numpy(rng.normal(size=sh).astype(np.float32)).to(DEV)
    x1, x2 = t(M * K), (t(M * k2) if k2 else None)                      # raw panel buffers (any contents are valid panels)
    w, b = t(N * (K + k2)) / 16, t(N)
    rows = t(M // S, N) if S else None
    return x1, x2, w, b, rows


def _args(x1, K, x2, k2, w, b, rows, S, y, M, N):
    bias = rows if rows is not None else b
    return (lib.ptr(x1), K, lib.ptr(x2), k2, lib.ptr(w), lib.ptr(bias), S, (rows.shape[0] if rows is not None else 1), lib.ptr(y), M, N, 1,
            lib.stream())


CASES = [(256 * 131, 256, 1024, 0, 0), (256 * 67, 256, 1024, 128, 0), (256 * 40, 256, 128, 0, 64), (256 * 24, 1024, 256, 0, 0)]


@pytest.mark.parametrize("arm", ["shipped", "plain", "unstaged", "gap2", "gap3", "setprio1", "setprio3", "waves3", "ring3", "persist",
                                 "persist_dephase", "persist_pipe", "timeline", "bn64"])
def test_correct_arms_reproduce_the_product_bit_for_bit(arm):
    Lm = build_measure.load()
    assert arm in Lm.mofa_measure_arms().decode().split(",")
    Lp = lib.load()
    for ci, (M, K, N, k2, S) in enumerate(CASES):
        if arm == "persist_pipe" and S:
            continue                                                     # (that twin has no per-ray-bias form)
        x1, x2, w, b, rows = _setup(M, K, N, k2, S, seed=ci)
        ref = torch.full((M * N,), float("nan"), device=DEV)
        lib.check(Lp.mofa_layer_forward(*_args(x1, K, x2, k2, w, b, rows, S, ref, M, N)), "product layer")
        for _ in range(2):
            y = torch.full((M * N,), float("nan"), device=DEV)
            build_measure.check(Lm, Lm.mofa_measure_layer_forward(arm.encode(), *_args(x1, K, x2, k2, w, b, rows, S, y, M, N)), arm)
            torch.cuda.synchronize()
            assert torch.equal(y, ref), (arm, M, K, N, k2, S)


def test_timeline_arm_leaves_a_stamp_per_workgroup_and_sink_arm_is_refused_as_a_result():
    Lm = build_measure.load()
    M, K, N = 256 * 64, 1024, 1024
    tiles = (M // 256) * (N // 128)
    x1, _, w, b, _ = _setup(M, K, N)
    y = torch.empty(M * N, device=DEV)
    tl = torch.zeros(tiles * (8 + 64), dtype=torch.int64, device=DEV)
    build_measure.check(Lm, Lm.mofa_measure_set_timeline(tl.data_ptr()), "set_timeline")
    build_measure.check(Lm, Lm.mofa_measure_layer_forward(b"timeline", *_args(x1, K, None, 0, w, b, None, 0, y, M, N)), "timeline")
    build_measure.check(Lm, Lm.mofa_measure_set_timeline(None), "set_timeline(off)")
    torch.cuda.synchronize()
    t = tl[: tiles * 8].reshape(tiles, 8).cpu().numpy()
    assert (t[:, 0] > 0).all() and (t[:, 1] >= t[:, 0]).all() and (t[:, 2] > t[:, 1]).all() and (t[:, 3] >= t[:, 2]).all()
    pan = tl[tiles * 8:].reshape(tiles, 64).cpu().numpy()
    assert (pan[:, : K // 16 - 1] > 0).all()                              # one stamp per panel barrier
    # the ablation arm computes nothing: its output must NOT be mistaken for a result (left untouched)
    y.fill_(7.0)
    build_measure.check(Lm, Lm.mofa_measure_layer_forward(b"sink_epilogue", *_args(x1, K, None, 0, w, b, None, 0, y, M, N)), "sink")
    torch.cuda.synchronize()
    assert bool((y == 7.0).all())
    assert Lm.mofa_measure_layer_forward(b"no_such_arm", *_args(x1, K, None, 0, w, b, None, 0, y, M, N)) != 0


def test_product_library_exports_no_measurement_symbols():
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert names and all(n.startswith("mofa_") for n in names)
    bad = [n for n in names if any(k in n for k in ("internal", "measure", "timeline", "probe", "persist", "ring3", "ablate"))]
    assert not bad, bad
    assert sorted(names) == sorted(lib.SIGNATURES)                       # exactly the C ABI of include/mofanerf_hip.h
