#!/usr/bin/env python3
"""Build the MEASUREMENT library build_arms/libmofanerf_measure.so from mofanerf_amd/csrc/measure/mofa_measure.hip (gfx950).

It holds what is deliberately NOT in libmofanerf_hip.so: the layer kernel under non-shipped policies (scheduling arms, time
stamps, an epilogue ablation that produces no results), the rejected ring3 / persistent twins and the pure-MFMA probe.  Only
tools/ (microbench_layer.py, ab_layer.py, timeline_layer.py) and tests/test_gpu_measure_arms.py load it; the product never does.

    python tools/build_measure.py [--force]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mofanerf_amd import build as pbuild

SRC = os.path.join(pbuild.CSRC, "measure", "mofa_measure.hip")
OUT = os.path.join(ROOT, "build_arms", "libmofanerf_measure.so")

_f = C.c_void_p
LAYER_ARGS = [C.c_char_p, _f, C.c_int32, _f, C.c_int32, _f, _f, C.c_int32, C.c_int64, _f, C.c_int64, C.c_int32, C.c_int32, _f]


def build(force=False, verbose=True):
    deps = [SRC, os.path.join(pbuild.CSRC, "mofa_layer.h"), os.path.join(pbuild.CSRC, "mofa_common.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [pbuild.hipcc()] + pbuild.FLAGS + [SRC, "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


def load():
    """ctypes handle with the three entry points typed; raises if the library has not been built."""
    if not os.path.exists(OUT):
        raise RuntimeError(f"{OUT} is missing: python tools/build_measure.py")
    L = C.CDLL(OUT)
    L.mofa_measure_arms.restype = C.c_char_p
    L.mofa_measure_last_error.restype = C.c_char_p
    L.mofa_measure_layer_forward.restype, L.mofa_measure_layer_forward.argtypes = C.c_int, LAYER_ARGS
    L.mofa_measure_set_timeline.restype, L.mofa_measure_set_timeline.argtypes = C.c_int, [_f]
    L.mofa_measure_mfma_kind_probe.restype = C.c_int
    L.mofa_measure_mfma_kind_probe.argtypes = [_f, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f]
    L.mofa_measure_mfma_valu_probe.restype = C.c_int
    L.mofa_measure_mfma_valu_probe.argtypes = [_f, C.c_int32, C.c_int32, C.c_int32, _f]
    L.mofa_measure_xcc_watch.restype = C.c_int
    L.mofa_measure_xcc_watch.argtypes = [_f, C.c_int32, C.c_uint64, _f]
    L.mofa_measure_mfma_peak_probe.restype = C.c_int
    L.mofa_measure_mfma_peak_probe.argtypes = [_f, C.c_int32, C.c_int32, C.c_int32, _f]
    return L


def check(L, rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {L.mofa_measure_last_error().decode()}")


if __name__ == "__main__":
    print("built", build(force="--force" in sys.argv))
