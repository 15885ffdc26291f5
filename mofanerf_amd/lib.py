"""ctypes binding of ``libmofanerf_hip.so`` (the C ABI in ``include/mofanerf_hip.h``).

The library is built in-tree by ``mofanerf_amd/build.py`` (``__graft_entry__.build()``).  There is no
fallback of any kind: if the shared object is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOFA_LIB") or os.path.join(_HERE, "libmofanerf_hip.so")   # MOFA_LIB: A/B builds (tools/)

_f = C.POINTER(C.c_float)
_fp = C.c_void_p      # device pointers travel as integers
_i32, _i64, _sz = C.c_int32, C.c_int64, C.c_size_t


ABI_VERSION = 5


class NetShape(C.Structure):
    """``MofaNetShape``: depth / width plus the encoding and code widths the reference's flags fix (tools/config_parser.py:51-56,
    113-118); the defaults are the shipped configuration (configs/exp_mofanerf.txt)."""
    _fields_ = [("D", C.c_int32), ("W", C.c_int32), ("pe_point_freqs", C.c_int32), ("pe_view_freqs", C.c_int32),
                ("ch_exp", C.c_int32), ("ch_shape", C.c_int32), ("ch_tex", C.c_int32)]

    def __init__(self, D, W, pe_point_freqs=10, pe_view_freqs=4, ch_exp=30, ch_shape=50, ch_tex=256):
        super().__init__(int(D), int(W), int(pe_point_freqs), int(pe_view_freqs), int(ch_exp), int(ch_shape), int(ch_tex))

    def __repr__(self):
        return (f"NetShape(D={self.D}, W={self.W}, multires={self.pe_point_freqs}, multires_views={self.pe_view_freqs}, "
                f"ch_exp={self.ch_exp}, ch_shape={self.ch_shape}, ch_tex={self.ch_tex})")


# name -> (restype, argtypes); mirrors include/mofanerf_hip.h one to one
SIGNATURES = {
    "mofa_abi_version": (C.c_int, []),
    "mofa_last_error": (C.c_char_p, []),
    "mofa_config_reload": (C.c_int, []),
    "mofa_test_hooks": (C.c_int, [C.c_uint32, _i32, _i32]),
    "mofa_device_init": (C.c_int, [_fp, C.POINTER(_i32), C.POINTER(_i32)]),
    "mofa_net_num_layers": (C.c_int, [NetShape]),
    "mofa_net_layer_dims": (C.c_int, [NetShape, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "mofa_pe_k_padded": (C.c_int, [_i32]),
    "mofa_net_packed_floats": (_sz, [NetShape]),
    "mofa_net_folded_floats": (_sz, [NetShape]),
    "mofa_net_workspace_floats": (_sz, [NetShape, _i64, _i64]),
    "mofa_net_pack": (C.c_int, [NetShape, C.POINTER(_fp), _fp, _fp]),
    "mofa_net_fold": (C.c_int, [NetShape, C.POINTER(_fp), C.POINTER(_fp), _fp, _fp, _fp, _fp, _fp]),
    "mofa_net_forward": (C.c_int, [NetShape, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i64, _fp, _fp, _i64, _i32, _fp, _fp,
                                   _fp, _fp, _fp, _fp, _fp]),
    "mofa_net_packed_t_floats": (_sz, [NetShape]),
    "mofa_net_tape_floats": (_sz, [NetShape, _i64]),
    "mofa_net_mask_tape_words": (_sz, [NetShape, _i64]),
    "mofa_net_backward_workspace_floats": (_sz, [NetShape, _i64, _i32]),
    "mofa_net_pack_t": (C.c_int, [NetShape, C.POINTER(_fp), _fp, _fp]),
    "mofa_net_backward": (C.c_int, [NetShape, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i64, _fp, _i64, _i32, _fp, _fp, _fp, _fp,
                                    _fp, _fp, C.POINTER(_fp), _fp, _fp]),
    "mofa_weight_grad_workspace_floats": (_sz, [_i64, _i32, _i32]),
    "mofa_weight_grad": (C.c_int, [_fp, _i32, _fp, _i32, _i64, _i64, _i32, _i32, _fp, _i32, _i32, _fp, _fp, _fp]),
    "mofa_head_weight_grad": (C.c_int, [_fp, _i32, _i32, _fp, _i32, _i64, _i64, _i32, _fp, _i32, _fp]),
    "mofa_pe_panels": (C.c_int, [_fp, _fp, _fp, _i64, _fp, _i64, _i32, _i32, _i64, _fp, _fp]),
    "mofa_pack_panels_t": (C.c_int, [_fp, _i32, _i32, _i32, _i32, _fp, _i32, _i32, _fp]),
    "mofa_layer_backward_data": (C.c_int, [_fp, _i32, _fp, _fp, _i32, _fp, _i64, _i32, _fp]),
    "mofa_layer_backward_data_bits": (C.c_int, [_fp, _i32, _fp, _fp, _i32, _fp, _i64, _i32, _fp]),
    "mofa_head_backward": (C.c_int, [_fp, _i32, _i32, _fp, _i32, _fp, _i32, _fp, _i64, _i64, _fp]),
    "mofa_head_backward_bits": (C.c_int, [_fp, _i32, _i32, _fp, _i32, _fp, _i32, _fp, _i64, _i64, _fp]),
    "mofa_bias_grad": (C.c_int, [_fp, _i64, _i64, _i32, _fp, _fp]),
    "mofa_bias_grad_rays": (C.c_int, [_fp, _i64, _i64, _i32, _i32, _fp, _fp]),
    "mofa_pe_backward": (C.c_int, [_fp, _i64, _fp, _fp, _fp, _i64, _i64, _i32, _i32, _fp, _fp, _fp]),
    "mofa_pe_backward_points": (C.c_int, [_fp, _i64, _fp, _i64, _i32, _fp, _fp]),
    "mofa_composite_backward": (C.c_int, [_fp, _fp, _i64, _fp, _fp, _i64, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                          _fp]),
    "mofa_panel_floats": (_sz, [_i64, _i32]),
    "mofa_pack_panels": (C.c_int, [_fp, _i32, _i32, _i32, _i32, _fp, _i32, _i32, _i32, _fp]),
    "mofa_to_panels": (C.c_int, [_fp, _i64, _i32, _fp, _i64, _fp]),
    "mofa_from_panels": (C.c_int, [_fp, _i64, _i64, _i32, _fp, _fp]),
    "mofa_layer_forward": (C.c_int, [_fp, _i32, _fp, _i32, _fp, _fp, _i32, _i64, _fp, _i64, _i32, _i32, _fp]),
    "mofa_layer_forward_masked": (C.c_int, [_fp, _i32, _fp, _i32, _fp, _fp, _i32, _i64, _fp, _i64, _i32, _i32, _fp, _fp]),
    "mofa_layer0_forward": (C.c_int, [_fp, _fp, _fp, _i64, _fp, _i64, _i32, _i32, _fp, _fp, _fp, _i64, _i32, _fp, _fp]),
    "mofa_layer0_forward_cam": (C.c_int, [_i32, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp, _i64, _fp, _i64, _i64, _i32,
                                          _i32, _fp, _fp, _fp, _i64, _i32, _fp]),
    "mofa_head_forward": (C.c_int, [_fp, _i32, _i64, _fp, _fp, _i32, _fp, _i32, _i64, _fp]),
    "mofa_view_bias": (C.c_int, [_fp, _i64, _i32, _fp, _i32, _i32, _fp, _fp, _i32, _fp]),
    "mofa_positional_encode": (C.c_int, [_fp, _i64, _i32, _fp, _fp]),
    "mofa_prof_begin": (C.c_int, []),
    "mofa_prof_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "mofa_get_rays": (C.c_int, [_i32, _i32, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _i64, _i64, _fp, _fp, _fp,
                                _fp]),
    "mofa_get_rays_at": (C.c_int, [_i32, _i32, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp, _i64, _fp, _fp, _fp, _fp]),
    "mofa_rays_pose_backward": (C.c_int, [_i32, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _i64, _i64, _fp, _fp, _fp, _fp]),
    "mofa_composite_forward": (C.c_int, [_fp, _fp, _i64, _fp, _fp, _i64, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp]),
    "mofa_sample_pdf_merge": (C.c_int, [_fp, _i64, _fp, _fp, _i64, _i64, _i32, _i32, _fp, _fp, _fp, _fp]),
    "mofa_sample_pdf": (C.c_int, [_fp, _i64, _fp, _fp, _i64, _i64, _i32, _i32, _fp, _fp]),
}

_lib: Optional[C.CDLL] = None


class MofaError(RuntimeError):
    pass


def _exact_fp32_only() -> None:
    """The library computes in exact fp32 (fp32 MFMA, bitwise an fmaf chain) and nothing else.  Earlier rounds carried an opt-in
    split-product experiment behind MOFA_GEMM (bf16x3 / bf16x6 / fp16x3); it is gone, and a leftover setting must not be
    silently ignored — a caller who asked for another arithmetic is told there is none."""
    mode = os.environ.get("MOFA_GEMM", "")
    if mode not in ("", "fp32"):
        raise MofaError(f"MOFA_GEMM={mode!r}: this library has no reduced-precision / split-product mode (exact fp32 only); "
                        "unset MOFA_GEMM")


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built — never falls back."""
    global _lib
    _exact_fp32_only()
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MofaError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950).  There is no CPU or eager fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        if lib.mofa_abi_version() != ABI_VERSION:
            raise MofaError(f"ABI version mismatch: library {lib.mofa_abi_version()} != binding {ABI_VERSION}")
        _lib = lib
    return _lib


PROF_KINDS = 12    # MOFA_PROF_KINDS: k_layer fwd, k_mlp_fused, k_layer<BWD>, k_wgrad, k_layer fwd with per-ray bias (view layer), k_net_chain<0> fwd,
                   # k_net_chain<2> bwd, k_net_chain<1> fwd + mask, and the HBM-bound ray kernels k_composite<1>, k_composite<2>, k_sample_pdf_merge; 11: k_net_chain_train
VERDICT_WORDS = 8  # MOFA_VERDICT_WORDS


class MofaWarning(UserWarning):
    """A condition the library works around (never a wrong result): e.g. a device whose chained-launch self-check failed takes the
    per-layer launches."""


_device_census = {}     # device index -> workgroups seen per XCD by mofa_device_init
_device_selfcheck = {}  # device index -> 1 (the chained launch reproduces the per-layer launches), 0 (it does not: per-layer launches), -1 (not run)


def device_init(device=None):
    """``mofa_device_init`` for ``device`` (default: the current one), once per device and process: the XCD census the chained launch
    relies on, the chained launch's SELF-CHECK (chained vs per-layer launches of a small fixed network, bit for bit, on the device)
    and the persistent kernel's LDS attribute.  This is the library's ONE synchronising call; it belongs where a network is bound to
    a device (``HipNet`` / ``Renderer.bind``), never inside a forward.  A device that fails either check takes the per-layer
    launches (bit-identical, ~1 % slower) and a ``MofaWarning`` says why.  Returns the census (8 counts)."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _device_census:
        counts, ok = (_i32 * 8)(), _i32(-1)
        with torch.cuda.device(idx):
            check(load().mofa_device_init(torch.cuda.current_stream(idx).cuda_stream, counts, C.byref(ok)), "mofa_device_init")
        _device_census[idx], _device_selfcheck[idx] = list(counts), int(ok.value)
        if ok.value != 1:
            import warnings
            warnings.warn(f"cuda:{idx}: {load().mofa_last_error().decode()} (bit-identical results, about 1 % slower)", MofaWarning, stacklevel=2)
    return _device_census[idx]


def chain_selfcheck(device=None) -> int:
    """1 / 0 / -1 as ``mofa_device_init`` reported it for ``device`` (taking the initialisation if it has not run yet)."""
    device_init(device)
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    return _device_selfcheck[torch.cuda.current_device() if idx is None else idx]


def test_hooks(chain_spin_limit: int = 0, chain_skip_xcd: int = -1, selfcheck_poison: bool = False) -> None:
    """``mofa_test_hooks`` (tests / tools only): force the chained launch's failure paths.  Defaults restore the shipped behaviour."""
    check(load().mofa_test_hooks(int(chain_spin_limit), int(chain_skip_xcd), int(bool(selfcheck_poison))), "mofa_test_hooks")


def reload_env() -> None:
    """Re-read the library's MOFA_* run-time knobs from the environment (they are read once at load time)."""
    check(load().mofa_config_reload(), "mofa_config_reload")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise MofaError(f"{what} failed (rc={rc}): {load().mofa_last_error().decode()}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a contiguous fp32 CUDA(HIP) tensor; ``None`` passes NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MofaError("the HIP path needs tensors on the GPU (got a CPU tensor); there is no CPU fallback")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise MofaError(f"expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return t.data_ptr()


def ptr_array(ts: Sequence[torch.Tensor]):
    arr = (_fp * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = ptr(t)
    return arr


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream
