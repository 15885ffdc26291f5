"""Checkpoint schema of the hot path: state-dict key names and tensor shapes.

The names are part of the drop-in boundary (SURVEY.md §8a layer table): a reference ``*.tar``
checkpoint (``run_train.py:371-379``; loaded at ``tools/create_model_condition.py:72-89``) must load
unchanged, so every key below equals what the reference's modules produce
(``models/model.py:97-110,177-187,206-223``; ``models/tex_encoder_mod.py:39-73``).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

PE_POINTS = 63      # 3 + 6*10  (multires = 10, tools/config_parser.py)
PE_VIEWS = 27       # 3 + 6*4   (multires_views = 4)
CH_EXP = 30
CH_SHAPE = 50
CH_TEX = 256
SKIP = 4            # skips=[4], tools/create_model_condition.py:23


def nerf_layers(D: int, W: int, ch_pts: int = PE_POINTS + CH_EXP, ch_shape: int = CH_SHAPE,
                ch_tex: int = CH_TEX, ch_views: int = PE_VIEWS) -> "OrderedDict[str, Tuple[int, int]]":
    """key prefix -> (out_features, in_features) for ``NeRF(D, W, use_viewdirs=True)``."""
    L: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
    L["xyzEncode.linears1.Linear0"] = (W, ch_pts)
    for i in range(1, 4):                                   # skipMLP(D=3, skip=None): Linear1..3
        L[f"xyzEncode.linears1.Linear{i}"] = (W, W)
    for name, cin in (("linear_BiM_xyz", ch_shape), ("linear_uv_xyzBiM", ch_tex)):
        L[f"{name}.linears1.Linear0"] = (W, cin + W)
        for i in range(1, SKIP + 1):
            L[f"{name}.linears1.Linear{i}"] = (W, W)
        L[f"{name}.linears2.Linear0"] = (W, W + cin + W)
        for i in range(1, D - SKIP - 1):
            L[f"{name}.linears2.Linear{i}"] = (W, W)
    L["linear_view_xyBMuv.0"] = (W // 2, ch_views + W)
    L["alpha_linear.0"] = (1, W)
    L["rgb_linear"] = (3, W // 2)
    return L


def style_layers(W: int = 256, ch_shape: int = CH_SHAPE, ch_out: int = CH_EXP) -> "OrderedDict[str, Tuple[int, int]]":
    L: "OrderedDict[str, Tuple[int, int]]" = OrderedDict()
    L["linears1.Linear0"] = (W, ch_shape)
    for i in range(1, 4):
        L[f"linears1.Linear{i}"] = (W, W)
    L["linears_scale"] = (ch_out, W)
    L["linears_bias"] = (ch_out, W)
    return L


def tex_encoder_shapes(code_len: int = CH_TEX) -> "OrderedDict[str, Tuple[int, ...]]":
    """full key -> shape for ``EnDeUVmap`` (conv weights are 4-D)."""
    S: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    chans = [3, 32, 32, 32, 32, 64, 128, 256]
    for i in range(7):
        S[f"encoder.down1.0.{2 * i}.weight"] = (chans[i + 1], chans[i], 4, 4)
        S[f"encoder.down1.0.{2 * i}.bias"] = (chans[i + 1],)
    for name, (o, c) in (("encoder.down2.0", (512, 4096)), ("encoder.mu", (code_len, 512)),
                         ("encoder.logstd", (code_len, 512))):
        S[name + ".weight"], S[name + ".bias"] = (o, c), (o,)
    for i in range(3):
        S[f"encoder.decoding.{2 * i}.weight"], S[f"encoder.decoding.{2 * i}.bias"] = (code_len, code_len), (code_len,)
    return S


def linear_shapes(layers: Dict[str, Tuple[int, int]]) -> "OrderedDict[str, Tuple[int, ...]]":
    S: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for k, (o, c) in layers.items():
        S[k + ".weight"], S[k + ".bias"] = (o, c), (o,)
    return S


def mac_per_point(D: int, W: int, folded: bool = True) -> int:
    """Multiply-accumulates per sample point.  ``folded`` drops the per-call-constant input columns
    (30 expression, 50 shape, 256 texture) and the per-ray-constant 27 view columns."""
    total = 0
    for k, (o, c) in nerf_layers(D, W).items():
        if folded:
            if k == "xyzEncode.linears1.Linear0":
                c -= CH_EXP
            elif k.startswith("linear_BiM_xyz") and k.endswith("Linear0"):
                c -= CH_SHAPE
            elif k.startswith("linear_uv_xyzBiM") and k.endswith("Linear0"):
                c -= CH_TEX
            elif k == "linear_view_xyBMuv.0":
                c -= PE_VIEWS
        total += o * c
    return total
