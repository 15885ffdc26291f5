"""Seeded synthetic weights / codes for benchmarks and parity fixtures.

There is no pretrained checkpoint in the build or on the GPU box (``download_pretrained_models.sh:9``
needs network), so both the golden-vector generator and ``bench.py`` regenerate weights from this
recipe instead of shipping 116 MB of tensors: every tensor is drawn from its own
``numpy.random.Generator`` keyed by ``crc32(key) ^ seed`` with the reference's initialisation law —
Xavier-uniform with ReLU gain for weights (``models/model.py:139-142,190-193,232-244``), PyTorch's
default ``U(-1/sqrt(fan_in), 1/sqrt(fan_in))`` for biases.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Tuple

import numpy as np
import torch

from . import schema


def _rng(key: str, seed: int) -> np.random.Generator:
    return np.random.default_rng((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF)


def make_state(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, tag: str = "") -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    for key, shp in shapes.items():
        rng = _rng(tag + key, seed)
        if key.endswith(".weight"):
            rf = int(np.prod(shp[2:])) if len(shp) > 2 else 1
            fan_out, fan_in = shp[0] * rf, shp[1] * rf
            a = math.sqrt(2.0) * math.sqrt(6.0 / (fan_in + fan_out))
            fan_store = fan_in
        else:
            w = shapes[key[:-5] + ".weight"]
            fan_store = w[1] * (int(np.prod(w[2:])) if len(w) > 2 else 1)
            a = 1.0 / math.sqrt(fan_store)
        out[key] = torch.from_numpy(rng.uniform(-a, a, size=shp).astype(np.float32))
    return out


def nerf_state(D: int, W: int, seed: int = 0, tag: str = "nerf", **widths) -> Dict[str, torch.Tensor]:
    """``widths``: ``ch_pts / ch_shape / ch_tex / ch_views`` of :func:`schema.nerf_layers` for networks built from non-shipped
    flags (multires, multires_views, input_ch_*Codes); the shipped widths keep their round-1 key tags, so existing fixtures
    regenerate byte for byte."""
    suffix = "".join(f"{k}={v}/" for k, v in sorted(widths.items()))
    return make_state(schema.linear_shapes(schema.nerf_layers(D, W, **widths)), seed, f"{tag}/{D}x{W}/{suffix}")


def style_state(seed: int = 0) -> Dict[str, torch.Tensor]:
    return make_state(schema.linear_shapes(schema.style_layers()), seed, "style/")


def tex_encoder_state(seed: int = 0) -> Dict[str, torch.Tensor]:
    return make_state(schema.tex_encoder_shapes(), seed, "texenc/")


def exp_sigma(seed: int = 0, n: int = 20, ch: int = schema.CH_EXP):
    """``expCodes_Sigma``: 20 × ``[1,30]`` ~ U(0,1) (models/render_class.py:53-56)."""
    rng = _rng("expsigma", seed)
    return [torch.from_numpy(rng.uniform(0, 1, size=(1, ch)).astype(np.float32)) for _ in range(n)]


def codes(seed: int = 0):
    """Shape / texture / expression codes with the value ranges of ``configs/texShpDistribution.npy``
    (shape mean∈[-0.034,0.001] std∈[0.0014,0.034]; texture mean∈[-0.047,0.48]
    std∈[0.084,0.53]) and U(0,1) expression codes."""
    rng = _rng("codes", seed)
    shape = rng.normal(rng.uniform(-0.034, 0.001, 50), rng.uniform(0.0014, 0.034, 50)).astype(np.float32)
    tex = rng.normal(rng.uniform(-0.047, 0.48, 256), rng.uniform(0.084, 0.53, 256)).astype(np.float32)
    exp = rng.uniform(0, 1, size=(1, 30)).astype(np.float32)
    return torch.from_numpy(shape)[None, :], torch.from_numpy(tex), torch.from_numpy(exp)


def intrinsics(H: int, W: int, focal_at_512: float = 1200.0) -> np.ndarray:
    """``run_fit.py:142-149``: focal 1200 at 512², principal point at the image centre."""
    f = focal_at_512 * H / 512.0
    return np.array([[f, 0.0, 0.5 * W], [0.0, f, 0.5 * H], [0.0, 0.0, 1.0]])
