"""Autograd glue for the fitting / training entry points (run_fit.py:305-313, run_train.py:333-357).

The per-point work — forward with a tape of every layer's output, backward-data GEMMs, ReLU masks, bias-gradient
column sums, positional-encoding and compositing backward — runs in HIP (``mofa_net_forward`` with a tape,
``mofa_net_backward``, ``mofa_composite_backward``).  What PyTorch's autograd carries is only the per-CALL /
per-RAY algebra whose inputs are the things the scripts optimise:

* the folded bias blob ``b' = b + W[:, const cols] @ code`` (five matvecs per network) — so gradients reach the shape /
  texture / expression codes (and, in training, the constant weight columns and every bias);
* the per-ray view-bias rows ``b + PE(viewdir) @ W[:, :27]^T`` — so gradients reach the camera pose through viewdirs;
* ``viewdirs = rays_d / |rays_d|`` and the ray tensors themselves.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn.functional as F

from . import lib, schema
from .hipnet import HipNet


def _pad_to(v: torch.Tensor, n: int) -> torch.Tensor:
    return v if v.shape[-1] == n else F.pad(v, (0, n - v.shape[-1]))


def _ru(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def fold_torch(h: HipNet, exp_code: torch.Tensor, shape_code: torch.Tensor, tex_code: torch.Tensor,
               detach_params: bool = False) -> torch.Tensor:
    """Differentiable twin of ``mofa_net_fold``: same blob layout (one ``n_padded`` slice per layer in state-dict order,
    the view layer skipped, heads padded to 4).  ``detach_params``: gradients flow to the codes only (fitting without
    weight gradients) — no network parameter receives a ``.grad``."""
    D = h.D
    lin = h._linears
    bim0, bim_skip, uv0, uv_skip, view = 4, 9, 4 + D, 9 + D, 4 + 2 * D
    e, s, t = exp_code.reshape(-1), shape_code.reshape(-1), tex_code.reshape(-1)
    parts: List[torch.Tensor] = []
    for li, l in enumerate(lin):
        if li == view:
            continue
        b, w = (l.bias.detach(), l.weight.detach()) if detach_params else (l.bias, l.weight)
        if li == 0:
            b = b + w[:, schema.PE_POINTS:schema.PE_POINTS + schema.CH_EXP] @ e
        elif li in (bim0, bim_skip):
            b = b + w[:, :schema.CH_SHAPE] @ s
        elif li in (uv0, uv_skip):
            b = b + w[:, :schema.CH_TEX] @ t
        parts.append(_pad_to(b, 4 if li > view else _ru(l.out_features, 64)))
    return torch.cat(parts)


def view_bias_torch(h: HipNet, viewdirs: torch.Tensor, detach_params: bool = False) -> torch.Tensor:
    """Differentiable twin of ``mofa_view_bias``: ``[R, roundup(W/2, 64)]``."""
    l = h._linears[-3]
    w, b = (l.weight.detach(), l.bias.detach()) if detach_params else (l.weight, l.bias)
    feats = [viewdirs]
    for i in range(4):                                  # multires_views = 4
        f = float(2 ** i)
        feats += [torch.sin(viewdirs * f), torch.cos(viewdirs * f)]
    pe = torch.cat(feats, -1)
    return _pad_to(pe @ w[:, :schema.PE_VIEWS].t() + b, _ru(l.out_features, 64))


class NetFn(torch.autograd.Function):
    """raw[R,S,4] = NeRF(PE(o + d z), folded biases, per-ray view bias) with a HIP backward.

    ``h.tape_recompute`` (set through ``Renderer.tape_recompute``): the forward keeps NO tape — it is the inference launch, four recycled
    activation buffers — and the backward first re-runs the forward of its sub-batch in tape mode, then walks it.  Same kernels, same
    values bit for bit (the two forward modes differ only in where layer outputs land); the saved state of a training step drops from
    98 KiB per fine-network point for EVERY sub-batch (52.6 GB at N_rand = 4096) to one sub-batch's tape at a time (bounded by netchunk,
    not by N_rand), for one extra forward pass per step."""

    @staticmethod
    def _forward(h: HipNet, ro, rd, zc, z_row_stride, S, fo, vb, tape):
        L = h._L
        R = ro.shape[0]
        raw = torch.empty(R, S, 4, dtype=torch.float32, device=ro.device)
        ws = h.workspace(R * S, R, ro.device)
        lib.check(L.mofa_net_forward(h.shape, lib.ptr(h.packed()), lib.ptr(fo), None, None, lib.ptr(ro), lib.ptr(rd),
                                     lib.ptr(zc), z_row_stride, None, None, R, S, lib.ptr(ws), lib.ptr(raw), lib.ptr(tape),
                                     lib.ptr(vb), lib.stream()), "mofa_net_forward(tape)" if tape is not None else "mofa_net_forward")
        return raw

    @staticmethod
    def forward(ctx, h: HipNet, rays_o, rays_d, z, z_row_stride: int, S: int, folded, vbias, *weights):
        """``weights``: empty (fitting: no weight gradients), or the 2D+7 weight tensors in state-dict order — then the
        backward also returns dW for their per-point column blocks (the constant columns get theirs through the folded
        biases / view-bias rows, i.e. through ordinary torch autograd)."""
        L = h._L
        R = rays_o.shape[0]
        dev = rays_o.device
        ro, rd = rays_o.detach().contiguous(), rays_d.detach().contiguous()
        zc = z.detach().contiguous()
        fo, vb = folded.detach().contiguous(), vbias.detach().contiguous()
        recompute = bool(getattr(h, "tape_recompute", False))
        tape = None if recompute else torch.empty(L.mofa_net_tape_floats(h.shape, R * S), dtype=torch.float32, device=dev)
        raw = NetFn._forward(h, ro, rd, zc, z_row_stride, S, fo, vb, tape)
        ctx.h, ctx.S, ctx.z_row_stride = h, S, z_row_stride
        if recompute:
            ctx.save_for_backward(ro, rd, zc, fo, vb)
            ctx.key = h._key()                      # the weights the forward ran on: the recomputation must see the same ones
        else:
            ctx.save_for_backward(ro, rd, zc, tape)
        ctx.recompute = recompute
        ctx.n_folded, ctx.vb_shape = fo.numel(), tuple(vb.shape)
        ctx.w_shapes = [tuple(w.shape) for w in weights]
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        h, S = ctx.h, ctx.S
        L = h._L
        if ctx.recompute:
            ro, rd, zc, fo, vb = ctx.saved_tensors
            if h._key() != ctx.key:
                raise lib.MofaError("tape_recompute: a network weight changed between forward and backward (the forward pass cannot be "
                                    "reproduced); step the optimizer after backward, or switch tape_recompute off")
            tape = torch.empty(L.mofa_net_tape_floats(h.shape, ro.shape[0] * S), dtype=torch.float32, device=ro.device)
            NetFn._forward(h, ro, rd, zc, ctx.z_row_stride, S, fo, vb, tape)
        else:
            ro, rd, zc, tape = ctx.saved_tensors
        R, dev = ro.shape[0], ro.device
        d_raw = d_raw.contiguous()
        d_folded = torch.empty(ctx.n_folded, dtype=torch.float32, device=dev)
        d_vb = torch.empty(ctx.vb_shape, dtype=torch.float32, device=dev)
        d_o, d_d = torch.empty_like(ro), torch.empty_like(rd)
        ws = h.backward_workspace(R * S, dev)
        dws = [torch.zeros(sh, dtype=torch.float32, device=dev) for sh in ctx.w_shapes]
        lib.check(L.mofa_net_backward(h.shape, lib.ptr(h.packed()), lib.ptr(h.packed_t()), lib.ptr(tape), lib.ptr(d_raw),
                                      lib.ptr(ro), lib.ptr(rd), lib.ptr(zc), ctx.z_row_stride, R, S, lib.ptr(ws),
                                      lib.ptr(d_folded), lib.ptr(d_vb), lib.ptr(d_o), lib.ptr(d_d),
                                      lib.ptr_array(dws) if dws else None, lib.stream()), "mofa_net_backward")
        del tape
        return (None, d_o, d_d, None, None, None, d_folded, d_vb, *dws)


class CompositeFn(torch.autograd.Function):
    """raw2outputs with a HIP backward.  Returns (rgb, disp, acc, depth, weights)."""

    @staticmethod
    def forward(ctx, raw, z, z_row_stride: int, rays_d, noise, white_bkgd: bool):
        L = lib.load()
        R, S = raw.shape[0], raw.shape[1]
        dev = raw.device
        rawc, zc, rd = raw.detach().contiguous(), z.detach().contiguous(), rays_d.detach().contiguous()
        o = [torch.empty(R, *sh, dtype=torch.float32, device=dev) for sh in ((3,), (), (), (), (S,))]
        lib.check(L.mofa_composite_forward(lib.ptr(rawc), lib.ptr(zc), z_row_stride, lib.ptr(rd), lib.ptr(noise), R, S,
                                           int(bool(white_bkgd)), *[lib.ptr(t) for t in o], lib.stream()),
                  "mofa_composite_forward")
        ctx.save_for_backward(rawc, zc, rd, noise if noise is not None else torch.empty(0, device=dev))
        ctx.meta = (z_row_stride, bool(white_bkgd), noise is not None)
        return tuple(o)

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_depth, g_weights):
        L = lib.load()
        rawc, zc, rd, noise = ctx.saved_tensors
        z_row_stride, white, has_noise = ctx.meta
        R, S = rawc.shape[0], rawc.shape[1]
        d_raw = torch.empty_like(rawc)
        d_rd = torch.empty_like(rd)
        c = lambda g: None if g is None else g.contiguous()
        if g_rgb is None:
            g_rgb = torch.zeros(R, 3, dtype=torch.float32, device=rawc.device)
        lib.check(L.mofa_composite_backward(lib.ptr(rawc), lib.ptr(zc), z_row_stride, lib.ptr(rd),
                                            lib.ptr(noise) if has_noise else None, R, S, int(white), lib.ptr(c(g_rgb)),
                                            lib.ptr(c(g_disp)), lib.ptr(c(g_acc)), lib.ptr(c(g_depth)),
                                            lib.ptr(c(g_weights)), lib.ptr(d_raw), lib.ptr(d_rd), lib.stream()),
                  "mofa_composite_backward")
        return d_raw, None, None, d_rd, None, None
