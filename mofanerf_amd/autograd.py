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
    h.check_codes(e, s, t)
    parts: List[torch.Tensor] = []
    for li, l in enumerate(lin):
        if li == view:
            continue
        b, w = (l.bias.detach(), l.weight.detach()) if detach_params else (l.bias, l.weight)
        if li == 0:
            b = b + w[:, h.ch_pe:h.ch_pe + h.ch_exp] @ e
        elif li in (bim0, bim_skip):
            b = b + w[:, :h.ch_shape] @ s
        elif li in (uv0, uv_skip):
            b = b + w[:, :h.ch_tex] @ t
        parts.append(_pad_to(b, 4 if li > view else _ru(l.out_features, 64)))
    return torch.cat(parts)


def view_bias_torch(h: HipNet, viewdirs: torch.Tensor, detach_params: bool = False) -> torch.Tensor:
    """Differentiable twin of ``mofa_view_bias``: ``[R, roundup(W/2, 64)]``."""
    l = h._linears[-3]
    w, b = (l.weight.detach(), l.bias.detach()) if detach_params else (l.weight, l.bias)
    feats = [viewdirs]
    for i in range(h.view_freqs):                       # multires_views (4 at the shipped configuration)
        f = float(2 ** i)
        feats += [torch.sin(viewdirs * f), torch.cos(viewdirs * f)]
    pe = torch.cat(feats, -1)
    return _pad_to(pe @ w[:, :h.ch_views].t() + b, _ru(l.out_features, 64))


class NetFn(torch.autograd.Function):
    """raw[R,S,4] = NeRF(PE(o + d z), folded biases, per-ray view bias) with a HIP backward.

    What the forward keeps for the backward (``mode``):

    * ``"tape"`` — every layer's fp32 output (98 KiB per fine-network point).  Needed when weight gradients are asked for (the
      weight-gradient GEMMs contract the layer INPUTS with the output gradients): training.
    * ``"mask"`` — ONE BIT per layer output, ``output > 0`` — all the backward needs when only codes / pose are optimised
      (fitting, run_fit.py:305-313 never steps the networks): the activations are recycled as in inference, the saved state is
      1/32 of the tape, nothing is recomputed, gradients are bit-identical to the tape's.  The default without weight gradients
      (``h.force_fp32_tape`` — ``Renderer.fit_tape = "fp32"`` — keeps the fp32 tape instead: the A/B arm).
    * ``"recompute"`` (``Renderer.tape_recompute``) — keep only the inputs and re-run the sub-batch's forward in tape mode inside the
      backward: one sub-batch's tape at a time (bounded by netchunk, not by N_rand), for one extra forward pass per step.  The
      re-run must reproduce the forward the loss saw: the weights are checked (``h._key()``); the library's run-time knobs
      (MOFA_PIPE / MOFA_FUSED / MOFA_CHAIN / MOFA_CHAIN_TRAIN) select between bit-identical forms of the one exact-fp32 arithmetic, so a ``reload_env()`` between
      forward and backward cannot change it (there is no other arithmetic mode in the library).

    ``pts`` given (``run_network(inputs, viewdirs, fn)`` under autograd): explicit points instead of ``o + d z``; the backward then
    returns ``d_pts`` (models/render_class.py:69-94 is an ordinary autograd graph in the reference)."""

    @staticmethod
    def _forward(h: HipNet, ro, rd, zc, z_row_stride, pts, R, S, fo, vb, tape, mask):
        L = h._L
        raw = torch.empty(R, S, 4, dtype=torch.float32, device=fo.device)
        ws = h.workspace(R * S, R, fo.device)
        h.check_verdict()
        lib.check(L.mofa_net_forward(h.shape, lib.ptr(h.packed()), lib.ptr(fo), None, None, lib.ptr(ro), lib.ptr(rd),
                                     lib.ptr(zc), z_row_stride, lib.ptr(pts), None, R, S, lib.ptr(ws), lib.ptr(raw), lib.ptr(tape),
                                     mask.data_ptr() if mask is not None else None, lib.ptr(vb), h.verdict_ptr(fo.device), lib.stream()),
                  "mofa_net_forward(tape)" if tape is not None else "mofa_net_forward")
        h.snapshot_verdict()
        return raw

    @staticmethod
    def forward(ctx, h: HipNet, rays_o, rays_d, z, z_row_stride: int, S: int, folded, vbias, pts, *weights):
        """``weights``: empty (fitting: no weight gradients), or the 2D+7 weight tensors in state-dict order — then the
        backward also returns dW for their per-point column blocks (the constant columns get theirs through the folded
        biases / view-bias rows, i.e. through ordinary torch autograd).  ``pts``: None, or explicit points [R*S,3] (then
        rays_o / rays_d / z are None)."""
        L = h._L
        R = vbias.shape[0]
        dev = vbias.device
        det = lambda t: None if t is None else t.detach().float().contiguous()
        ro, rd, zc, pc = det(rays_o), det(rays_d), det(z), det(pts)
        fo, vb = folded.detach().contiguous(), vbias.detach().contiguous()
        mode = ("recompute" if bool(getattr(h, "tape_recompute", False))
                else ("tape" if weights or bool(getattr(h, "force_fp32_tape", False)) else "mask"))
        tape = mask = None
        if mode == "tape":
            tape = torch.empty(L.mofa_net_tape_floats(h.shape, R * S), dtype=torch.float32, device=dev)
        elif mode == "mask":
            mask = torch.empty(L.mofa_net_mask_tape_words(h.shape, R * S), dtype=torch.int64, device=dev)
        raw = NetFn._forward(h, ro, rd, zc, z_row_stride, pc, R, S, fo, vb, tape, mask)
        ctx.h, ctx.S, ctx.z_row_stride, ctx.R = h, S, z_row_stride, R
        ctx.has_pts = pc is not None
        geo = [pc] if pc is not None else [ro, rd, zc]
        if mode == "recompute":
            ctx.save_for_backward(*geo, fo, vb)
            ctx.key = h._key()                      # the weights the forward ran on: the recomputation must see the same ones
        else:
            ctx.save_for_backward(*geo, tape if tape is not None else mask)
        ctx.mode = mode
        ctx.n_folded, ctx.vb_shape = fo.numel(), tuple(vb.shape)
        ctx.w_shapes = [tuple(w.shape) for w in weights]
        return raw

    @staticmethod
    def backward(ctx, d_raw):
        h, S, R = ctx.h, ctx.S, ctx.R
        L = h._L
        saved = list(ctx.saved_tensors)
        pc = ro = rd = zc = None
        if ctx.has_pts:
            pc = saved.pop(0)
        else:
            ro, rd, zc = saved[:3]
            del saved[:3]
        dev = d_raw.device
        tape = mask = None
        if ctx.mode == "recompute":
            fo, vb = saved
            if h._key() != ctx.key:
                raise lib.MofaError("tape_recompute: a network weight changed between forward and backward (the forward pass cannot be "
                                    "reproduced); step the optimizer after backward, or switch tape_recompute off")
            # re-run in the cheapest mode that serves this backward: the mask-only tape unless weight gradients are wanted
            if ctx.w_shapes:
                tape = torch.empty(L.mofa_net_tape_floats(h.shape, R * S), dtype=torch.float32, device=dev)
            else:
                mask = torch.empty(L.mofa_net_mask_tape_words(h.shape, R * S), dtype=torch.int64, device=dev)
            NetFn._forward(h, ro, rd, zc, ctx.z_row_stride, pc, R, S, fo, vb, tape, mask)
        elif ctx.mode == "tape":
            tape, = saved
        else:
            mask, = saved
        d_raw = d_raw.contiguous()
        d_folded = torch.empty(ctx.n_folded, dtype=torch.float32, device=dev)
        d_vb = torch.empty(ctx.vb_shape, dtype=torch.float32, device=dev)
        d_o = d_d = d_p = None
        if pc is not None:
            d_p = torch.empty_like(pc)
        else:
            d_o, d_d = torch.empty_like(ro), torch.empty_like(rd)
        ws = h.backward_workspace(R * S, dev, with_weight_grads=bool(ctx.w_shapes))
        dws = [torch.zeros(sh, dtype=torch.float32, device=dev) for sh in ctx.w_shapes]
        lib.check(L.mofa_net_backward(h.shape, lib.ptr(h.packed()), lib.ptr(h.packed_t()), lib.ptr(tape),
                                      mask.data_ptr() if mask is not None else None, lib.ptr(d_raw),
                                      lib.ptr(ro), lib.ptr(rd), lib.ptr(zc), ctx.z_row_stride, lib.ptr(pc), R, S, lib.ptr(ws),
                                      lib.ptr(d_folded), lib.ptr(d_vb), lib.ptr(d_o), lib.ptr(d_d), lib.ptr(d_p),
                                      lib.ptr_array(dws) if dws else None, h.verdict_ptr(dev), lib.stream()), "mofa_net_backward")
        h.snapshot_verdict()
        del tape, mask
        return (None, d_o, d_d, None, None, None, d_folded, d_vb, d_p, *dws)


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
