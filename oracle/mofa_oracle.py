"""CPU oracle for the MoFaNeRF ray-marching hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch fp32 restatement (torch CPU ops, functional style, weights passed as
plain ``{state-dict key: tensor}`` dicts) of the reference algorithm.  It is the *checker* for the
HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it.  Nothing under ``mofanerf_amd/`` imports it, and the product path raises when the
HIP library is missing instead of falling back to anything here.

Pinning: the reference ships no tests or golden vectors of its own, so this oracle is
pinned against outputs of the reference itself, imported on CPU in the build container by
``tests/golden/make_golden.py`` (fixtures ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
replays them, CPU only).

Each function cites the reference file:line (relative to zhuhao-nju/mofanerf) it restates.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


# --------------------------------------------------------------------------------------------
# positional encoding — models/model.py:15-63 (duplicate tools/run_nerf_helpers.py:15-63)
# --------------------------------------------------------------------------------------------
def pe_freqs(n_freqs: int) -> Tensor:
    """``2 ** linspace(0, L-1, L)`` (models/model.py:32): exact powers of two."""
    return 2.0 ** torch.linspace(0.0, float(n_freqs - 1), steps=n_freqs)


def positional_encode(x: Tensor, n_freqs: int) -> Tensor:
    """γ(x) = [x, sin(2^0 x), cos(2^0 x), …]; every block is 3 wide, frequency-major
    (models/model.py:24-45).  ``[N,3] -> [N, 3 + 6 L]``."""
    blocks = [x]
    for f in pe_freqs(n_freqs):
        xf = x * f
        blocks.append(torch.sin(xf))
        blocks.append(torch.cos(xf))
    return torch.cat(blocks, -1)


# --------------------------------------------------------------------------------------------
# the conditioned MLP — models/model.py:80-137 (NeRF), :202-230 (skipMLP), :174-199 (StyleModule)
# --------------------------------------------------------------------------------------------
def _lin(st: State, key: str, x: Tensor) -> Tensor:
    return F.linear(x, st[key + ".weight"], st[key + ".bias"])


def _count(st: State, prefix: str) -> int:
    n = 0
    while f"{prefix}.Linear{n}.weight" in st:
        n += 1
    return n


def skip_mlp(st: State, prefix: str, x: Tensor) -> Tensor:
    """models/model.py:226-230.  ReLU after EVERY Linear, including the last; when the second
    stack exists its input is ``[x ‖ h]``."""
    h = x
    for i in range(_count(st, prefix + ".linears1")):
        h = torch.relu(_lin(st, f"{prefix}.linears1.Linear{i}", h))
    n2 = _count(st, prefix + ".linears2")
    if n2:
        h = torch.cat([x, h], 1)
        for i in range(n2):
            h = torch.relu(_lin(st, f"{prefix}.linears2.Linear{i}", h))
    return h


def nerf_forward(st: State, pts: Tensor, shape: Tensor, views: Tensor, tex: Tensor) -> Tensor:
    """models/model.py:121-137.  ``pts [n,93] shape [n,50] views [n,27] tex [n,256] -> [n,4]``
    (rgb pre-sigmoid, sigma pre-ReLU)."""
    xyz = skip_mlp(st, "xyzEncode", pts)
    sig = skip_mlp(st, "linear_BiM_xyz", torch.cat([shape, xyz], 1))
    alpha = _lin(st, "alpha_linear.0", sig)
    col = skip_mlp(st, "linear_uv_xyzBiM", torch.cat([tex, sig], 1))
    v = torch.relu(_lin(st, "linear_view_xyBMuv.0", torch.cat([views, col], 1)))
    rgb = _lin(st, "rgb_linear", v)
    return torch.cat([rgb, alpha], -1)


def style_module(st: State, shape_row: Tensor) -> Tuple[Tensor, Tensor]:
    """models/model.py:195-199: 4×(Linear+ReLU) then two heads.  ``[1,50] -> ([1,30],[1,30])``."""
    h = shape_row
    for i in range(_count(st, "linears1")):
        h = torch.relu(_lin(st, f"linears1.Linear{i}", h))
    return _lin(st, "linears_scale", h), _lin(st, "linears_bias", h)


def tex_encoder(st: State, uv_map_hw3: Tensor) -> Tensor:
    """models/tex_encoder_mod.py:79-100 as called from models/render_class.py:184-185.
    ``[512,512,3] -> [1,256]``.  The ZeroPad2d is a no-op at 512²; ``logstd`` is unused."""
    x = uv_map_hw3.permute(2, 0, 1).unsqueeze(0)
    for i in range(7):
        x = F.leaky_relu(F.conv2d(x, st[f"encoder.down1.0.{2 * i}.weight"], st[f"encoder.down1.0.{2 * i}.bias"],
                                  stride=2, padding=1), 0.2)
    x = x.reshape(-1, 256 * 4 * 4)
    x = F.leaky_relu(_lin(st, "encoder.down2.0", x), 0.2)
    z = _lin(st, "encoder.mu", x)
    for i in range(3):
        z = F.leaky_relu(_lin(st, f"encoder.decoding.{2 * i}", z), 0.1)
    return z


# --------------------------------------------------------------------------------------------
# rays — tools/run_nerf_helpers.py:153-168, tools/load_facescape.py:9-38
# --------------------------------------------------------------------------------------------
def pose_spherical(phi_deg: float, theta_deg: float, radius: float) -> Tensor:
    """tools/load_facescape.py:33-38 (float32 matrices multiplied in float32)."""
    def f32(rows):
        return np.array(rows).astype(np.float32)
    ph, th = phi_deg / 180.0 * np.pi, theta_deg / 180.0 * np.pi
    t = f32([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]])
    rx = f32([[1, 0, 0, 0], [0, np.cos(th), -np.sin(th), 0], [0, np.sin(th), np.cos(th), 0], [0, 0, 0, 1]])
    ry = f32([[np.cos(ph), 0, -np.sin(ph), 0], [0, 1, 0, 0], [np.sin(ph), 0, np.cos(ph), 0], [0, 0, 0, 1]])
    return torch.from_numpy(ry @ (rx @ t))


def get_rays(H: int, W: int, K, c2w: Tensor) -> Tuple[Tensor, Tensor]:
    """tools/run_nerf_helpers.py:153-168: pinhole rays, pixel (row j, col i)."""
    ii = torch.linspace(0, W - 1, W)[None, :].expand(H, W)
    jj = torch.linspace(0, H - 1, H)[:, None].expand(H, W)
    dirs = torch.stack([(ii - float(K[0][2])) / float(K[0][0]), -(jj - float(K[1][2])) / float(K[1][1]),
                        -torch.ones(H, W)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, -1].expand(rays_d.shape)
    return rays_o, rays_d


def ndc_rays(H: int, W: int, focal: float, near: float, rays_o: Tensor, rays_d: Tensor) -> Tuple[Tensor, Tensor]:
    """tools/run_nerf_helpers.py:182-200: shift the origins to the near plane, then project to normalised device
    coordinates (forward-facing scenes; ``render(..., ndc=True)``, models/render_class.py:166-169)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1. / (W / (2. * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1. / (H / (2. * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1. + 2. * near / rays_o[..., 2]
    d0 = -1. / (W / (2. * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1. / (H / (2. * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2. * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# --------------------------------------------------------------------------------------------
# compositing — models/render_class.py:440-482
# --------------------------------------------------------------------------------------------
def raw2outputs(raw: Tensor, z_vals: Tensor, rays_d: Tensor, noise: Optional[Tensor] = None,
                white_bkgd: bool = False):
    """``raw [R,S,4], z [R,S], rays_d [R,3] -> rgb [R,3], disp [R], acc [R], weights [R,S], depth [R]``.
    ``noise`` is the already-scaled additive sigma noise (render_class.py:461-470) or None."""
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    sig = raw[..., 3] if noise is None else raw[..., 3] + noise
    alpha = 1.0 - torch.exp(-torch.relu(sig) * dists)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * trans
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth = torch.sum(weights * z_vals, -1)
    acc = torch.sum(weights, -1)
    disp = 1.0 / torch.max(1e-10 * torch.ones_like(depth), depth / acc)   # NaN where acc == 0
    if white_bkgd:
        rgb_map = rgb_map + (1.0 - acc[..., None])
    return rgb_map, disp, acc, weights, depth


# --------------------------------------------------------------------------------------------
# importance resampling — tools/run_nerf_helpers.py:203-247
# --------------------------------------------------------------------------------------------
def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor) -> Tensor:
    """Inverse-CDF sampling.  ``bins [R,B], weights [R,B-1], u [R,n] or [n] -> [R,n]``.
    ``u`` is explicit: ``linspace(0,1,n)`` for ``det`` (:212), uniform randoms otherwise (:215)."""
    w = weights + 1e-5
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = u.expand(list(cdf.shape[:-1]) + [u.shape[-1]]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - c0) / denom
    return b0 + t * (b1 - b0)


# --------------------------------------------------------------------------------------------
# the renderer — models/render_class.py:69-123, 239-352, 125-197, 354-437
# --------------------------------------------------------------------------------------------
class OracleRenderer:
    """Functional mirror of ``myRenderer``'s per-call state (render_class.py:40-58, 417-424).

    ``coarse``/``fine``: NeRF state dicts; ``style``: StyleModule state dict; ``tex_enc``: texture
    encoder state dict (only for :meth:`render`); ``exp_sigma``: list of ``[1,30]`` tensors."""

    def __init__(self, coarse: State, fine: Optional[State], style: State, exp_sigma: Sequence[Tensor],
                 tex_enc: Optional[State] = None, netchunk: int = 1024 * 64, multires: int = 10,
                 multires_views: int = 4):
        self.coarse, self.fine, self.style, self.tex_enc = coarse, fine, style, tex_enc
        self.exp_sigma = list(exp_sigma)
        self.netchunk, self.L, self.Lv = netchunk, multires, multires_views

    # render_class.py:69-109 ------------------------------------------------------------------
    def run_network(self, pts: Tensor, viewdirs: Tensor, st: State, shape_codes: Tensor, tex_code: Tensor,
                    exp_type: int) -> Tensor:
        flat = pts.reshape(-1, 3)
        n = flat.shape[0]
        scale, bias = style_module(self.style, shape_codes[0:1, :])
        e = scale * self.exp_sigma[exp_type] + bias
        x93 = torch.cat([positional_encode(flat, self.L), e.expand(n, -1)], -1)
        bm = shape_codes[0, :].expand(n, -1)
        dirs = viewdirs[:, None].expand(pts.shape).reshape(-1, 3)
        v27 = positional_encode(dirs, self.Lv)
        tex = tex_code.reshape(1, -1).expand(n, -1)
        outs = [nerf_forward(st, x93[i:i + self.netchunk], bm[i:i + self.netchunk], v27[i:i + self.netchunk],
                             tex[i:i + self.netchunk]) for i in range(0, n, self.netchunk)]
        return torch.cat(outs, 0).reshape(list(pts.shape[:-1]) + [4])

    # render_class.py:239-352 -----------------------------------------------------------------
    def render_rays(self, rays: Tensor, shape_codes: Tensor, tex_code: Tensor, exp_type: int, N_samples: int,
                    N_importance: int, perturb: float = 0.0, white_bkgd: bool = False, lindisp: bool = False,
                    t_rand: Optional[Tensor] = None, u: Optional[Tensor] = None, noise0: Optional[Tensor] = None,
                    noise1: Optional[Tensor] = None, retraw: bool = False, keep: bool = False,
                    w0_perturb: Optional[Tensor] = None):
        """``rays [R,11] = o3 d3 near far viewdir3``.  Stochastic inputs are explicit:
        ``t_rand [R,S]`` (stratified jitter, :305), ``u [R,Ni]`` (:215), ``noise*`` (:463).
        ``w0_perturb [R,S]`` (checker-only, not in the reference): multiplies the coarse weights before the resampling —
        ``1 + U(-1e-6, 1e-6)`` mimics the rounding differences of a second correct fp32 implementation, so the outputs under
        it measure how far THIS algorithm's pixels move by themselves (tests/harness.py::envelope_stats)."""
        R = rays.shape[0]
        o, d, vd = rays[:, 0:3], rays[:, 3:6], rays[:, 8:11]
        near, far = rays[:, 6:7], rays[:, 7:8]
        t = torch.linspace(0.0, 1.0, steps=N_samples)
        z = near * (1.0 - t) + far * t if not lindisp else 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
        z = z.expand(R, N_samples)
        if perturb > 0.0:
            mids = 0.5 * (z[..., 1:] + z[..., :-1])
            upper, lower = torch.cat([mids, z[..., -1:]], -1), torch.cat([z[..., :1], mids], -1)
            z = lower + (upper - lower) * t_rand
        pts = o[..., None, :] + d[..., None, :] * z[..., :, None]
        raw0 = self.run_network(pts, vd, self.coarse, shape_codes, tex_code, exp_type)
        rgb0, disp0, acc0, w0, _ = raw2outputs(raw0, z, d, noise0, white_bkgd)
        out = {"rgb_map": rgb0, "disp_map": disp0, "acc_map": acc0}
        dbg = {"z_coarse": z, "raw_coarse": raw0, "weights_coarse": w0}
        if N_importance > 0:
            zmid = 0.5 * (z[..., 1:] + z[..., :-1])
            if u is None:
                u = torch.linspace(0.0, 1.0, steps=N_importance)
            zs = sample_pdf(zmid, (w0 if w0_perturb is None else w0 * w0_perturb)[..., 1:-1], u)
            zs = zs.detach()                                   # render_class.py:326: no gradient through the resampling
            zf, _ = torch.sort(torch.cat([z, zs], -1), -1)
            pts = o[..., None, :] + d[..., None, :] * zf[..., :, None]
            raw1 = self.run_network(pts, vd, self.fine if self.fine is not None else self.coarse, shape_codes,
                                    tex_code, exp_type)
            rgb1, disp1, acc1, w1, _ = raw2outputs(raw1, zf, d, noise1, white_bkgd)
            out = {"rgb_map": rgb1, "disp_map": disp1, "acc_map": acc1, "rgb0": rgb0, "disp0": disp0,
                   "acc0": acc0, "z_std": torch.std(zs, dim=-1, unbiased=False)}
            if retraw:
                out["raw"] = raw1
            dbg.update({"z_samples": zs, "z_fine": zf, "raw_fine": raw1, "weights_fine": w1})
        elif retraw:
            out["raw"] = raw0
        if keep:
            out["_dbg"] = dbg
        return out

    # render_class.py:111-123 + :354-437 (render_fitting) / :125-197 (render) --------------------
    def render(self, rays_o: Tensor, rays_d: Tensor, chunk: int, shape_codes: Tensor, exp_type: int,
               near: float, far: float, tex_code: Optional[Tensor] = None, uv_map: Optional[Tensor] = None,
               exp_codes: Optional[Tensor] = None, **kw):
        """``render_fitting`` when ``tex_code`` is given (and ``exp_codes`` stored at slot 20,
        :420-423); ``render`` when ``uv_map`` is given (texture encoder evaluated first, :184)."""
        if exp_codes is not None:
            if len(self.exp_sigma) == 20:
                self.exp_sigma.append(exp_codes)
            else:
                self.exp_sigma[20] = exp_codes
        if tex_code is None:
            tex_code = tex_encoder(self.tex_enc, uv_map)
        sh = rays_d.shape
        vd = (rays_d / torch.norm(rays_d, dim=-1, keepdim=True)).reshape(-1, 3).float()
        o, d = rays_o.reshape(-1, 3).float(), rays_d.reshape(-1, 3).float()
        rays = torch.cat([o, d, near * torch.ones_like(d[:, :1]), far * torch.ones_like(d[:, :1]), vd], -1)
        parts: Dict[str, list] = {}
        dbg: Dict[str, list] = {}
        per_ray = ("t_rand", "u", "noise0", "noise1", "w0_perturb")
        for i in range(0, rays.shape[0], chunk):
            kwi = {k: (v[i:i + chunk] if k in per_ray and v is not None and v.dim() > 1 else v) for k, v in kw.items()}
            r = self.render_rays(rays[i:i + chunk], shape_codes, tex_code, exp_type, **kwi)
            for k, v in (r.pop("_dbg", None) or {}).items():
                dbg.setdefault(k, []).append(v)
            for k, v in r.items():
                parts.setdefault(k, []).append(v)
        allr = {k: torch.cat(v, 0) for k, v in parts.items()}
        allr = {k: v.reshape(list(sh[:-1]) + list(v.shape[1:])) for k, v in allr.items()}
        extras = {k: v for k, v in allr.items() if k not in ("rgb_map", "disp_map", "acc_map")}
        extras["losses"] = 0    # lossesLog.out() is always the int 0 (render_class.py:30-37, encoder returns {})
        if dbg:                 # keep=True: per-ray intermediates, flat over rays
            extras["_dbg"] = {k: torch.cat(v, 0) for k, v in dbg.items()}
        return [allr["rgb_map"], allr["disp_map"], allr["acc_map"], extras]
