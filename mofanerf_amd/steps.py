"""The callers' inner loops, reduced to the calls that cross the boundary (SURVEY.md §8b "what the build's own harness
must reproduce"): a photometric fitting step (run_fit.py:268-313), a training step (run_train.py:278-364) and the bulk
renderer's identity loop (render_refine_trainSet.py:245-295).  They exist so that the boundary is exercised with the
reference's argument shapes by tests and timing tools; they carry no I/O, CLI or dataset code (out of scope)."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import dist as mdist


def sample_train_batch(K, c2w: torch.Tensor, lm3d: torch.Tensor, target: torch.Tensor, n_rand: int, precrop_frac: float = 0.0,
                       generator=None, draws=None):
    """The batch construction of run_train.py:306-330 without leaving the device and without the H x W ray grid: 3D landmarks ->
    pixel table (`rays.project_landmarks`), landmark-biased + uniform pixel choice (`rays.train_pixels`), rays through exactly those
    pixels (`rays.rays_at_pixels`, HIP) and the target colours at them.  Returns ``(batch_rays [2,N,3], target_s [N,3], pixels [N,2])``."""
    from . import rays as mrays
    H, W = int(target.shape[0]), int(target.shape[1])
    lm2d = mrays.project_landmarks(K, c2w, lm3d)
    pix = mrays.train_pixels(lm2d, n_rand, H, W, precrop_frac=precrop_frac, generator=generator, draws=draws)
    batch = mrays.rays_at_pixels(K, c2w, pix[:, 0], pix[:, 1], H, W)
    return batch, target[pix[:, 0], pix[:, 1]], pix


def sample_fit_batch(K, c2w: torch.Tensor, landmarks_rc: torch.Tensor, target: torch.Tensor, n_rand: int, scale: int = 1,
                     generator=None, draws=None):
    """The batch construction of run_fit.py:281-293: landmark-biased pixels on non-empty target pixels (`rays.fit_pixels`), rays
    through them — differentiable in the fitted pose ``c2w`` (`mofa_rays_pose_backward`) — and the target colours."""
    from . import rays as mrays
    H, W = int(target.shape[0]), int(target.shape[1])
    pix = mrays.fit_pixels(landmarks_rc, n_rand, target, scale=scale, generator=generator, draws=draws)
    batch = mrays.rays_at_pixels(K, c2w, pix[:, 0], pix[:, 1], H, W)
    return batch, target[pix[:, 0], pix[:, 1]], pix


def fit_step(render, render_kwargs: Dict, optimizers: List[torch.optim.Optimizer], H: int, W: int, K, batch_rays,
             target_rgb, shape_code, tex_code, exp_code, light_scale, chunk: int):
    """One iteration of run_fit.py's loop: 1024 sampled rays -> render_fitting -> L1(light*rgb, target) -> backward ->
    Adam steps on (codes, pose, light).  ``batch_rays [2,N,3]`` may carry gradients to the pose."""
    rgb, disp, acc, extras = render.render_fitting(H, W, K, chunk=chunk, rays=batch_rays,
                                                   shapeCodes=shape_code.expand(batch_rays.shape[1], -1), uvCodes=tex_code,
                                                   expType=20, expCodes=exp_code, **render_kwargs)
    for o in optimizers:
        o.zero_grad()
    loss = torch.nn.functional.l1_loss(light_scale[0] * rgb, target_rgb)          # run_fit.py:309
    loss.backward()
    for o in optimizers:
        o.step()
    return loss.detach(), rgb.detach()


def train_step(render, render_kwargs: Dict, optimizer: torch.optim.Optimizer, bucket: Optional[mdist.GradBucket], H: int,
               W: int, K, batch_rays, target_rgb, shape_codes, uv_map, exp_type: int, chunk: int):
    """One iteration of run_train.py's loop: render() on N_rand rays (texture encoder evaluated on the UV map),
    MSE(rgb) + MSE(rgb0) + encoder losses, backward, gradient all-reduce over the data-parallel ranks, Adam."""
    rgb, disp, acc, extras = render.render(H, W, K, chunk=chunk, rays=batch_rays, shapeCodes=shape_codes, uvMap=uv_map,
                                           expType=exp_type, retraw=True, **render_kwargs)
    if bucket is not None:
        bucket.zero()
    else:
        optimizer.zero_grad()
    loss = torch.mean((rgb - target_rgb) ** 2) + torch.mean((extras["rgb0"] - target_rgb) ** 2) + extras["losses"]
    loss.backward()
    if bucket is not None:
        bucket.sync()
    optimizer.step()
    return loss.detach()


def bulk_render_identities(render, render_kwargs: Dict, identities: List, render_one, rank: int = 0, world: int = 1):
    """render_refine_trainSet.py's outer loop, sharded: rank r renders identities[shard_range(...)] — the reference's own
    begin_person/end_person knob (:158-159) — with no data-path collective.  ``render_one(identity)`` issues the
    render_path calls for one identity."""
    done = []
    for ident in mdist.shard_list(list(identities), rank, world):
        done.append(render_one(ident))
    return done
