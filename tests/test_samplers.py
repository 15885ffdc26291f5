"""the landmark-biased pixel samplers of the training / fitting scripts on the device, pinned against the
scripts' own ``LMModule.sample_point`` (run_train.py:119-148, run_fit.py:35-82).  ``tests/golden/kat_samplers.npz`` holds what the
reference returned under seeded ``np.random`` together with the draws it consumed; fed the same draws, the device samplers must
return the same pixel lists BIT FOR BIT (integer work).  The same assertions run on CPU tensors (``-m "not gpu"``) and on the GPU."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from mofanerf_amd import rays

G = np.load(os.path.join(GOLDEN, "kat_samplers.npz"))
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def _as_gathered(sel, H, W):
    """What the scripts' ``rays_o[sel[:, 0], sel[:, 1]]`` actually reads: negative indices wrap (the projection's columns are -x)."""
    sel = torch.as_tensor(sel).long().clone()
    assert sel[:, 0].min() >= -H and sel[:, 0].max() < H and sel[:, 1].min() >= -W and sel[:, 1].max() < W   # the reference would raise
    sel[:, 0] += (sel[:, 0] < 0) * H
    sel[:, 1] += (sel[:, 1] < 0) * W
    return sel


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("tag", ["a", "b"])
def test_training_sampler_matches_the_reference_bit_for_bit(tag, dev):
    H = int(G["train_H"])
    K, pose = G["train_K"], torch.from_numpy(G[f"train_{tag}_pose"]).to(dev)
    lm3d = torch.from_numpy(G["train_table"][int(G[f"train_{tag}_id"]), int(G[f"train_{tag}_exp"])] / 50.).to(dev)
    lm2d = rays.project_landmarks(K, pose, lm3d)
    assert lm2d.device.type == dev and lm2d.dtype == torch.long
    assert torch.equal(lm2d.cpu(), torch.from_numpy(G[f"train_{tag}_lm2d"]).long())            # projection: K @ Rt @ X, floor-divide, rotate
    assert int(lm2d[:, 1].max()) < 0                                                               # columns come out as -x (mirrored): the gather wraps them
    n = int(G[f"train_{tag}_n"])
    px = rays.train_pixels(lm2d, n, H, H, precrop_frac=float(G[f"train_{tag}_precrop"]),
                           draws={"rand": G[f"train_{tag}_rand"], "choice": G[f"train_{tag}_choice"]})
    assert px.device.type == dev and px.shape == (n, 2) and px.dtype == torch.long
    assert torch.equal(px.cpu(), _as_gathered(G[f"train_{tag}_select"], H, H))
    # without explicit draws: same structure from the device RNG (uniform part distinct, one shared offset table)
    g = torch.Generator(device=dev).manual_seed(3)
    q = rays.train_pixels(lm2d, n, H, H, generator=g)
    p = int(n / 5 * 3 // 68)
    uni = q[: n - 68 * p].cpu()
    assert len({(int(a), int(b)) for a, b in uni}) == uni.shape[0] and q.min() >= 0 and q.max() < H


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("tag", ["a", "b"])
def test_fitting_sampler_matches_the_reference_bit_for_bit(tag, dev):
    """tag a: more candidates than N_rand (draws with replacement); tag b: an almost empty target, fewer candidates than N_rand —
    the reference's per-row ``ndarray.repeat(tm, 0)[:n]`` (the first ceil(n/tm) candidates, each tm times), which a tiling would not
    reproduce."""
    lm = torch.from_numpy(G["fit_lm"]).to(dev)
    target = torch.from_numpy(G["fit_target" if tag == "a" else "fit_target_small"]).to(dev)
    n = int(G[f"fit_{tag}_n"])
    choice = G[f"fit_{tag}_choice"]
    px = rays.fit_pixels(lm, n, target, scale=2, draws={"rand": G[f"fit_{tag}_rand"], "rand_outline": G[f"fit_{tag}_rand_outline"],
                                                        "choice": choice if choice.size else None})
    ref = torch.from_numpy(G[f"fit_{tag}_select"]).long()
    assert px.device.type == dev and torch.equal(px.cpu(), ref)
    if tag == "b":
        assert choice.size == 0 and G[f"fit_{tag}_rand_outline"].shape[0] > 0
        runs = (ref[1:] != ref[:-1]).any(-1).sum() + 1                  # repeated IN PLACE: far fewer runs than rows
        assert int(runs) <= n // 2
    with pytest.raises(ValueError):                                      # explicit draws of the wrong length are refused, not padded
        rays.fit_pixels(lm, n, target, scale=2, draws={"rand": G[f"fit_{tag}_rand"][:-1], "rand_outline": G[f"fit_{tag}_rand_outline"], "choice": None})


def test_samplers_fail_like_the_reference_instead_of_returning_short_or_piled_up_batches():
    """(i) a uniform part larger than its window raised in the reference (np.random.choice(replace=False), run_train.py:146) —
    never a silently short batch; (ii) ``strict=True`` raises IndexError for a sample outside [-size, size) like the reference's gather,
    instead of clamping it onto a border pixel."""
    H = int(G["train_H"])
    lm2d = torch.from_numpy(G["train_a_lm2d"]).long()
    with pytest.raises(ValueError, match="distinct pixels"):
        rays.train_pixels(lm2d, 4096, H, H, precrop_frac=0.05)                  # 5 % window: 12 x 12 pixels for ~1,600 uniform draws
    with pytest.raises(ValueError, match="draws\\['choice'\\]"):
        rays.train_pixels(lm2d, int(G["train_a_n"]), H, H, draws={"rand": G["train_a_rand"], "choice": G["train_a_choice"][:-5]})
    far = lm2d.clone()
    far[0, 0] = 5 * H                                                           # a landmark projected far outside the frame (bad pose)
    q = rays.train_pixels(far, 1024, H, H, generator=torch.Generator().manual_seed(0))
    assert q.min() >= 0 and q.max() < H                                         # default: clamped to the border, no host sync
    with pytest.raises(IndexError, match="outside"):
        rays.train_pixels(far, 1024, H, H, generator=torch.Generator().manual_seed(0), strict=True)
    lm = torch.from_numpy(G["fit_lm"])
    target = torch.from_numpy(G["fit_target"])
    far = lm.clone()
    far[3] = torch.tensor([4000, 4000])
    with pytest.raises(IndexError, match="outside"):
        rays.fit_pixels(far, 1024, target, scale=2, generator=torch.Generator().manual_seed(0), strict=True)
    assert rays.fit_pixels(far, 1024, target, scale=2, generator=torch.Generator().manual_seed(0)).shape == (1024, 2)


@pytest.mark.gpu
def test_batch_construction_on_the_device_matches_the_scripts_gather():
    """steps.sample_train_batch / sample_fit_batch: landmarks -> pixels -> rays -> target colours without the H x W ray grid.  With the
    reference's draws the pixels are the reference's, and the rays are bit-identical to the full-frame kernel's at those pixels (what
    `rays_o[select[:, 0], select[:, 1]]` gathers in run_train.py:326-329 / run_fit.py:287-292)."""
    from mofanerf_amd import steps
    dev = "cuda"
    H = int(G["train_H"])
    K = G["train_K"]
    pose = torch.from_numpy(G["train_a_pose"]).to(dev)
    lm3d = torch.from_numpy(G["train_table"][int(G["train_a_id"]), int(G["train_a_exp"])] / 50.).to(dev)
    target = torch.rand(H, H, 3, device=dev)
    n = int(G["train_a_n"])
    batch, tgt, pix = steps.sample_train_batch(K, pose, lm3d, target, n, draws={"rand": G["train_a_rand"], "choice": G["train_a_choice"]})
    assert torch.equal(pix.cpu(), _as_gathered(G["train_a_select"], H, H)) and batch.shape == (2, n, 3)
    ro, rd = rays.get_rays(H, H, K, pose, device=dev)
    assert torch.equal(batch[1], rd[pix[:, 0], pix[:, 1]]) and torch.equal(batch[0], ro[pix[:, 0], pix[:, 1]])
    assert torch.equal(tgt, target[pix[:, 0], pix[:, 1]])
    # fitting: half-resolution target, pose with a gradient
    lm = torch.from_numpy(G["fit_lm"]).to(dev)
    tgt_img = torch.from_numpy(G["fit_target"]).to(dev)
    pose_f = pose.clone().requires_grad_(True)
    Kh = K / 2.0
    Kh[2, 2] = 1.0
    batch, tgt, pix = steps.sample_fit_batch(Kh, pose_f, lm, tgt_img, int(G["fit_a_n"]), scale=2,
                                             draws={"rand": G["fit_a_rand"], "rand_outline": G["fit_a_rand_outline"], "choice": G["fit_a_choice"]})
    assert torch.equal(pix.cpu(), torch.from_numpy(G["fit_a_select"]).long())
    batch[1].sum().backward()
    assert pose_f.grad is not None and float(pose_f.grad.abs().sum()) > 0
    assert bool((tgt.sum(-1) != 0).float().mean() > 0.9)                      # only outline extras may sit on empty pixels
