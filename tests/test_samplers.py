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
