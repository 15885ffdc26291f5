import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(autouse=True)
def _seeded():
    """Every test starts from the same RNG state: several tests build randomly initialised networks / UV maps and assert
    properties of the result (a non-zero gradient, a changed parameter), which an unlucky draw - e.g. a network whose density
    is negative on all 64 rays of a tiny frame - would fail for reasons that have nothing to do with the code under test."""
    import torch
    torch.manual_seed(20260927)
    np.random.seed(20260927)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(20260927)
    yield


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
        return cache[name]

    return load


def nan_equal_close(a, b, atol, rtol=0.0):
    """NaN-aware closeness (disp is NaN where acc == 0, SURVEY.md §7 hard part 4)."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    na, nb = np.isnan(a), np.isnan(b)
    assert (na == nb).all(), f"NaN pattern differs: {na.sum()} vs {nb.sum()}"
    d = np.abs(np.where(na, 0, a) - np.where(nb, 0, b))
    lim = atol + rtol * np.abs(np.where(nb, 0, b))
    assert (d <= lim).all(), f"max abs err {d.max():.3e} (limit {atol:.1e}+{rtol:.1e}*|ref|)"
    return float(d.max())


@pytest.fixture
def knob(monkeypatch):
    """Set one of the library's run-time knobs (MOFA_PIPE / MOFA_FUSED / MOFA_CHAIN) for one test.  The library reads its knobs ONCE at load time (no getenv on the launch
    paths), so after changing the environment the snapshot is re-read explicitly — and once more when the test ends."""
    from mofanerf_amd import lib

    def set_(name, value):
        monkeypatch.setenv(name, value)
        lib.reload_env()

    yield set_
    monkeypatch.undo()
    lib.reload_env()
