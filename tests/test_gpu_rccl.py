"""The RCCL (backend ``nccl``) branches of the multi-GPU path, executed on ONE GPU.

Two ranks cannot share a device under RCCL, so the 1-GPU box runs a ONE-rank ``nccl`` group with ``MOFA_DIST_FORCE_COLLECTIVES=1``:
every collective an N-rank job issues is issued (not short-circuited) on the same device tensors — see tools/rccl_world1.py.  The
2-GPU form of the same flows is tests/test_gpu_steps.py::test_two_rank_rccl_flows_on_two_gpus."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MOFA_DIST_BACKEND", "MASTER_PORT")}
    env.update(MOFA_DIST_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); env["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
    return env


def _json(cmd, timeout=900):
    out = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_rccl_collectives_of_dist_py_execute_on_one_rank():
    """all_gather_into_tensor into the frame buffer, the flat 32.6 M-float bucket all-reduce, barrier(device_ids), max-over-ranks on
    a device tensor, the helper collectives of train_dp / bulk_render — all on backend nccl with librccl mapped into the process."""
    j = _json([sys.executable, os.path.join(ROOT, "tools", "rccl_world1.py")])
    assert j["ok"] and j["backend"] == "nccl" and j["world"] == 1
    assert any("librccl" in p for p in j["librccl"]), j
    assert j["bucket_floats"] > 32_000_000                 # coarse 1.6 M + fine 27.5 M + texture encoder 3.3 M + StyleModule + codes
    assert j["host_tensor_refused"] is True                 # the backend under test is the one that rejects host buffers


@pytest.mark.parametrize("mode", ["render", "train"])
def test_bench_line_through_the_rccl_branches(mode):
    """bench.py with the N > 1 code path forced at one rank: process group on nccl, the collective inside the timed region, the
    N > 1 JSON fields (backend, collective time) present."""
    extra = ["--size", "64"] if mode == "render" else ["--size", "64", "--rays", "256"]
    j = _json([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", mode, "--steps", "2", "--warmup", "1", "--arch", "8", "64", "10", "64",
               "--cpu-rays", "0"] + extra)
    assert j["n_gpus"] == 1 and j["backend"] == "nccl" and j["rccl_ranks"] == 1 and j["value"] > 0
    assert j["collective"]["avg_ms_per_step_rank0"] > 0
    assert ("all_gather" in j["collective"]["what"]) == (mode == "render")


def test_train_dp_and_bulk_render_through_the_rccl_branches(tmp_path):
    j = _json([sys.executable, os.path.join(ROOT, "tools", "train_dp.py"), "--steps", "2", "--rays", "128", "--size", "32", "--arch", "8", "64", "10", "64"])
    assert j["world"] == 1 and j["parameters_identical_across_ranks"] is True
    j = _json([sys.executable, os.path.join(ROOT, "tools", "bulk_render.py"), "--out", str(tmp_path / "rf"), "--identities", "1", "--expressions", "1",
               "--views", "1", "--size", "32", "--arch", "8", "64", "10", "64"])
    assert j["world"] == 1 and j["images_rendered_total"] == 1
