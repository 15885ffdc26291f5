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


def _ipc_child(q_in, q_out):
    import torch
    t = q_in.get()                        # a CUDA tensor of the parent, opened here through a HIP IPC memory handle
    ok = bool(t.is_cuda and float(t.sum()) == 1024.0)
    t.mul_(3.0)                           # written in the child, read back by the parent through the same memory
    torch.cuda.synchronize()
    q_out.put(ok)


def test_hip_ipc_memory_handles_work_between_processes():
    """RCCL's intra-node transport maps peer buffers through HIP IPC memory handles; on these hosts only the dmabuf mode works
    (HSA_ENABLE_IPC_MODE_LEGACY=0, defaulted by mofanerf_amd.dist before a group is created — without it the 8-GPU run dies with
    "hipIpcGetMemHandle: invalid argument").  One GPU is enough to exercise that mechanism: a device tensor is handed to a second
    process (torch.multiprocessing: hipIpcGetMemHandle / hipIpcOpenMemHandle), modified there and read back here."""
    code = (
        "import os, sys; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "from mofanerf_amd import dist as mdist; mdist._rccl_env()\n"
        "import torch, torch.multiprocessing as mp\n"
        "from test_gpu_rccl import _ipc_child\n"
        "if __name__ == '__main__':\n"
        "    mp.set_start_method('spawn')\n"
        "    qi, qo = mp.Queue(), mp.Queue()\n"
        "    p = mp.Process(target=_ipc_child, args=(qi, qo)); p.start()\n"
        "    t = torch.ones(1024, device='cuda')\n"
        "    qi.put(t)\n"
        "    ok = qo.get(timeout=120); p.join(60)\n"
        "    torch.cuda.synchronize()\n"
        "    print('IPC', ok, float(t.sum()), os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))\n" % (ROOT, ROOT))
    out = subprocess.run([sys.executable, "-c", code], env={k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"},
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-2500:])
    line = [l for l in out.stdout.splitlines() if l.startswith("IPC")][-1].split()
    assert line[1] == "True" and float(line[2]) == 3072.0 and line[3] == "0", line
