#!/usr/bin/env python3
"""BASELINE config 5 — run_train.py's loop shape, data parallel: one process per GPU, every rank draws its own N_rand rays of its
own synthetic identity, local forward/backward through the HIP path, ONE all-reduce of the flat gradient bucket (RCCL), identical
Adam step.  Prints per-step time and verifies that all ranks hold bit-identical parameters afterwards.

  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_dp.py --steps 5
  (functional test on one GPU: MOFA_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ... --arch 8 64 10 64 --rays 256)
"""
import argparse, hashlib, json, os, sys, time
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import dist as mdist, factory, rays as mrays, steps, synth


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--rays", type=int, default=4096, help="N_rand per rank")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--arch", type=int, nargs=4, default=[8, 256, 10, 1024])
    a = ap.parse_args(argv)
    rank, world, local = mdist.init_from_env()
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    Dc, Wc, Df, Wf = a.arch
    args = factory.default_args(netdepth=Dc, netwidth=Wc, netdepth_fine=Df, netwidth_fine=Wf, no_reload=True, device=dev,
                                basedir="/nonexistent", lrate=5e-4)
    kw_train, _, _, grad_vars, _, _, render = factory.create_nerf(args)
    kw_train["network_fn"].load_state_dict(synth.nerf_state(Dc, Wc, 0, "coarse"))       # same initial weights on every rank
    kw_train["network_fine"].load_state_dict(synth.nerf_state(Df, Wf, 0, "fine"))
    render.idSpecificMod.load_state_dict(synth.style_state(0))
    render.texEncoder.load_state_dict(synth.tex_encoder_state(0))
    for dst, src in zip(render.expCodes_Sigma, synth.exp_sigma(0)):
        dst.data[:] = src.to(dev)
    kw_train.update(near=8.0, far=26.0)
    render.train()
    params = [p for p in grad_vars if p.requires_grad]
    opt = torch.optim.Adam(params, lr=args.lrate)
    bucket = mdist.GradBucket(params)
    K = synth.intrinsics(a.size, a.size)
    rng = np.random.default_rng(1000 + rank)                                               # per-rank data
    uv = torch.from_numpy(rng.uniform(0, 1, (512, 512, 3)).astype(np.float32)).to(dev)
    shape = synth.codes(rank)[0].to(dev)
    losses, times = [], []
    lm3d = torch.from_numpy(rng.uniform([-1.6, -2.0, -0.3], [1.6, 2.0, 1.2], (68, 3))).to(dev)       # this rank's identity: 68 3D landmarks
    image = torch.from_numpy(rng.uniform(0, 1, (a.size, a.size, 3)).astype(np.float32)).to(dev)      # ... and its target view
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    for it in range(a.steps):
        pose = mrays.pose_spherical(float(rng.uniform(-60, 60)), 0.0, 16.0)[:3, :4].to(dev)
        # run_train.py:306-330 on the device: landmark projection, landmark-biased + uniform pixels, rays through those pixels, colours
        batch, target, _ = steps.sample_train_batch(K, pose, lm3d, image, a.rays, precrop_frac=0.5 if it == 0 else 0.0, generator=gen)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = steps.train_step(render, kw_train, opt, bucket, a.size, a.size, K, batch, target, shape.expand(a.rays, -1), uv,
                                int(rng.integers(0, 20)), chunk=a.rays)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        losses.append(float(loss))
    # every rank must hold identical parameters
    h = hashlib.sha256()
    for p in params:
        h.update(p.detach().cpu().numpy().tobytes())
    digest = int(h.hexdigest()[:15], 16)
    same = True
    if mdist.active():
        t = torch.tensor([digest], dtype=torch.int64)
        if dist.get_backend() != "gloo":                      # RCCL needs every buffer — input AND outputs — on the GPU
            t = t.to(dev)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        same = all(int(g.item()) == digest for g in gathered)
    if rank == 0:
        print(json.dumps({"world": world, "rays_per_rank": a.rays, "steps": a.steps, "ms_per_step": [round(t * 1e3, 1) for t in times],
                          "loss_rank0": [round(l, 5) for l in losses], "parameters_identical_across_ranks": bool(same),
                          "bucket_floats": bucket.numel}), flush=True)
    if mdist.active():
        mdist.barrier(); dist.destroy_process_group()
    assert same, "ranks diverged"
    return same


if __name__ == "__main__":
    main()
