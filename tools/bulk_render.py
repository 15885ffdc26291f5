#!/usr/bin/env python3
"""BASELINE config 4 — render_refine_trainSet.py's job shape: identities x expressions x views at half resolution,
identities sharded across the GPUs of a node (the reference's begin_person/end_person knob), one process per GPU, no
data-path collective, resumable (finished PNGs are skipped).  Synthetic identities (seeded codes + UV maps).

  python tools/bulk_render.py --out /tmp/rf --identities 4 --expressions 2 --views 2 --size 256
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bulk_render.py ...
"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import dist as mdist, factory, rays, steps, synth


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--identities", type=int, default=8)
    ap.add_argument("--expressions", type=int, default=2)
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--arch", type=int, nargs=4, default=[8, 256, 10, 1024])
    a = ap.parse_args(argv)
    rank, world, local = mdist.init_from_env()
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    Dc, Wc, Df, Wf = a.arch
    args = factory.default_args(netdepth=Dc, netwidth=Wc, netdepth_fine=Df, netwidth_fine=Wf, no_reload=True, device=dev,
                                basedir="/nonexistent")
    _, kw, _, _, _, _, render = factory.create_nerf(args)
    kw["network_fn"].load_state_dict(synth.nerf_state(Dc, Wc, 0, "coarse"))
    kw["network_fine"].load_state_dict(synth.nerf_state(Df, Wf, 0, "fine"))
    render.idSpecificMod.load_state_dict(synth.style_state(0))
    render.texEncoder.load_state_dict(synth.tex_encoder_state(0))
    kw.update(near=8.0, far=26.0)
    render.eval()
    K = synth.intrinsics(a.size, a.size)
    angles = np.linspace(-60, 60, a.views)

    def render_identity(ident):
        shape = synth.codes(ident)[0].to(dev)
        uv = torch.from_numpy(np.random.default_rng(ident).uniform(0, 1, (1, 512, 512, 3)).astype(np.float32)).to(dev)
        d = os.path.join(a.out, f"{ident:03d}")
        os.makedirs(d, exist_ok=True)
        n = 0
        with torch.no_grad():
            for e in range(a.expressions):
                for v, ang in enumerate(angles):
                    pose = rays.pose_spherical(float(ang), 0.0, 16.0)[None]
                    r = render.render_path(pose, [a.size, a.size, float(K[0][0])], K, args.chunk, kw, uvMap=uv,
                                           expType=torch.tensor([e]), savedir=d, shapeCodes=shape, name=f"{e:02d}_{v}")
                    n += 0 if isinstance(r[0], int) else 1          # (0, 0) = already on disk
        return n

    from mofanerf_amd.io import PngSink
    torch.cuda.synchronize(); t0 = time.perf_counter()
    render.png_sink = PngSink(workers=4)                 # one sink for the job: PNG encoding overlaps the following frames
    done = steps.bulk_render_identities(render, kw, list(range(a.identities)), render_identity, rank, world)
    render.png_sink.close(); render.png_sink = None      # every file is on disk inside the timed region
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    total = torch.tensor([float(sum(done))], dtype=torch.float64)
    if mdist.active():
        if torch.distributed.get_backend() == "nccl":
            total = total.to(dev)                      # keep the tensor the collective reduces INTO
        torch.distributed.all_reduce(total)
    dt = mdist.barrier_max(dt, dev)
    if rank == 0:
        imgs = int(total.item())
        print(json.dumps({"images_rendered_rank0": sum(done), "images_rendered_total": imgs, "world": world,
                          "seconds": round(dt, 3), "size": a.size,
                          "rays_per_s_rank0": round(sum(done) * a.size * a.size / dt, 1),
                          "rays_per_s_total": round(imgs * a.size * a.size / dt, 1)}), flush=True)
    return sum(done)


if __name__ == "__main__":
    main()
