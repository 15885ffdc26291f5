import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mofanerf_amd import synth, lib, rays as mrays
dev = torch.device('cuda')
render, kw, _ = bench.build_product(dev)
bm, tex, exp = (t.to(dev) for t in synth.codes(0))
H = W = 512
K = synth.intrinsics(H, W)
c2w = bench.pose_spherical(20.0, 0.0, 16.0)[:3, :4].contiguous().to(dev)
L = lib.load()
def rays_rows(r0, r1):
    n = (r1 - r0) * W
    o, d, v = (torch.empty(n, 3, device=dev) for _ in range(3))
    lib.check(L.mofa_get_rays(H, W, float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2]), lib.ptr(c2w), r0 * W, n, lib.ptr(o), lib.ptr(d), lib.ptr(v), lib.stream()), 'rays')
    return torch.stack([o, d], 0)
def run(r0, r1, reps):
    r = rays_rows(r0, r1)
    with torch.no_grad():
        render.render_fitting(H, W, K, chunk=196608, rays=r, shapeCodes=bm, uvCodes=tex, expType=20, expCodes=exp, **kw)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(reps):
            render.render_fitting(H, W, K, chunk=196608, rays=r, shapeCodes=bm, uvCodes=tex, expType=20, expCodes=exp, **kw)
        torch.cuda.synchronize()
    return (time.time() - t) / reps
full = run(0, 512, 1)
for n in (2, 4, 8):
    rows = 512 // n
    t = run(192, 192 + rows, 2)
    print(f"N={n}: one rank's {rows} rows: {t:.3f} s  vs full frame / {n} = {full / n:.3f} s  -> strong-scaling efficiency of the compute part {full / n / t * 100:.1f} %", flush=True)
print(f"full frame {full:.3f} s")
