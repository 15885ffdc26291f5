import sys, time, torch
sys.path.insert(0, '/root/repo')
from mofanerf_amd.model import EnDeUVmap
from mofanerf_amd import synth
m = EnDeUVmap().cuda(); m.load_state_dict(synth.tex_encoder_state(0))
x = torch.rand(1, 3, 512, 512, device='cuda')
core = m.encoder
def run(use_unfold, it=20):
    def f():
        y = (core._convs(x) if use_unfold else core.down1[0](x)).reshape(-1, 4096)
        out = core.decoding(core.mu(core.down2(y)))
        out.sum().backward()
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
print('miopen conv fwd+bwd ms:', round(run(False), 3))
print('unfold+gemm fwd+bwd ms:', round(run(True), 3))
with torch.no_grad():
    def g(u):
        for _ in range(3): (core._convs(x) if u else core.down1[0](x))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): (core._convs(x) if u else core.down1[0](x))
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20 * 1e3
    print('miopen conv fwd ms:', round(g(False), 3), ' unfold fwd ms:', round(g(True), 3))
