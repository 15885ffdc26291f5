"""Documents (on the CPU oracle, no GPU) why end-to-end parity is tiered in tests/harness.py::compare_render:
the reference's own outputs move by far more than 1e-4 under perturbations at the fp32 rounding floor."""
import numpy as np
import torch

from mofanerf_amd import synth
from oracle import mofa_oracle as orc

T = torch.from_numpy


def _setup(golden):
    g = golden("e2e_small.npz")
    arch = [int(v) for v in g["arch"]]
    o = orc.OracleRenderer(synth.nerf_state(arch[0], arch[1], 0, "coarse"), synth.nerf_state(arch[2], arch[3], 0, "fine"),
                           synth.style_state(0), synth.exp_sigma(0), netchunk=1 << 20)
    o.exp_sigma.append(T(g["exp"]))
    H = int(g["H"])
    ro, rd = orc.get_rays(H, H, g["K"], T(g["c2w"]))
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    vd = rd / torch.norm(rd, dim=-1, keepdim=True)
    return g, o, ro, rd, vd


def test_one_ulp_of_sample_position_moves_rgb_by_more_than_1e_5(golden):
    """Nudge every fine sample position by +-1 ulp (what two correct fp32 implementations of sample_pdf differ by):
    pts = o + d*z feeds sin/cos(2^9 x), so per-ray RGB moves by 1e-5..1e-3 — the tier-B bound of compare_render."""
    g, o, ro, rd, vd = _setup(golden)
    zf = T(g["z_fine"])
    rng = np.random.default_rng(0)
    sign = T(rng.choice([-1.0, 1.0], size=zf.shape).astype(np.float32))
    zp = torch.nextafter(zf, zf + sign)
    zp, _ = torch.sort(zp, -1)
    outs = []
    with torch.no_grad():
        for z in (zf, zp):
            pts = ro[:, None, :] + rd[:, None, :] * z[:, :, None]
            raw = o.run_network(pts, vd, o.fine, T(g["bm"]), T(g["tex"]), 20)
            outs.append(orc.raw2outputs(raw, z, rd)[0])
    np.testing.assert_array_equal(outs[0].numpy(), g["rgb"].reshape(-1, 3))      # unperturbed == the reference fixture
    d = (outs[1] - outs[0]).abs().max(-1)[0]
    print(f"1-ulp z perturbation: max |d rgb| = {float(d.max()):.2e}, median = {float(d.median()):.2e}")
    assert float(d.max()) > 1e-5          # far above the 2e-6 a bit-identical-position ray shows
    assert float(d.max()) < 1e-3          # tier-B bound


def test_sample_pdf_is_discontinuous_at_the_fp32_noise_floor(golden):
    """A 1e-7 relative perturbation of the coarse weights moves some of the reference's new samples by > 1e-3
    (denom<1e-5 -> 1 branch of tools/run_nerf_helpers.py:243 for the empty bins of near-opaque rays)."""
    g = golden("e2e_small.npz")
    z, w = T(g["z_coarse"]), T(g["weights_coarse"])
    rng = np.random.default_rng(0)
    wp = w * T((1 + 1e-7 * rng.standard_normal(w.shape)).astype(np.float32))
    zmid = .5 * (z[:, 1:] + z[:, :-1])
    u = torch.linspace(0., 1., 64)
    a, b = orc.sample_pdf(zmid, w[:, 1:-1], u), orc.sample_pdf(zmid, wp[:, 1:-1], u)
    assert np.array_equal(a.numpy(), g["z_samples"])
    moved = ((a - b).abs() > 1e-3).any(-1).float().mean()
    print(f"rays with a sample moved by > 1e-3 under 1e-7 weight noise: {float(moved):.1%}")
    assert float(moved) > 0.02
