"""CPU-side checks of the C ABI: the library loads without a GPU and exports every symbol the header declares;
argument validation returns MOFA_EINVAL with a message (no compute is launched)."""
import ctypes
import os
import re

from mofanerf_amd import build, lib, schema

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mofanerf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mofa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    so = build.build()
    L = ctypes.CDLL(so)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/mofanerf_hip.h but not exported"
    assert sorted(lib.SIGNATURES) == syms, "python binding and header disagree"


def test_library_exports_nothing_but_the_declared_abi():
    """Built with -fvisibility=hidden: the dynamic symbol table holds exactly the functions of include/mofanerf_hip.h — no
    mofa_internal_* hand-offs between translation units, no measurement entry points (those live in build_arms/libmofanerf_measure.so,
    tools/build_measure.py), and the product sources carry no measurement arms."""
    import subprocess
    so = build.build()
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    funcs = sorted(l.split()[-1] for l in out.splitlines() if l.split()[1] in ("T", "W"))
    assert funcs == header_symbols(), sorted(set(funcs) ^ set(header_symbols()))
    csrc = os.path.join(ROOT, "mofanerf_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            src = open(os.path.join(csrc, f)).read()
            for arm in ("MOFA_TIMELINE", "MOFA_ABLATE", "MOFA_SPLIT_FAKE", "MOFA_SETPRIO", "k_layer_persist", "k_layer_ring3", "k_mfma_peak_probe", "k_layer_split", "MOFA_GEMM"):
                assert arm not in src, (f, arm)


def test_plan_sizes_and_argument_validation():
    L = lib.load()
    assert L.mofa_abi_version() == 1
    for D, W in ((8, 256), (10, 1024), (8, 64)):
        s = lib.NetShape(D, W)
        assert L.mofa_net_num_layers(s) == 2 * D + 7 == len(schema.nerf_layers(D, W))
        Wp, Hp = (W + 63) // 64 * 64, (W // 2 + 63) // 64 * 64
        n_plain = 3 + 4 + 4 + 2 * (D - 6)
        want = Wp * 64 + n_plain * Wp * Wp + 2 * Wp * Wp + 2 * Wp * 2 * Wp + Hp * Wp + Wp + 3 * Hp
        assert want <= L.mofa_net_packed_floats(s) <= want + 64 * (2 * D + 7)
        assert L.mofa_net_folded_floats(s) == (2 * D + 4) * Wp + 8
        assert L.mofa_net_workspace_floats(s, 1000, 10) == 4 * 1024 * Wp + 10 * Hp + 64
    assert L.mofa_net_num_layers(lib.NetShape(3, 256)) == -1
    assert L.mofa_layer_forward(None, 16, None, 0, None, None, 0, 1, None, 256, 64, 1, None) == -1
    assert b"null pointer" in L.mofa_last_error()
    assert L.mofa_composite_forward(1, 1, 0, 1, None, 4, 1, 0, 1, 1, 1, 1, 1, None) == -1
    assert b"S >= 2" in L.mofa_last_error()
    assert L.mofa_sample_pdf_merge(1, 0, 1, 1, 0, 4, 5000, 5000, 1, 1, 1, None) == -1      # 3 S + Ni floats of LDS per ray: 64 KiB
    assert b"3 S + Ni" in L.mofa_last_error()
    assert L.mofa_composite_backward(1, 1, 0, 1, None, 4, 20000, 0, 1, None, None, None, None, 1, None, None) == -1


def test_product_never_imports_the_oracle():
    """The product package must not reach into oracle/ (nor any CPU fallback): grep the sources."""
    pkg = os.path.join(ROOT, "mofanerf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "mofa_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_hot_kernels_fit_their_occupancy_without_scratch():
    """ISA-level regression guard, read from the built library's own code objects (tools/kernel_resources.py; no GPU needed): the MFMA
    kernels are written for TWO workgroups per CU (<= 256 registers per lane in the unified VGPR file) and must not touch scratch —
    a compiler or source change that spills them would halve the matrix pipe's feed long before any parity test noticed."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    rs = {r["kernel"]: r for r in kernel_resources.resources(build.build())}
    hot = [k for k in rs if k.startswith(("mofa::k_layer<", "mofa::k_mlp_fused", "mofa::k_wgrad<"))]
    assert len(hot) >= 15, sorted(rs)
    for k in hot:
        r = rs[k]
        assert r["scratch"] == 0 and r["vgpr_spill"] == 0, (k, r)
        assert r["vgpr"] + r["agpr"] <= 256, (k, r)
    dom = rs["mofa::k_layer<128, false, false, false, true, mofa::ShippedPolicy>"]
    assert dom["vgpr"] <= 200 and dom["agpr"] == 0, dom            # 197 since round 2; the refactor into mofa_layer.h + policy did not move it
    for k, r in rs.items():                                        # the ray-side kernels run many rays per CU: keep them light
        if k.startswith(("mofa::k_composite", "mofa::k_sample_pdf_merge", "mofa::k_get_rays")):
            assert r["vgpr"] <= 128 and r["scratch"] == 0, (k, r)
