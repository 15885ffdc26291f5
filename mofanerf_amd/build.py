"""In-tree build of ``libmofanerf_hip.so`` for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmofanerf_hip.so")
SOURCES = ["mofa_mlp.hip", "mofa_rays.hip", "mofa_bwd.hip", "mofa_net.hip"]
# -fvisibility=hidden: the library exports exactly the C ABI of include/mofanerf_hip.h (declared under a visibility pragma there);
# the mofa_internal_* hand-offs between the translation units stay out of the dynamic symbol table
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden", "-Wall",
         "-Wno-unused-function"]
LINK_FLAGS = ["-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]


def csrc_digest() -> str:
    """sha256 over the PRODUCT's kernel sources (names + contents of csrc/*.hip, csrc/*.h and the C-ABI header; csrc/measure/ is not
    part of libmofanerf_hip.so): identifies WHICH kernels a profile was taken on (profiles/hbm_traffic.json records it; bench.py
    quotes those counters only for the same digest)."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, n) for n in os.listdir(CSRC) if n.endswith((".hip", ".h"))]
    files.append(os.path.join(HERE, "..", "include", "mofanerf_hip.h"))
    for f in sorted(files, key=lambda q: os.path.relpath(q, HERE)):
        h.update(os.path.relpath(f, HERE).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".map"))] + [os.path.join(HERE, "..", "include", "mofanerf_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return OUT
    cmd = [hipcc()] + FLAGS + LINK_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
