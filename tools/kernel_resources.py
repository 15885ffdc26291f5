#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in libmofanerf_hip.so, read from the code object's own metadata (the AMDGPU
notes hipcc embeds: .vgpr_count, .agpr_count, .sgpr_count, .group_segment_fixed_size, .private_segment_fixed_size, spill counts).
This is the ISA's truth; rocprofv3's per-dispatch VGPR column on this image reports a different unit (100 for a 197-register
kernel), so the kernel-stats summaries take their register columns from here.  Runs without a GPU.

    python tools/kernel_resources.py [lib.so] [out.md]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def resources(so):
    MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
    notes = ""
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", so, os.path.join(d, "unused.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]          # one bundle per translation unit
        if not starts:
            raise SystemExit(f"no uncompressed offload bundle in {so}")
        for k, b in enumerate(starts):
            part, co = os.path.join(d, f"bundle{k}.bin"), os.path.join(d, f"gfx950_{k}.co")
            open(part, "wb").write(blob[b:(starts[k + 1] if k + 1 < len(starts) else len(blob))])
            ls = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", f"--input={part}"], capture_output=True, text=True)
            tgt = [l.strip() for l in ls.stdout.splitlines() if "gfx950" in l]
            if not tgt:
                continue
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}", f"--targets={tgt[0]}",
                            f"--output={co}"], check=True)
            notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, None])[1]
        name = get("name")
        if name is None:
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        out.append({"kernel": dem.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""),
                    "vgpr": int(get("vgpr_count")), "agpr": int(get("agpr_count")), "sgpr": int(get("sgpr_count")),
                    "lds_static": int(get("group_segment_fixed_size")), "scratch": int(get("private_segment_fixed_size")),
                    "vgpr_spill": int(get("vgpr_spill_count") or 0), "sgpr_spill": int(get("sgpr_spill_count") or 0)})
    return sorted(out, key=lambda r: r["kernel"])


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mofanerf_amd", "libmofanerf_hip.so")
    rs = resources(so)
    lines = [f"# Kernel resources of `{os.path.relpath(so, ROOT)}` (code-object metadata, gfx950)", "",
             "| kernel | VGPR | AGPR | SGPR | static LDS B | scratch B | VGPR spills | SGPR spills |", "|---|---|---|---|---|---|---|---|"]
    for r in rs:
        lines.append(f"| `{r['kernel'][:110]}` | {r['vgpr']} | {r['agpr']} | {r['sgpr']} | {r['lds_static']} | {r['scratch']} | {r['vgpr_spill']} | {r['sgpr_spill']} |")
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
        print("wrote", sys.argv[2], len(rs), "kernels")
    else:
        print(txt)


if __name__ == "__main__":
    main()
