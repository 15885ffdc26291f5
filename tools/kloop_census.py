#!/usr/bin/env python3
"""Instruction census of the MFMA loops of libmofanerf_hip.so, read from its own code objects (no GPU needed): for every innermost loop
(backward branch) that holds >= 16 v_mfma instructions — the K loops of the layer / chained / persistent / weight-gradient kernels — how
many MFMA, LDS, LDS-DMA (VMEM), scalar and VECTOR instructions it contains, and which vector opcodes.

Why: on gfx950 a vector instruction between two MFMAs is not free — it costs the matrix pipe 6-13 cycles whatever it computes
(profiles/r06_probe_dual_issue.md) — so the K loops are written to contain none but the MFMAs (profiles/r06_ab_kloop_addr.md), and
tests/test_abi_cpu.py holds them to it.

    python tools/kloop_census.py [lib.so] [out.md]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _category(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "vector"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    return "scalar" if op.startswith("s_") else "other"


def disassemble(so):
    """[(demangled kernel name, [(opcode, operands)], {label: instruction index})] of every gfx950 kernel in `so`."""
    out = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", so, os.path.join(d, "unused.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for k, b in enumerate(starts):
            part, co = os.path.join(d, f"bundle{k}.bin"), os.path.join(d, f"gfx950_{k}.co")
            open(part, "wb").write(blob[b:(starts[k + 1] if k + 1 < len(starts) else len(blob))])
            ls = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={part}"], capture_output=True, text=True)
            tgt = [l.strip() for l in ls.stdout.splitlines() if "gfx950" in l]
            if not tgt:
                continue
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--targets={tgt[0]}", f"--output={co}"], check=True)
            text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--symbolize-operands", co], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]+ <(_Z\S+)>:", line)
                if m:
                    dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                    cur = (dem.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], [], {})
                    out.append(cur)
                    continue
                if cur is None:
                    continue
                m = re.match(r"^[0-9a-f]+ <(L\d+)>:", line)
                if m:
                    cur[2][m.group(1)] = len(cur[1])
                    continue
                m = re.match(r"^\s+(\S+)\s*(.*?)\s*(//.*)?$", line)
                if m and not line.startswith("Disassembly") and m.group(1)[0].isalpha():
                    cur[1].append((m.group(1), m.group(2)))
    return out


def census(so):
    """{kernel: [{"mfma": n, "vector": n, "lds": n, "vmem": n, "scalar": n, "vector_ops": {opcode: n}}, ...]} — one record per innermost MFMA loop."""
    res = {}
    for name, ins, labels in disassemble(so):
        loops = []
        for i, (op, args) in enumerate(ins):
            if op.startswith(("s_cbranch", "s_branch")) and args.strip() in labels and labels[args.strip()] <= i:
                loops.append((labels[args.strip()], i))
        dense = [(a, b) for a, b in loops if sum(1 for op, _ in ins[a:b + 1] if op.startswith("v_mfma")) >= 16]
        recs = []
        for a, b in dense:
            if any((x, y) != (a, b) and x >= a and y <= b for x, y in dense):
                continue                                   # not innermost
            c = collections.Counter(_category(op) for op, _ in ins[a:b + 1])
            v = collections.Counter(op for op, _ in ins[a:b + 1] if _category(op) == "vector")
            recs.append({"mfma": c["mfma"], "vector": c["vector"], "lds": c["lds"], "vmem": c["vmem"], "scalar": c["scalar"], "vector_ops": dict(v)})
        if recs:
            res[name] = recs
    return res


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "mofanerf_amd", "libmofanerf_hip.so")
    rows = ["# MFMA loops of " + os.path.basename(so) + " (tools/kloop_census.py: innermost loops with >= 16 v_mfma, from the code objects)", "",
            "| kernel | MFMA | vector | LDS | LDS-DMA / VMEM | scalar | vector opcodes |", "|---|---|---|---|---|---|---|"]
    for k, recs in sorted(census(so).items()):
        for r in recs:
            rows.append(f"| `{k[:80]}` | {r['mfma']} | {r['vector']} | {r['lds']} | {r['vmem']} | {r['scalar']} | "
                        + ", ".join(f"{n} `{o}`" for o, n in sorted(r["vector_ops"].items(), key=lambda t: -t[1])) + " |")
    text = "\n".join(rows) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)
