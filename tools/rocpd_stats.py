#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace as the `--stats` table: calls, total/avg duration, share.
Register / LDS columns come from the code object itself (tools/kernel_resources.py — the ISA's .vgpr_count etc.): the
per-dispatch `vgpr_count` field of this rocprofv3 build reports another unit (100 for the 197-VGPR layer kernel).
    python tools/rocpd_stats.py gpurun_out/prof_r01/bench_results.db profiles/r01_kernel_stats.md "<command line>"
"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources

db, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
disp = cur.execute("select name, min(lds_size), max(lds_size), min(grid_x), max(grid_x), min(workgroup_x) from kernels group by name").fetchall()
meta = {r[0]: r[1:] for r in disp}
norm = lambda n: n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
try:
    isa = {r["kernel"]: r for r in kernel_resources.resources(os.path.join(kernel_resources.ROOT, "mofanerf_amd", "libmofanerf_hip.so"))}
except (SystemExit, OSError, Exception) as e:      # noqa: BLE001  (no llvm tools: leave the columns empty rather than wrong)
    print("kernel_resources unavailable:", e)
    isa = {}
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
    f.write("VGPR / AGPR / SGPR / spills: code-object metadata of `libmofanerf_hip.so` (tools/kernel_resources.py), not rocprofv3's dispatch field.\n\n")
    f.write("| kernel | calls | total s | avg us | % GPU time | VGPR | AGPR | SGPR | SGPR spills | LDS B (dispatch) | grid (WGs) |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
    for name, calls, tot, avg, pct in rows:
        m = meta.get(name, (None,) * 5)
        short = norm(name)
        r = isa.get(short, {})
        grid = f"{(m[2] or 0) // max(1, (m[4] or 1))}..{(m[3] or 0) // max(1, (m[4] or 1))}"
        lds = f"{m[0]}" if m[0] == m[1] else f"{m[0]}..{m[1]}"
        f.write(f"| `{short[:90]}` | {calls} | {tot / 1e6:.4f} | {avg:.3f} | {pct:.3f} | {r.get('vgpr', '')} | {r.get('agpr', '')} | {r.get('sgpr', '')} | "
                f"{r.get('sgpr_spill', '')} | {lds} | {grid} |\n")
print("wrote", out)
