#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace as the `--stats` table: calls, total/avg duration, share.
    python tools/rocpd_stats.py gpurun_out/prof_r01/bench_results.db profiles/r01_kernel_stats.md "<command line>"
"""
import sqlite3
import sys

db, out, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
disp = cur.execute("select name, min(vgpr_count), min(accum_vgpr_count), min(sgpr_count), min(lds_size), min(grid_x), "
                   "max(grid_x), min(workgroup_x) from kernels group by name").fetchall()
meta = {r[0]: r[1:] for r in disp}
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
    f.write("| kernel | calls | total s | avg us | % GPU time | arch VGPR | accum VGPR | SGPR | LDS B | grid (WGs x 256 thr) |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for name, calls, tot, avg, pct in rows:
        m = meta.get(name, (None,) * 7)
        short = name.replace("mofa::(anonymous namespace)::", "mofa::").split("(")[0].replace("void ", "")
        grid = f"{(m[4] or 0) // max(1, (m[6] or 1))}..{(m[5] or 0) // max(1, (m[6] or 1))}"
        f.write(f"| `{short[:80]}` | {calls} | {tot / 1e6:.4f} | {avg:.3f} | {pct:.3f} | {m[0]} | {m[1]} | {m[2]} | {m[3]} | {grid} |\n")
print("wrote", out)
