#!/usr/bin/env python3
"""Per-kernel summary of a directory of `rocprofv3 --pmc` csv extracts (one counter group per file, as tools/gpu_profile_*.sh leave them):
matrix-pipe busy, waves parked / issuing, instruction counts, bytes through the L2s' fabric ports (FETCH_SIZE x2-corrected on gfx950,
/opt/skills/guides/MI355X_MICROARCH.md), L2 hit rate — as a markdown table.  Dispatches whose counters are below a fifth of the kernel's largest
are left out (mofa_device_init's self-check launches the same kernels on a small network).

    python tools/pmc_summary.py gpurun_out/r06 pmc_fit_ out.md "<what ran>" """
import collections, csv, glob, os, sys

src, prefix, dst, what = sys.argv[1], sys.argv[2], sys.argv[3], (sys.argv[4] if len(sys.argv) > 4 else "")
m = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, prefix + "*.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        m[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, v in sorted(m.items()):
    a = {}
    for c, x in v.items():
        keep = [y for y in x if y >= 0.2 * max(x)] if max(x) > 0 else x
        a[c] = (sum(keep), len(keep))
    n = max(t[1] for t in a.values())
    g = lambda c: a[c][0] if c in a else None
    cell = lambda x, f: (f % x) if x is not None else ""
    busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * g("GRBM_GUI_ACTIVE") / 8) if g("GRBM_GUI_ACTIVE") else None
    parked = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None
    fetch = g("FETCH_SIZE") * 2048 / a["FETCH_SIZE"][1] / 1e9 if g("FETCH_SIZE") else None
    write = g("WRITE_SIZE") * 1024 / a["WRITE_SIZE"][1] / 1e9 if g("WRITE_SIZE") else None
    hit = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if g("TCC_HIT_sum") else None
    rows.append(f"| `{k[:70]}` | {n} | {cell(busy, '%.4f')} | {cell(parked, '%.4f')} | {cell(fetch, '%.2f')} | {cell(write, '%.2f')} | {cell(hit, '%.3f')} |")
with open(dst, "w") as f:
    f.write(f"# rocprofv3 --pmc summary\n\n{what}\n\nOne pass per counter group (no tracing next to --pmc); FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); per-dispatch averages; "
            "matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); waves parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES.\n\n"
            "| kernel | dispatches | matrix pipe busy | waves parked | GB read / dispatch (x2) | GB written / dispatch | L2 hit |\n|---|---|---|---|---|---|---|\n")
    f.write("\n".join(rows) + "\n")
print(open(dst).read())
