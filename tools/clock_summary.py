#!/usr/bin/env python3
"""Summarise tools/clock_probe.sh sample files next to the bench lines they were taken under: engine clock, socket power, rays/s and energy
per frame per arm (the table of profiles/r05_clocks_power.md / r06_clocks_power.md).

    python tools/clock_summary.py <label>=<samples.txt>:<bench.json> ...   (samples above 800 W count as "under load")
"""
import json
import re
import statistics
import sys


def samples(path, floor_w=800.0):
    out = []
    for line in open(path):
        c, p = re.search(r"\((\d+)Mhz\)", line), re.search(r"Power \(W\): ([\d.]+)", line)
        if c and p and float(p.group(1)) > floor_w:
            out.append((int(c.group(1)), float(p.group(1))))
    return out


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


def main(argv):
    print("| arm | samples under load | sclk MHz median (min / max) | socket power W p10 / p50 / p90 (mean) | rays/s | `roofline.frac` | J per frame (mean W x s/frame) |")
    print("|---|---|---|---|---|---|---|")
    for a in argv:
        label, rest = a.rsplit("=", 1)
        spath, bpath = rest.split(":", 1)
        s = samples(spath)
        d = json.loads([l for l in open(bpath).read().splitlines() if l.startswith("{")][-1])
        clk, pw = [c for c, _ in s], [p for _, p in s]
        mean = statistics.fmean(pw)
        print(f"| {label} | {len(s)} | {int(statistics.median(clk))} ({min(clk)} / {max(clk)}) | {pct(pw, .1):.0f} / {pct(pw, .5):.0f} / {pct(pw, .9):.0f} ({mean:.1f}) | "
              f"{d['value']:,.0f} | {d['roofline']['frac']:.4f} | {mean * d['ms_per_step'] / 1e3:,.0f} |")


if __name__ == "__main__":
    main(sys.argv[1:])
