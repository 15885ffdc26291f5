#!/usr/bin/env python3
"""Launch the OPT-IN fp16x3 layer kernel (pre-split piece panels) a few times at M=196608, K=N=1024 on real piece-panel data so
that rocprofv3 --pmc can attribute counters to individual dispatches.  `--summarise DIR...` turns the counter CSVs into a table."""
import csv, glob, os, sys
if len(sys.argv) > 1 and sys.argv[1] == "--summarise":
    acc = {}
    for d in sys.argv[2:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_layer_split" not in r["Kernel_Name"]:
                    continue
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print("| counter | per launch (mean of the timed launches) |\n|---|---|")
    for k in sorted(acc):
        v = acc[k][-4:]
        print(f"| {k} | {sum(v) / len(v):.6g} |")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench_layer as mb
ms, tf = mb.run_split_hh(196608, 1024, 1024, iters=4)
print(f"done: {ms:.3f} ms, {tf:.1f} TFLOP/s")
