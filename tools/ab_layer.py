#!/usr/bin/env python3
"""Interleaved A/B of layer-kernel arms: the product against arms of the measurement library (tools/build_measure.py), three
rounds in one process.
    python tools/ab_layer.py                     # product vs every arm that computes results
    python tools/ab_layer.py ring3,gap2          # product vs the named arms
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import microbench_layer as m
arms = sys.argv[1].split(",") if len(sys.argv) > 1 else [a for a in m.measure_lib().mofa_measure_arms().decode().split(",")
                                                          if a not in ("timeline", "sink_epilogue")]
m.ab(["product"] + arms, ((196608, 1024, 1024, 0), (196608, 256, 256, 0), (32768, 1024, 1024, 0)))
for rnd in range(3):
    print(f"round {rnd} backward-data W1024: {m.run_bwd(196608, 1024, 1024, iters=20)[1]:.1f} TFLOP/s", flush=True)
