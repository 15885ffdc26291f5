#!/usr/bin/env python3
"""profiles/hbm_traffic_chain.json from the rocprofv3 --pmc passes of tools/gpu_profile_chain.sh over tools/pmc_chain.py (the shipped
fine network 1024 x 10 on 196,608 points, one launch of k_net_chain): what bench.py quotes as `roofline.traffic` while the kernel
sources hash to what the passes were taken on.  Units / gfx950 corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes
(FETCH_SIZE / WRITE_SIZE in KiB; the read side doubled: 128-B requests of 16-B/lane streams are tallied at 64 B; WRITE_SIZE as is).

    python tools/make_traffic_chain_json.py gpurun_out/r04 profiles/hbm_traffic_chain.json"""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import build, schema

src, dst = sys.argv[1], sys.argv[2]
m = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(src, "pmc_chain_*.csv"))):
    for r in csv.DictReader(open(f)):
        m[r["Counter_Name"]].append(float(r["Counter_Value"]))
# every process's first two k_net_chain<0> dispatches are mofa_device_init's self-check (a 10 x 512 network on 4,096 points: ~1/40 of a
# benchmark launch in every counter) — they are not the kernel this file describes: keep the dispatches within a factor 5 of the largest
m = {c: [x for x in v if x >= 0.2 * max(v)] for c, v in m.items()}
launches = {c: len(v) for c, v in m.items()}
m = {c: sum(v) / len(v) for c, v in m.items()}
M, D, W = 196608, 10, 1024
mac = schema.mac_per_point(D, W, folded=True) - (W * 1 + (W // 2) * 3)                 # the MFMA layers (the two heads are k_head)
n_mfma = 2 * D + 5
# per layer: its input panels once, its weights once, its output once (the per-layer launches' accounting, summed over the chain)
act_out = (n_mfma - 1) * M * W * 4 + M * (W // 2) * 4
act_in = M * 64 * 4 + (n_mfma - 1 + 2) * M * W * 4
out = {"kernel": "mofa::k_net_chain<0>",
       "shape": f"fine network {W} x {D} ({n_mfma} MFMA layers) on M = {M} points (768 row tiles): {2 * mac * M / 1e12:.2f} TFLOP per launch",
       "csrc_sha256": build.csrc_digest(),
       "launches_averaged": min(launches.values()),
       "fetch_size_corrected_x2_bytes": int(m["FETCH_SIZE"] * 2048), "write_size_bytes": int(m["WRITE_SIZE"] * 1024),
       "bytes_per_launch": int(m["FETCH_SIZE"] * 2048 + m["WRITE_SIZE"] * 1024),
       "algorithmic_read_bytes": act_in + mac * 4,
       "algorithmic_write_bytes": act_out,
       "algorithmic_bytes_per_launch": act_in + mac * 4 + act_out,
       "algorithmic_note": "every layer's output written once and read once by its consumer(s) (the two skip layers read their block's input a second "
                           "time), the encoding panels read once, every weight once",
       "note": "one rocprofv3 --pmc pass per counter group (no tracing next to --pmc) over tools/pmc_chain.py; per-dispatch counters averaged over the "
               "four launches (mofa_device_init's two small self-check launches of the same kernel left out); mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}
if "GRBM_GUI_ACTIVE" in m:
    out["mfma_busy_fraction"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8), 4)
if "TCC_HIT_sum" in m:
    out["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
if "SQ_WAVE_CYCLES" in m:
    out["waves_parked_fraction"] = round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out)[:1200])
