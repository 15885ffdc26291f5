#!/usr/bin/env python3
"""profiles/hbm_traffic_fused.json from the rocprofv3 --pmc passes of tools/gpu_profile_fused.sh over tools/pmc_fused.py (the shipped
coarse network 256 x 8 on 196,608 points, one launch of the persistent kernel k_mlp_fused): the hbm_traffic.json of the ≤256-wide
kernel.  Units / gfx950 corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in KiB; the read
side doubled: 128-B requests of 16-B/lane streams are tallied at 64 B; WRITE_SIZE taken as is).

    python tools/make_traffic_fused_json.py gpurun_out/r04 profiles/hbm_traffic_fused.json"""
import csv, glob, json, os, re, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import build, schema

src, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pmc_fused_*.csv"))):
    for r in csv.DictReader(open(f)):
        k = re.search(r"(k_mlp_\w+(<[^>]*>)?)", r["Kernel_Name"]).group(1)
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
M, D, W = 196608, 8, 256
mac = schema.mac_per_point(D, W, folded=True) - (W * 1 + (W // 2) * 3)          # the MFMA layers (heads are k_head)
layers = 2 * D + 5
out = {"shape": f"coarse network {W} x {D} ({layers} MFMA layers) on M = {M} points (1,536 tiles of 128): {2 * mac * M / 1e9:.1f} GFLOP per launch",
       "csrc_sha256": build.csrc_digest(),
       "algorithmic_bytes_per_launch": {"points_in": M * 12, "raw_out_by_k_head": 0, "weights_once": int(mac * 4),
                                        "note": "a kernel that kept a tile's activations on chip would move only the points in, the sigma features "
                                                "(alpha head), the view layer's output and each weight once; this kernel round-trips every layer's "
                                                f"output through L2 / MALL by design: {layers} x M x {W} x 4 B = {layers * M * W * 4 / 1e9:.2f} GB written and read"},
       "kernels": {}}
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    e = {"launches_averaged": len(next(iter(v.values())))}
    if "FETCH_SIZE" in m:
        e.update(fetch_size_corrected_x2_bytes=int(m["FETCH_SIZE"] * 2048), write_size_bytes=int(m["WRITE_SIZE"] * 1024),
                 bytes_per_launch=int(m["FETCH_SIZE"] * 2048 + m["WRITE_SIZE"] * 1024))
    if "GRBM_GUI_ACTIVE" in m:
        e["mfma_busy_fraction"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8), 4)
    if "TCC_HIT_sum" in m:
        e["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
    if "SQ_WAVE_CYCLES" in m:
        e.update(waves_parked_fraction=round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4), issue_stall_fraction=round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4),
                 issuing_fraction=round(m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"], 4))
    if "SQ_INSTS_MFMA" in m:
        e.update(insts_mfma=int(m["SQ_INSTS_MFMA"]), insts_valu_non_mfma=int(m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]), insts_lds=int(m["SQ_INSTS_LDS"]),
                 insts_vmem=int(m["SQ_INSTS_VMEM"]))
    if "SQ_LDS_BANK_CONFLICT" in m:
        e["lds_bank_conflict_fraction"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 4)
    out["kernels"][k] = e
out["note"] = ("one rocprofv3 --pmc pass per counter group (no tracing next to --pmc) over tools/pmc_fused.py; per-dispatch counters averaged over the "
               "launches of each kernel; mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out)[:1500])
