#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the rocprofv3 --pmc passes of tools/gpu_profile_round.sh (one pass per counter group over the
single-layer driver tools/pmc_layer.py, M = 196608, K = N = 1024).  Units and gfx950 corrections as /opt/skills/guides/
MI355X_MICROARCH.md ("HBM [CDNA4]") prescribes: FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE reports exactly half the bytes of a
wide coalesced streaming read (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE is uncalibrated -> taken as is and
cross-checked against the algorithmic write.  The kernel-source digest of the tree the passes ran on is recorded, so bench.py
can tell a stale file from a current one.

    python tools/make_traffic_json.py gpurun_out/r03 profiles/hbm_traffic.json
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mofanerf_amd import build

src, dst = sys.argv[1], sys.argv[2]
M, K, N = 196608, 1024, 1024


def rows(name):
    p = os.path.join(src, f"pmc_{name}.csv")
    with open(p) as f:
        return list(csv.DictReader(f))


GRID = (M // 256) * (N // 128) * 256        # work-items of the full-shape launch (the N = 128 control launches M // 256 workgroups)


def mean(rs, counter, grid=GRID):
    v = [float(r["Counter_Value"]) for r in rs if r["Counter_Name"] == counter and int(r["Grid_Size"]) == grid]
    if not v:
        raise SystemExit(f"no {counter} rows")
    return sum(v) / len(v), len(v)


fetch, n = mean(rows("FETCH_SIZE"), "FETCH_SIZE")
write, _ = mean(rows("WRITE_SIZE"), "WRITE_SIZE")
kname = rows("FETCH_SIZE")[0]["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].replace(", ", ",")
out = {
    "kernel": kname,
    "shape": f"M={M} points, K=N={K} (fine-net W x W layer; {2 * M * K * N / 1e9:.1f} GFLOP/launch)",
    "csrc_sha256": build.csrc_digest(),
    "launches_averaged": n,
    "fetch_size_raw_bytes": int(fetch * 1024),
    "fetch_size_corrected_x2_bytes": int(fetch * 1024 * 2),
    "write_size_bytes": int(write * 1024),
    "bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
    "algorithmic_read_bytes": (M * K + N * K + N) * 4,
    "algorithmic_write_bytes": M * N * 4,
    "algorithmic_bytes_per_launch": (M * K + N * K + N + M * N) * 4,
    "note": "one rocprofv3 --pmc pass per counter group over tools/pmc_layer.py (tools/gpu_profile_round.sh; per-dispatch counters; "
            "FETCH_SIZE / WRITE_SIZE in KiB); read side doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for "
            "16-B/lane streams); WRITE_SIZE uncalibrated per the guide (compare with algorithmic_write_bytes); mfma_busy_fraction = "
            "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); cross-check: TCC_MISS_sum x 128 B",
}
try:      # controls: N = 512 / 256 / 128 (4 / 2 / 1 feature tiles per point tile): the same activation read, smaller weight slabs
    ctl = {}
    for n_ in (512, 256, 128):
        g_ = (M // 256) * (n_ // 128) * 256
        f_, _ = mean(rows("FETCH_SIZE"), "FETCH_SIZE", grid=g_)
        w_, _ = mean(rows("WRITE_SIZE"), "WRITE_SIZE", grid=g_)
        ctl[f"n{n_}"] = {"fetch_size_corrected_x2_bytes": int(f_ * 2048), "write_size_bytes": int(w_ * 1024),
                         "algorithmic_read_bytes": (M * K + n_ * K + n_) * 4, "algorithmic_write_bytes": M * n_ * 4,
                         "weight_slab_bytes": n_ * K * 4}
    out["controls_same_activations_narrower_layers"] = ctl
except (OSError, SystemExit):
    pass
try:
    busy, _ = mean(rows("SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE"), "SQ_VALU_MFMA_BUSY_CYCLES")
    gui, _ = mean(rows("SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE"), "GRBM_GUI_ACTIVE")
    out["mfma_busy_fraction"] = round(busy / (1024 * gui / 8), 4)
except (OSError, SystemExit):
    pass
try:
    hit, _ = mean(rows("TCC_HIT_sum+TCC_MISS_sum"), "TCC_HIT_sum")
    miss, _ = mean(rows("TCC_HIT_sum+TCC_MISS_sum"), "TCC_MISS_sum")
    out["l2_hit_rate"] = round(hit / (hit + miss), 4)
    out["l2_miss_x128_bytes"] = int(miss * 128)
except (OSError, SystemExit):
    pass
try:
    conf, _ = mean(rows("SQ_LDS_BANK_CONFLICT+SQ_LDS_IDX_ACTIVE"), "SQ_LDS_BANK_CONFLICT")
    out["lds_bank_conflict_cycles"] = conf
except (OSError, SystemExit):
    pass
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
