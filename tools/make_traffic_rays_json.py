#!/usr/bin/env python3
"""profiles/hbm_traffic_rays.json from the rocprofv3 --pmc passes of tools/gpu_profile_rays.sh over tools/pmc_rays.py (the ray kernels at the
benchmark's shapes, 196,608 rays per launch): what bench.py quotes as `roofline_hbm[*].traffic` while the kernel sources hash to what the
passes were taken on.  Units / gfx950 corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in KiB; the
read side doubled: 128-B requests of 16-B/lane streams are tallied at 64 B — exact for the float4 loads of `raw`, an upper bound for the
4-byte-per-lane z / weights rows; WRITE_SIZE as is).

    python tools/make_traffic_rays_json.py gpurun_out/r06 profiles/hbm_traffic_rays.json"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mofanerf_amd import build

src, dst = sys.argv[1], sys.argv[2]
m = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pmc_rays_*.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        m[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
R = 196608
out = {"csrc_sha256": build.csrc_digest(), "rays_per_launch": R, "launches_averaged": 4, "kernels": {},
       "note": "one rocprofv3 --pmc pass per counter group (no tracing next to --pmc) over tools/pmc_rays.py; per-dispatch counters averaged over the "
               "four launches of each kernel; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as reported"}
for kind, per_ray in bench.HBM_KINDS.items():
    name = bench.KERNELS[kind][0]
    c = next((v for k, v in m.items() if k.replace("mofa::", "") == name.replace("mofa::", "")), None)
    if c is None:
        continue
    a = {k: sum(v) / len(v) for k, v in c.items()}
    rec = {"rays_per_launch": R, "algorithmic_bytes_per_ray": per_ray, "algorithmic_bytes_per_launch": per_ray * R}
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        rec.update(fetch_size_corrected_x2_bytes=int(a["FETCH_SIZE"] * 2048), write_size_bytes=int(a["WRITE_SIZE"] * 1024),
                   bytes_per_launch=int(a["FETCH_SIZE"] * 2048 + a["WRITE_SIZE"] * 1024))
        rec["traffic_over_algorithmic"] = round(rec["bytes_per_launch"] / rec["algorithmic_bytes_per_launch"], 3)
    if "TCC_HIT_sum" in a:
        rec["l2_hit_rate"] = round(a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"]), 4)
    if "SQ_WAVE_CYCLES" in a:
        rec["waves_parked_fraction"] = round(a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"], 4)
        if "SQ_ACTIVE_INST_ANY" in a:
            rec["waves_issuing_fraction"] = round(a["SQ_ACTIVE_INST_ANY"] / a["SQ_WAVE_CYCLES"], 4)
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"):
        if k in a:
            rec[k.lower() + "_per_ray"] = round(a[k] / R, 1)
    if "SQ_INSTS_VALU" in a:      # one ray = one wavefront: its vector instructions x 4 cycles (wave64 on a 16-lane SIMD) over 1,024 SIMDs at 2.39 GHz
        rec["valu_issue_floor_us_per_launch"] = round(a["SQ_INSTS_VALU"] * 4 / (1024 * 2.39e9) * 1e6, 1)
    out["kernels"][name] = rec
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out)[:2000])
