#!/usr/bin/env python3
"""What happens to RESIDENT persistent workgroups while several processes share one device?  (measurement library: k_xcc_watch)

    python tools/probe_shared_device.py [procs=8] [seconds=1.0] [launches=3]

Every process launches `launches` times 512 workgroups that hold a slot like k_net_chain's (two per CU) for `seconds` and poll XCC_ID / HW_ID /
the real-time clock.  Per process and launch: workgroups whose XCC_ID changed while they ran (k_net_chain reads it ONCE and keeps producer and
consumer of a row tile on that XCD's L2), whose HW_ID changed (another CU / SIMD: restored elsewhere), and the longest time a workgroup spent off
the chip between two polls (compute-wave save / restore).  procs = 1 is the control.  profiles/r06_shared_device_chain.md
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(seconds, launches):
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from mofanerf_amd import lib
    import build_measure
    Lm = build_measure.load()
    out = torch.zeros(512 * 8, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    res = []
    for _ in range(launches):
        out.zero_()
        t0 = time.perf_counter()
        build_measure.check(Lm, Lm.mofa_measure_xcc_watch(out.data_ptr(), 512, int(seconds * 1e8), lib.stream()), "xcc_watch")
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        o = out.cpu().view(512, 8)
        per_xcc = [int((o[:, 0] == x).sum()) for x in range(8)]
        res.append({"wall_s": round(wall, 2), "wg_xcc_changed": int((o[:, 1] > 0).sum()), "wg_hwid_changed": int((o[:, 2] > 0).sum()),
                    "max_gap_ms": round(float(o[:, 3].max()) / 1e5, 3), "wg_gap_over_1ms": int((o[:, 3] > 1e5).sum()),
                    "wg_gap_over_100ms": int((o[:, 3] > 1e7).sum()), "polls_min": int(o[:, 4].min()), "polls_max": int(o[:, 4].max()),
                    "entry_spread_ms": round(float(o[:, 6].max() - o[:, 6].min()) / 1e5, 2),
                    "run_ms_max": round(float((o[:, 7] - o[:, 6]).max()) / 1e5, 2), "wg_per_xcc_at_entry": per_xcc})
    print(json.dumps(res), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child(float(sys.argv[2]), int(sys.argv[3]))
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(seconds), str(launches)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
          for _ in range(procs)]
    tot = {"launches": 0, "wg_xcc_changed": 0, "wg_hwid_changed": 0, "wg_gap_over_1ms": 0, "wg_gap_over_100ms": 0, "max_gap_ms": 0.0, "wall_s_max": 0.0, "run_ms_max": 0.0}
    for i, p in enumerate(ps):
        o, e = p.communicate(timeout=900)
        line = [l for l in o.splitlines() if l.startswith("[")]
        if not line:
            print(f"process {i}: no result (rc {p.returncode}): {e[-400:]}")
            continue
        for r in json.loads(line[-1]):
            tot["launches"] += 1
            for k in ("wg_xcc_changed", "wg_hwid_changed", "wg_gap_over_1ms", "wg_gap_over_100ms"):
                tot[k] += r[k]
            tot["max_gap_ms"] = max(tot["max_gap_ms"], r["max_gap_ms"])
            tot["wall_s_max"] = max(tot["wall_s_max"], r["wall_s"])
            tot["run_ms_max"] = max(tot["run_ms_max"], r["run_ms_max"])
        print(f"process {i}: {o.splitlines()[-1][:600]}")
    print(f"SUMMARY procs={procs} seconds={seconds}: {json.dumps(tot)}")


if __name__ == "__main__":
    main()
