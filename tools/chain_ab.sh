#!/bin/bash
# A/B of launch-form options, one arm per "NAME=VALUE" environment assignment (e.g. MOFA_CHAIN=0; the round-5 arms MOFA_CHAIN_GANG=n and
# MOFA_CHAIN_NSPLIT=2 existed only in the commits named in profiles/r05_ab_chain_gang.txt):
# frame rate + live roofline, socket power / clock, and three PMC passes (fabric fetch bytes, L2 hit rate, matrix-pipe busy).
#   bash tools/chain_ab.sh <tag> ARM [ARM...]        # an arm "X=1" runs with X=1 exported; "base" runs with nothing set
set -u
tag=${1:-r05}; shift
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/$tag/chain_ab; mkdir -p $out
quick="--steps 3 --warmup 1 --cpu-rays 0 --variant-steps 0 --fit-steps 0 --train-steps 0 --parity-rays 0"
run() { if [ "$1" = base ]; then shift; "$@"; else a=$1; shift; env "$a" "$@"; fi; }
for rep in 1 2; do
  for arm in "$@"; do
    run $arm bash tools/clock_probe.sh $out/clocks_${arm}_$rep.txt python bench.py $quick > $out/bench_${arm}_$rep.json 2> /dev/null
  done
done
for arm in "$@"; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=$(echo $c | tr ' ' '+')
    rm -rf /tmp/pmcg_$n
    run $arm timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcg_$n -o pmc -- python tools/pmc_chain.py > /dev/null 2> $out/pmc_${arm}_$n.err
    f=$(find /tmp/pmcg_$n -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && grep -E "Kernel_Name|k_net_chain|k_layer<128" "$f" | cut -c1-420 > $out/pmc_${arm}_$n.csv
  done
done
python - "$out" "$@" <<'PY'
import csv, glob, json, re, sys, collections, statistics as st
out, arms = sys.argv[1], sys.argv[2:]
for g in arms:
    vals = []
    for f in sorted(glob.glob(f"{out}/bench_{g}_*.json")):
        t = open(f).read().strip()
        if t: vals.append(json.loads(t))
    pw, ck = [], []
    for f in glob.glob(f"{out}/clocks_{g}_*.txt"):
        for l in open(f):
            p = re.search(r"Power.*?:\s*([\d.]+)", l); c = re.search(r"sclk.*?\((\d+)Mhz\)", l)
            if p and c and float(p.group(1)) > 800: pw.append(float(p.group(1))); ck.append(int(c.group(1)))
    m = collections.defaultdict(list)
    for f in glob.glob(f"{out}/pmc_{g}_*.csv"):
        for r in csv.DictReader(open(f)): m[r["Counter_Name"]].append(float(r["Counter_Value"]))
    launches = len(m.get("FETCH_SIZE", [])) or 1
    s = {k: sum(v) for k, v in m.items()}
    line = {"arm": g, "rays_per_s": [v["value"] for v in vals], "roofline_frac": [v["roofline"]["frac"] for v in vals],
            "power_w_median": st.median(pw) if pw else None, "sclk_mhz_median": st.median(ck) if ck else None, "samples": len(pw)}
    # per pmc_chain.py run (4 forwards of the fine network): totals, so that per-layer and chained forms compare like for like
    if "FETCH_SIZE" in s: line["fetch_gb_per_forward_x2"] = round(s["FETCH_SIZE"] * 2048 / 1e9 / 4, 1)
    if "TCC_HIT_sum" in s: line["l2_hit_rate"] = round(s["TCC_HIT_sum"] / (s["TCC_HIT_sum"] + s["TCC_MISS_sum"]), 4)
    if "GRBM_GUI_ACTIVE" in s: line["mfma_busy"] = round(s["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * s["GRBM_GUI_ACTIVE"] / 8), 4)
    if pw and vals: line["joule_per_frame"] = round(st.median(pw) * 262144 / (sum(v["value"] for v in vals) / len(vals)))
    print(json.dumps(line))
PY
