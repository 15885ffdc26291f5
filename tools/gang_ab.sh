#!/bin/bash
# A/B of k_net_chain's gang re-alignment option (MOFA_CHAIN_GANG=n polls; 0 = off): frame rate + live roofline, socket power / clock,
# and the two PMC passes that say whether the sharers of a row tile's panels found them in the L2 (FETCH_SIZE, TCC hit / miss).
#   bash tools/gang_ab.sh <tag> [polls...]
set -u
tag=${1:-r05}; shift
polls=${@:-"0 64"}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=gpurun_out/$tag/gang; mkdir -p $out
quick="--steps 3 --warmup 1 --cpu-rays 0 --variant-steps 0 --fit-steps 0 --train-steps 0 --parity-rays 0"
for rep in 1 2; do
  for g in $polls; do
    MOFA_CHAIN_GANG=$g bash tools/clock_probe.sh $out/clocks_gang${g}_$rep.txt python bench.py $quick > $out/bench_gang${g}_$rep.json 2> /dev/null
  done
done
for g in $polls; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=$(echo $c | tr ' ' '+')
    rm -rf /tmp/pmcg_$n
    MOFA_CHAIN_GANG=$g timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcg_$n -o pmc -- python tools/pmc_chain.py > /dev/null 2> $out/pmc_gang${g}_$n.err
    f=$(find /tmp/pmcg_$n -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && grep -E "Kernel_Name|k_net_chain" "$f" | cut -c1-420 > $out/pmc_gang${g}_$n.csv
  done
done
python - <<PY
import csv, glob, json, re, collections, statistics as st
out = "$out"
for g in "$polls".split():
    vals = [json.load(open(f)) for f in sorted(glob.glob(f"{out}/bench_gang{g}_*.json")) if open(f).read().strip()]
    pw, ck = [], []
    for f in glob.glob(f"{out}/clocks_gang{g}_*.txt"):
        for l in open(f):
            p = re.search(r"Power.*?:\s*([\d.]+)", l); c = re.search(r"sclk.*?\((\d+)Mhz\)", l)
            if p and c and float(p.group(1)) > 800: pw.append(float(p.group(1))); ck.append(int(c.group(1)))
    m = collections.defaultdict(list)
    for f in glob.glob(f"{out}/pmc_gang{g}_*.csv"):
        for r in csv.DictReader(open(f)): m[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in m.items()}
    line = {"gang_polls": int(g), "rays_per_s": [v["value"] for v in vals], "roofline_frac": [v["roofline"]["frac"] for v in vals],
            "power_w_median": st.median(pw) if pw else None, "sclk_mhz_median": st.median(ck) if ck else None, "samples": len(pw)}
    if "FETCH_SIZE" in m: line["fetch_gb_per_launch_x2"] = round(m["FETCH_SIZE"] * 2048 / 1e9, 1)
    if "TCC_HIT_sum" in m: line["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
    if "GRBM_GUI_ACTIVE" in m: line["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["GRBM_GUI_ACTIVE"] / 8), 4)
    print(json.dumps(line))
PY
