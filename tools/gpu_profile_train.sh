#!/bin/bash
# Diagnostic passes over ONE training network pass (tools/pmc_train.py: fine 1024 x 10, 1,536 rays x 128), chained (MOFA_CHAIN_TRAIN=1: k_net_chain_train) and per
# layer (the default): a kernel trace with the per-dispatch durations of the backward kernels, then rocprofv3 --pmc passes (one counter group
# per pass, no tracing next to --pmc).   bash tools/gpu_profile_train.sh r06
set -u
tag=${1:-rXX}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/$tag
for c in 1 0; do
  rm -rf /tmp/trt_$c
  MOFA_CHAIN_TRAIN=$c timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/trt_$c -o tr -- python tools/pmc_train.py > /dev/null 2> gpurun_out/$tag/train_trace_chain$c.err
  db=$(find /tmp/trt_$c -name '*.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/$tag/kernel_stats_trainpass_chain$c.md "MOFA_CHAIN_TRAIN=$c rocprofv3 --kernel-trace --stats -- python tools/pmc_train.py" > /dev/null
  [ -n "$db" ] && python - "$db" > gpurun_out/$tag/train_dispatches_chain$c.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
print(cols)
rows = cur.execute("select name, start, end, grid_x, workgroup_x from kernels where name like '%k_net_chain%' or name like '%k_wgrad<%' or name like '%k_layer<128, false, true%' order by start").fetchall()
import collections
agg = collections.defaultdict(list)
for n, s, e, g, w in rows:
    agg[n.split('(')[0][-60:]].append((e - s) / 1e3)
for n, v in agg.items():
    print(n, len(v), 'total_ms', round(sum(v) / 1e3, 3), 'us each (first 12):', [round(x, 1) for x in v[:12]])
PY
done
for c in 1 0; do
  for g in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $g | tr ' ' '+')
    rm -rf /tmp/pmct_$n
    MOFA_CHAIN_TRAIN=$c PMC_TRAIN_STEPS=2 timeout 300 rocprofv3 --pmc $g --output-format csv -d /tmp/pmct_$n -o pmc -- python tools/pmc_train.py > /dev/null 2> gpurun_out/$tag/pmc_train_chain${c}_$n.err
    f=$(find /tmp/pmct_$n -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && grep -E "Kernel_Name|k_net_chain|k_wgrad<|k_layer<128, false, true" "$f" | cut -c1-520 > gpurun_out/$tag/pmc_train_chain${c}_$n.csv
  done
done
ls gpurun_out/$tag | head -50
