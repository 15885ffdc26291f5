#!/bin/bash
# rocprofv3 --pmc passes over the chained wide-network kernel (tools/pmc_chain.py), one pass per counter group (no tracing next to --pmc),
# summarised per kernel into gpurun_out/$tag/pmc_chain_<group>.csv;  bash tools/gpu_profile_chain.sh r04
set -u
tag=${1:-rXX}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/$tag
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  n=$(echo $c | tr ' ' '+')
  rm -rf /tmp/pmcc_$n
  timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcc_$n -o pmc -- python tools/pmc_chain.py > /dev/null 2> gpurun_out/$tag/pmc_chain_$n.err
  f=$(find /tmp/pmcc_$n -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Kernel_Name|k_net_chain" "$f" | cut -c1-420 > gpurun_out/$tag/pmc_chain_$n.csv
done
ls -la gpurun_out/$tag | head -20
