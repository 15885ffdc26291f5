#!/bin/bash
# rocprofv3 --pmc passes over the persistent network kernels (tools/pmc_fused.py), one pass per counter group (no tracing next to --pmc),
# summarised per kernel into gpurun_out/$tag/pmc_fused_<group>.csv;  bash tools/gpu_profile_fused.sh r04 [arms...]
set -u
tag=${1:-rXX}; shift
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/$tag
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '+')
  rm -rf /tmp/pmcf_$n
  timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcf_$n -o pmc -- python tools/pmc_fused.py "$@" > /dev/null 2> gpurun_out/$tag/pmc_fused_$n.err
  f=$(find /tmp/pmcf_$n -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Kernel_Name|k_mlp_" "$f" | cut -c1-420 > gpurun_out/$tag/pmc_fused_$n.csv
done
ls -la gpurun_out/$tag | head -40
