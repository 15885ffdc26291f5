#!/bin/bash
# rocprofv3 --pmc passes over ONE fitting network pass (tools/pmc_fit.py: k_net_chain<1> forward + mask tape, k_net_chain<2> backward), one counter
# group per pass, summarised by tools/pmc_summary.py.   bash tools/gpu_profile_fit.sh r06
set -u
tag=${1:-rXX}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/$tag
for g in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAVE_CYCLES" FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $g | tr ' ' '+')
  rm -rf /tmp/pmcf_$n
  timeout 300 rocprofv3 --pmc $g --output-format csv -d /tmp/pmcf_$n -o pmc -- python tools/pmc_fit.py > /dev/null 2> gpurun_out/$tag/pmc_fit_$n.err
  f=$(find /tmp/pmcf_$n -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Kernel_Name|k_net_chain|k_colsum|k_mask_pack" "$f" | cut -c1-520 > gpurun_out/$tag/pmc_fit_$n.csv
done
python tools/pmc_summary.py gpurun_out/$tag pmc_fit_ gpurun_out/$tag/pmc_fit_pass.md "tools/pmc_fit.py: the shipped fine network 1024 x 10, 1,024 rays x 128 (512 row tiles), three fitting passes: k_net_chain<1> = forward writing the mask tape, k_net_chain<2> = the backward's two chained launches (view layer + texture stack | shape stack + xyzEncode 3..1)."
