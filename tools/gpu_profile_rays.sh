#!/bin/bash
# rocprofv3 --pmc passes over the HBM-bound ray kernels (tools/pmc_rays.py), one pass per counter group (no tracing next to --pmc),
# summarised per kernel into gpurun_out/$tag/pmc_rays_<group>.csv;  bash tools/gpu_profile_rays.sh r06
set -u
tag=${1:-rXX}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/$tag
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '+')
  rm -rf /tmp/pmcr_$n
  timeout 240 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcr_$n -o pmc -- python tools/pmc_rays.py > /dev/null 2> gpurun_out/$tag/pmc_rays_$n.err
  f=$(find /tmp/pmcr_$n -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Kernel_Name|k_composite|k_sample_pdf" "$f" | cut -c1-420 > gpurun_out/$tag/pmc_rays_$n.csv
done
ls -la gpurun_out/$tag | head -20
