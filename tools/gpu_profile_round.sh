#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun): rocprofv3 kernel traces of the three bench modes, summarised to
# markdown by tools/rocpd_stats.py (the 25 MB databases stay in /tmp), plus the two PMC passes behind roofline.traffic.
#   bash tools/gpu_profile_round.sh r02
set -u
tag=${1:-rXX}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/$tag
for mode in render fit train; do
  extra="--steps 1 --warmup 1 --cpu-rays 0"
  [ $mode = fit ] && extra="--steps 5 --warmup 2 --cpu-rays 0"
  [ $mode = train ] && extra="--steps 2 --warmup 4 --cpu-rays 0"
  rm -rf /tmp/prof_$mode
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o bench -- python bench.py --mode $mode $extra > gpurun_out/$tag/bench_${mode}_under_rocprof.json 2> gpurun_out/$tag/bench_${mode}_under_rocprof.err
  db=$(find /tmp/prof_$mode -name '*.db' | head -1)
  python tools/rocpd_stats.py "$db" gpurun_out/$tag/kernel_stats_${mode}.md "rocprofv3 --kernel-trace --stats -- python bench.py --mode $mode $extra"
done
# one rocprofv3 --pmc pass per counter group (no tracing options next to --pmc)
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '+')
  rm -rf /tmp/pmc_$n
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$n -o pmc -- python tools/pmc_layer.py > /dev/null 2> gpurun_out/$tag/pmc_$n.err
  f=$(find /tmp/pmc_$n -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && grep -E "Kernel_Name|k_layer" "$f" | cut -c1-400 | head -80 > gpurun_out/$tag/pmc_$n.csv
done
python tools/make_traffic_json.py gpurun_out/$tag gpurun_out/$tag/hbm_traffic.json > /dev/null 2> gpurun_out/$tag/hbm_traffic.err
python tools/kernel_resources.py mofanerf_amd/libmofanerf_hip.so gpurun_out/$tag/kernel_resources.md
# the same counters for the persistent <= 256-wide network kernel (the north star's kernel) and its generic twin
bash tools/gpu_profile_fused.sh $tag pipelined generic > /dev/null 2>&1
python tools/make_traffic_fused_json.py gpurun_out/$tag gpurun_out/$tag/hbm_traffic_fused.json > /dev/null 2> gpurun_out/$tag/hbm_traffic_fused.err
# the RCCL branches on this one GPU (one-rank nccl group, every collective issued): kernel trace of what actually ran
rm -rf /tmp/prof_rccl
rocprofv3 --kernel-trace --stats -d /tmp/prof_rccl -o rccl -- python tools/rccl_world1.py > gpurun_out/$tag/rccl_world1.json 2> gpurun_out/$tag/rccl_world1.err
db=$(find /tmp/prof_rccl -name '*.db' | head -1)
[ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/$tag/kernel_stats_rccl_world1.md "rocprofv3 --kernel-trace --stats -- python tools/rccl_world1.py"
# the round's plain bench lines (no profiler attached), with the PMC traffic of THIS build next to the live roofline figure
cp gpurun_out/$tag/hbm_traffic.json profiles/hbm_traffic.json
python bench.py > gpurun_out/$tag/bench_n1.json 2> gpurun_out/$tag/bench_n1.err
python bench.py --mode fit > gpurun_out/$tag/bench_fit_n1.json 2> gpurun_out/$tag/bench_fit_n1.err
python bench.py --mode train > gpurun_out/$tag/bench_train_n1.json 2> gpurun_out/$tag/bench_train_n1.err
python bench.py --arch 8 256 8 256 --cpu-rays 0 > gpurun_out/$tag/bench_n1_variant_fine256x8.json 2> /dev/null
ls -la gpurun_out/$tag
