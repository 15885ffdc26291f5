#!/bin/bash
# Round-5 profile recipe (run on the GPU box through gpurun):  bash tools/gpu_profile_round5.sh r05
#   1. rocprofv3 --kernel-trace --stats of the three bench modes, summarised to markdown (tools/rocpd_stats.py)
#   2. rocprofv3 --pmc passes over the chained kernel (separate passes, no tracing next to --pmc) -> hbm_traffic_chain.json
#   3. engine clock + socket power under the benchmark with the chained launch and with per-layer launches
set -u
tag=${1:-r05}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/$tag
for mode in render fit train; do
  extra="--steps 1 --warmup 1 --cpu-rays 0 --variant-steps 0 --fit-steps 0 --train-steps 0 --parity-rays 0"
  [ $mode = fit ] && extra="--steps 5 --warmup 2 --cpu-rays 0"
  [ $mode = train ] && extra="--steps 2 --warmup 4 --cpu-rays 0"
  rm -rf /tmp/prof_$mode
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o bench -- python bench.py --mode $mode $extra > gpurun_out/$tag/bench_${mode}_under_rocprof.json 2> gpurun_out/$tag/bench_${mode}_under_rocprof.err
  db=$(find /tmp/prof_$mode -name '*.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/$tag/kernel_stats_${mode}.md "rocprofv3 --kernel-trace --stats -- python bench.py --mode $mode $extra"
done
bash tools/gpu_profile_chain.sh $tag > /dev/null 2>&1
python tools/make_traffic_chain_json.py gpurun_out/$tag gpurun_out/$tag/hbm_traffic_chain.json > /dev/null 2> gpurun_out/$tag/hbm_traffic_chain.err
python tools/kernel_resources.py mofanerf_amd/libmofanerf_hip.so gpurun_out/$tag/kernel_resources.md > /dev/null
quick="--steps 3 --warmup 1 --cpu-rays 0 --variant-steps 0 --fit-steps 0 --train-steps 0 --parity-rays 0"
[ "${SKIP_CLOCKS:-0}" = 1 ] || for c in 1 0 1 0; do
  MOFA_CHAIN=$c bash tools/clock_probe.sh gpurun_out/$tag/clocks_chain${c}_$RANDOM.txt python bench.py $quick > gpurun_out/$tag/bench_clock_chain${c}_$RANDOM.json 2> /dev/null
done
ls -la gpurun_out/$tag | head -40
