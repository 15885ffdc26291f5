#!/bin/bash
# Round-6 profile recipe (run on the GPU box through gpurun):  bash tools/gpu_profile_round6.sh r06
#   1. pytest -m gpu, smoke()
#   2. rocprofv3 --kernel-trace --stats of the three bench modes, summarised to markdown (tools/rocpd_stats.py)
#   3. rocprofv3 --pmc passes (separate passes, no tracing next to --pmc): the chained kernel -> hbm_traffic_chain.json, the HBM-side
#      ray kernels -> hbm_traffic_rays.json (both copied into profiles/ of THIS box copy so that the bench lines below quote them under
#      the source-digest guard)
#   4. the plain bench lines: default flags, --mode fit, --mode train (default per-layer backward and MOFA_CHAIN_TRAIN=1), the driver's flags
#   5. the chained launch's race hunt (tools/stress_chain.py)
set -u
tag=${1:-r06}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -q -rs 2>&1 | tail -25 > gpurun_out/$tag/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.txt 2>&1
for mode in render fit train; do
  extra="--steps 1 --warmup 1 --cpu-rays 0 --variant-steps 0 --fit-steps 0 --train-steps 0 --bulk-identities 0 --parity-rays 0"
  [ $mode = fit ] && extra="--steps 5 --warmup 2 --cpu-rays 0"
  [ $mode = train ] && extra="--steps 2 --warmup 4 --cpu-rays 0"
  rm -rf /tmp/prof_$mode
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o bench -- python bench.py --mode $mode $extra > gpurun_out/$tag/bench_${mode}_under_rocprof.json 2> gpurun_out/$tag/bench_${mode}_under_rocprof.err
  db=$(find /tmp/prof_$mode -name '*.db' | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" gpurun_out/$tag/kernel_stats_${mode}.md "rocprofv3 --kernel-trace --stats -- python bench.py --mode $mode $extra" > /dev/null
done
bash tools/gpu_profile_chain.sh $tag > /dev/null 2>&1
python tools/make_traffic_chain_json.py gpurun_out/$tag gpurun_out/$tag/hbm_traffic_chain.json > /dev/null 2> gpurun_out/$tag/hbm_traffic_chain.err
bash tools/gpu_profile_rays.sh $tag > /dev/null 2>&1
python tools/make_traffic_rays_json.py gpurun_out/$tag gpurun_out/$tag/hbm_traffic_rays.json > /dev/null 2> gpurun_out/$tag/hbm_traffic_rays.err
cp gpurun_out/$tag/hbm_traffic_chain.json gpurun_out/$tag/hbm_traffic_rays.json profiles/ 2>/dev/null
python tools/kernel_resources.py mofanerf_amd/libmofanerf_hip.so gpurun_out/$tag/kernel_resources.md > /dev/null
python bench.py > gpurun_out/$tag/bench_n1.json 2> gpurun_out/$tag/bench_n1.err
python bench.py --mode fit > gpurun_out/$tag/bench_fit_n1.json 2> gpurun_out/$tag/bench_fit_n1.err
python bench.py --mode train > gpurun_out/$tag/bench_train_n1.json 2> gpurun_out/$tag/bench_train_n1.err
MOFA_CHAIN_TRAIN=1 python bench.py --mode train > gpurun_out/$tag/bench_train_n1_chain_train.json 2> gpurun_out/$tag/bench_train_n1_chain_train.err
t0=$SECONDS
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/bench_n1_steps20_warmup5.json 2> gpurun_out/$tag/bench_n1_steps20_warmup5.err
echo "wall clock of the whole command: $((SECONDS - t0)) s" >> gpurun_out/$tag/bench_n1_steps20_warmup5.err
[ "${SKIP_STRESS:-0}" = 1 ] || timeout 1500 python tools/stress_chain.py ${STRESS_SCALE:-0.5} > gpurun_out/$tag/stress_chain.txt 2>&1
ls -la gpurun_out/$tag | head -60
