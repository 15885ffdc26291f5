# A/B of two library builds on the 256x8 + 256x8 variant (separate processes, interleaved)
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_measure_arms.py tests/test_gpu_config1.py -m gpu -q -x --timeout=600 2>&1 | tail -3
for r in 1 2 3; do
 for L in base new; do
  if [ $L = base ]; then export MOFA_LIB=$PWD/build_arms/libmofanerf_base.so; else unset MOFA_LIB; fi
  python bench.py --arch 8 256 8 256 --cpu-rays 0 --cpu-tile-reps 0 --parity-rays 0 --steps 4 --warmup 1 --variant-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value'],1), d['roofline']['achieved'], d['roofline']['frac'])"
 done
done
