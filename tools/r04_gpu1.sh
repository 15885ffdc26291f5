#!/bin/bash
# round-4 GPU pass 1: whole GPU suite, the headline line (with the fine-256x8 series), fitting with the mask / fp32 tape
python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -40 > gpurun_out/r04_gputests.log
tail -5 gpurun_out/r04_gputests.log
python bench.py > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err; tail -c 600 gpurun_out/r04_bench_n1.json; tail -3 gpurun_out/r04_bench_n1.err
python bench.py --mode fit --cpu-rays 0 > gpurun_out/r04_bench_fit.json 2>/dev/null; tail -c 700 gpurun_out/r04_bench_fit.json
python bench.py --mode fit --cpu-rays 0 --tape fp32 > gpurun_out/r04_bench_fit_variant_tape_fp32.json 2>/dev/null; tail -c 700 gpurun_out/r04_bench_fit_variant_tape_fp32.json
