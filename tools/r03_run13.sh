#!/bin/bash
# round 3, final code: the whole GPU suite, smoke, then kernel statistics + counters of the two BASELINE-named kernels again
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gpu_tests_final.log 2>&1
tail -3 gpurun_out/r03_gpu_tests_final.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke_final.log 2>&1
tail -1 gpurun_out/r03_smoke_final.log
bash tools/profile.sh r03_c3_wave_pipe "--config c3 --band 8192 --variant wave_pipe" > gpurun_out/prof_wp.log 2>&1
bash tools/profile.sh r03_c2_scalar "--config c2 --band 8192 --variant scalar" > gpurun_out/prof_sc.log 2>&1
python bench.py --vendor wait > gpurun_out/r03_bench_full3.json 2> gpurun_out/r03_bench_full3.err
tail -1 gpurun_out/r03_bench_full3.err
