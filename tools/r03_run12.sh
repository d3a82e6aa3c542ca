#!/bin/bash
# round 3, after the WAVE_PIPE / SCALAR rebuild: the full bench line with the vendor library, and kernel statistics +
# counters of the two BASELINE-named kernels on the configs BASELINE names them for
set -e
mkdir -p gpurun_out
python bench.py --vendor wait > gpurun_out/r03_bench_full2.json 2> gpurun_out/r03_bench_full2.err
bash tools/profile.sh r03_c3_wave_pipe "--config c3 --band 8192 --variant wave_pipe" > gpurun_out/prof_wp.log 2>&1
bash tools/profile.sh r03_c2_scalar "--config c2 --band 8192 --variant scalar" > gpurun_out/prof_sc.log 2>&1
for t in r03_c3_wave_pipe r03_c2_scalar; do python tools/summarize_profile.py $t > gpurun_out/summ_$t.log 2>&1 || true; done
tail -2 gpurun_out/summ_r03_c3_wave_pipe.log
