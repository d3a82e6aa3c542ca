#!/bin/bash
# tools/r04_evidence.sh -- the rocprofv3 evidence of round 4 (run on the GPU box through gpurun, summaries made afterwards here by
# tools/summarize_profile.py <tag>):  headline; config 4 / config 3 with uniform columns (SPMV_AUTO -> the binned layout); config 3
# band 8192 through SPMV_WAVE (the rebuilt short-row path).
set -e
mkdir -p gpurun_out
tools/profile.sh r04 > gpurun_out/prof_r04.log 2>&1
tools/profile.sh r04_c4_uniform "--band 0" > gpurun_out/prof_r04_c4u.log 2>&1
tools/profile.sh r04_c3_uniform "--config c3 --band 0" > gpurun_out/prof_r04_c3u.log 2>&1
tools/profile.sh r04_c3_wave "--config c3 --band 8192 --variant wave" > gpurun_out/prof_r04_c3w.log 2>&1
echo evidence done
