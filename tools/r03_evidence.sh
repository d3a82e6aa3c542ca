#!/bin/bash
# tools/r03_evidence.sh -- the round-3 measurements DESIGN.md section 6 quotes, as the commands that produced them (run on the GPU
# box through gpurun, in a few calls; outputs land in gpurun_out/, the files DESIGN.md cites are copied into profiles/).
set -e
mkdir -p gpurun_out
PART=${1:-all}
if [ $PART = all ] || [ $PART = tests ]; then
  python -m pytest tests -m gpu -x -q
  SPMV_SORTED_FROM=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_structures.py tests/test_gpu_extras.py -m gpu -x -q
fi
if [ $PART = all ] || [ $PART = bench ]; then
  python bench.py > gpurun_out/r03_bench_full.json                               # -> profiles/r03_bench_n1_full.json
fi
if [ $PART = all ] || [ $PART = prof1 ]; then
  bash tools/profile.sh r03                                                      # headline: kernel stats + FETCH/WRITE passes
  bash tools/profile.sh r03_band65536 "--band 65536"
  bash tools/profile.sh r03_band200000 "--band 200000"
fi
if [ $PART = all ] || [ $PART = prof2 ]; then
  bash tools/profile.sh r03_band1M "--band 1000000"
  bash tools/profile.sh r03_c3 "--config c3 --band 8192"
  bash tools/profile.sh r03_c2_uniform "--config c2 --band 0"
fi
if [ $PART = all ] || [ $PART = prof3 ]; then
  bash tools/profile.sh r03_c5shard "--config c5shard --band 8192"
  bash tools/profile.sh r03_c5shard_uniform "--config c5shard --band 0"
  export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_stencil/trace -- python3 tools/stencil_time.py 200 > gpurun_out/prof_r03_stencil/stencil.jsonl
fi
# (then, anywhere: for t in r03 r03_band65536 r03_band200000 r03_band1M r03_c3 r03_c2_uniform r03_c5shard r03_c5shard_uniform; do python tools/summarize_profile.py $t; done)
