#!/bin/bash
# tools/profile.sh -- rocprofv3 evidence for bench.py (run on the GPU box through gpurun).
#   pass 1: --kernel-trace --stats           -> per-kernel average duration
#   pass 2: --pmc FETCH_SIZE                 -> L2<->fabric read requests (TCC has 4 slots, FETCH_SIZE uses 3)
#   pass 3: --pmc WRITE_SIZE                 -> write side (separate pass, MI355X_MICROARCH.md "PMC slots")
#   pass 4/5: the same two counters on tools/bin/ubench_stream (a pure 16-B/lane stream of known
#             size: the calibration the guide asks for before quoting FETCH_SIZE absolutes)
#   usage: tools/profile.sh <tag> ["extra bench.py args"]
# Never combined with --sys-trace/--hip-trace (gpurun refuses that).  Output: gpurun_out/prof_<tag>/
set -e
TAG=${1:-r01}
EXTRA=${2:-}          # extra bench.py arguments, e.g. "--variant panel --band 0"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --no-extras --no-cpu-baseline --traffic off $EXTRA"
# pass 1 keeps the CPU baseline when no extra arguments are given, so that the committed bench line
# (profiles/<tag>_bench_n1.json) and the kernel statistics come from ONE process (one plan, one device); the other
# workloads stay out of it -- their launches of the same kernels would blur the per-kernel averages
if [ -z "$EXTRA" ]; then TRACE_ARGS="--steps 50 --warmup 5 --no-extras --traffic off"; else TRACE_ARGS="$ARGS"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $TRACE_ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- tools/bin/ubench_stream 256 > $OUT/cal_fetch.txt 2> $OUT/cal_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- tools/bin/ubench_stream 256 > $OUT/cal_write.txt 2> $OUT/cal_write.err
find $OUT -name "*.csv" | head -40
