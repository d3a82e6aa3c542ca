set -e
mkdir -p gpurun_out
: > gpurun_out/r03_stencil.jsonl
for blk in 0 256 512 1024; do
  if [ $blk = 0 ]; then
    echo '{"env": "default"}' >> gpurun_out/r03_stencil.jsonl
    python tools/stencil_time.py 200 256 >> gpurun_out/r03_stencil.jsonl 2>> gpurun_out/r03_stencil.err
    echo '{"env": "SPMV_AUTOTUNE=1"}' >> gpurun_out/r03_stencil.jsonl
    SPMV_AUTOTUNE=1 python tools/stencil_time.py 200 256 >> gpurun_out/r03_stencil.jsonl 2>> gpurun_out/r03_stencil.err
  else
    echo "{\"env\": \"SPMV_TILED_BLOCK=$blk\"}" >> gpurun_out/r03_stencil.jsonl
    SPMV_TILED_BLOCK=$blk python tools/stencil_time.py 200 256 >> gpurun_out/r03_stencil.jsonl 2>> gpurun_out/r03_stencil.err
  fi
done
python bench.py > gpurun_out/r03_bench1.json 2> gpurun_out/r03_bench1.err
