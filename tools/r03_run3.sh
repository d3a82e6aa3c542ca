set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sorted_blocks.py tests/test_gpu_xskip.py -m gpu -x -q > gpurun_out/r03_pytest2.log 2>&1
timeout -k 10 600 python tools/explore.py "$(cat tools/exp/r03_sorted_blocks2.json)" > gpurun_out/r03_sorted_blocks2.jsonl 2> gpurun_out/r03_sorted_blocks2.err
