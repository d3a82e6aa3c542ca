set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sorted_blocks.py -m gpu -x -q > gpurun_out/r03_pytest3.log 2>&1
timeout -k 10 600 python tools/explore.py "$(cat tools/exp/r03_sorted_blocks4.json)" > gpurun_out/r03_sorted_blocks4.jsonl 2> gpurun_out/r03_sorted_blocks4.err
