set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sorted_blocks.py tests/test_gpu_plans.py tests/test_gpu_xskip.py -m gpu -x -q > gpurun_out/r03_pytest4.log 2>&1
timeout -k 10 600 python tools/explore.py "$(cat tools/exp/r03_auto_sweep.json)" > gpurun_out/r03_auto_sweep.jsonl 2> gpurun_out/r03_auto_sweep.err
