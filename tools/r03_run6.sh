set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest5.log 2>&1
timeout -k 10 600 python tools/explore.py "$(cat tools/exp/r03_auto_sweep.json)" > gpurun_out/r03_auto_sweep2.jsonl 2> gpurun_out/r03_auto_sweep2.err
