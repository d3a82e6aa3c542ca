set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest0.log 2>&1
tools/bin/ubench_gather_lines > gpurun_out/r03_gather_lines.jsonl 2> gpurun_out/r03_gather_lines.err
python tools/explore.py "$(cat tools/exp/r03_wide.json)" > gpurun_out/r03_wide.jsonl 2> gpurun_out/r03_wide.err
python bench.py > gpurun_out/r03_bench0.json 2> gpurun_out/r03_bench0.err
