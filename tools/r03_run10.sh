set -e
mkdir -p gpurun_out
timeout -k 10 300 python tools/explore.py "$(cat tools/exp/r03_scalar.json)" > gpurun_out/r03_scalar.jsonl 2> gpurun_out/r03_scalar.err
bash tools/xskip_scope.sh 262144 16 > gpurun_out/r03_xskip_scope.log 2>&1
