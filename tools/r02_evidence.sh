#!/bin/bash
# tools/r02_evidence.sh -- every measurement DESIGN.md section 6 "Round 2" quotes, as the commands that produced it.
# Run on the GPU box (through gpurun, in a few calls: the whole list is ~6 GPU-minutes); outputs land in gpurun_out/,
# the files DESIGN.md cites were copied from there into profiles/ (profiles/README.md says which).
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q                                              # parity: the first gate
# the same parity files with every chunk sorted / every panel plan in LDS mode / round 1's timed plans
SPMV_SORTED_FROM=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_structures.py tests/test_gpu_extras.py tests/test_file_io.py -m gpu -x -q
SPMV_PANEL_LDS=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_structures.py tests/test_gpu_extras.py -m gpu -x -q
SPMV_AUTOTUNE=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_structures.py -m gpu -x -q
python bench.py > gpurun_out/r02_bench_full.json                                 # -> profiles/r02_bench_n1_full.json
bash tools/profile.sh r02                                                        # headline: kernel stats + FETCH/WRITE passes
bash tools/profile.sh r02_band65536 "--band 65536"
bash tools/profile.sh r02_c3 "--config c3 --band 8192"
bash tools/profile.sh r02_c2_uniform "--config c2 --band 0"
bash tools/profile.sh r02_c4_uniform "--band 0"
# (then, anywhere: for t in r02 r02_band65536 r02_c3 r02_c2_uniform r02_c4_uniform; do python tools/summarize_profile.py $t; done)
python tools/explore.py "$(cat tools/exp/r02_calibrate.json)" > gpurun_out/r02_calibrate.jsonl   # the plan's prices
python tools/explore.py "$(cat tools/exp/r02_sorted2.json)" > gpurun_out/r02_sorted2.jsonl       # sorted vs staged
python tools/explore.py "$(cat tools/exp/r02_panel_lds.json)" > gpurun_out/r02_panel_lds.jsonl   # the LDS-panel experiment
python tools/bitmap_time.py > gpurun_out/r02_bitmap_time.jsonl                                   # bitmap formats, XSKIP
python tools/vendor_compare.py --out gpurun_out/r02_vendor.jsonl                                 # rocSPARSE beside it
tools/bin/ubench_gather > gpurun_out/r02_gather_ubench.jsonl                                     # (hipcc lines: file headers)
tools/bin/ubench_lds_atomic > gpurun_out/r02_lds_atomic.jsonl
python bench.py --scaling strong --steps 5 --warmup 2 > gpurun_out/r02_bench_strong1.json       # config 5 on one GPU
spmv-test_amd/bin/spmv_dist_selftest --rows-per-rank 4194304 --band 8192 > gpurun_out/r02_dist_selftest.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --rows-per-gpu 2097152 > gpurun_out/r02_bench_gloo2.json
