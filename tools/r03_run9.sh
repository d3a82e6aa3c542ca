set -e
mkdir -p gpurun_out
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err ) 2> gpurun_out/r03_bench2.time
timeout -k 10 300 python tools/explore.py "$(cat tools/exp/r03_sb_bounds.json)" > gpurun_out/r03_sb_bounds.jsonl 2> gpurun_out/r03_sb_bounds.err
spmv-test_amd/bin/spmv_dist_selftest --local --ranks 8 --pipeline 4 --exchange peer --rows-per-rank 2097152 --band 8192 > gpurun_out/r03_dist_selftest_local8.json 2> gpurun_out/r03_dist_selftest_local8.err
spmv-test_amd/bin/spmv_dist_selftest --pipeline 4 --exchange p2p --rows-per-rank 4194304 --band 8192 > gpurun_out/r03_dist_selftest_pipe_world1.json 2> gpurun_out/r03_dist_selftest_pipe_world1.err
python bench.py --backend native --scaling strong --total-blocks 8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench_native_strong1.json 2> gpurun_out/r03_bench_native_strong1.err
