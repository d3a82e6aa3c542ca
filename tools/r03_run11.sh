set -e
mkdir -p gpurun_out gpurun_out/prof_r03_stencil
( time python bench.py > gpurun_out/r03_bench5.json 2> gpurun_out/r03_bench5.err ) 2> gpurun_out/r03_bench5.time
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_pytest7.log 2>&1
bash tools/r03_evidence.sh prof3 > gpurun_out/r03_evidence_prof3.log 2>&1
( time python bench.py --vendor wait > gpurun_out/r03_bench_full.json 2> gpurun_out/r03_bench_full.err ) 2> gpurun_out/r03_bench_full.time
