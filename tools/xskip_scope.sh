#!/bin/bash
# tools/xskip_scope.sh -- round 3, VERDICT item 7: SPMV_XSKIP at config 2's DENSITY (16 nonzeros per row, uniform columns)
# on the largest matrix its (output blocks x inputs) table admits, with 0 / 50 / 90 % zeros in x: times (plain runs) and
# FETCH_SIZE (rocprofv3 --pmc, separate runs) of k_xskip against the row-major CSR kernel.  -> gpurun_out/r03_xskip_scope.jsonl
set -e
export TMPDIR=/tmp
OUT=gpurun_out/xskip_scope
rm -rf $OUT; mkdir -p $OUT
ROWS=${1:-262144}; PER=${2:-16}
for z in 0.0 0.5 0.9; do
  python3 tools/xskip_traffic.py $z $ROWS $PER > $OUT/time_$z.json 2> $OUT/time_$z.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$z -- python3 tools/xskip_traffic.py $z $ROWS $PER > $OUT/prof_$z.json 2> $OUT/prof_$z.err
done
python3 - <<'PY'
import csv, glob, json
from collections import defaultdict
rows = []
for z in ("0.0", "0.5", "0.9"):
    t = json.loads(open(f"gpurun_out/xskip_scope/time_{z}.json").read().strip().splitlines()[-1])
    acc = defaultdict(list)
    for f in glob.glob(f"gpurun_out/xskip_scope/fetch_{z}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for name in ("k_xskip", "k_xs_combine", "k_adaptive"):
                if name + "(" in k or name + "<" in k:
                    acc[name].append(float(r["Counter_Value"]))
    t["FETCH_SIZE_KiB_mean_per_launch"] = {k: round(sum(v) / len(v), 1) for k, v in acc.items()}
    t["FETCH_bytes_per_nonzero(2x: gfx950 half-count)"] = {k: round(2 * 1024 * sum(v) / len(v) / t["nnz"], 2) for k, v in acc.items()}
    rows.append(t)
open("gpurun_out/r03_xskip_scope.jsonl", "w").write("\n".join(json.dumps(r) for r in rows) + "\n")
print("\n".join(json.dumps(r) for r in rows))
PY
