#!/bin/bash
# tools/sq_counters.sh TAG 'EXPLORE_JSON' KERNEL_SUBSTRING -- SQ counters of one kernel under tools/explore.py (GPU box, through
# gpurun): three rocprofv3 --pmc passes (waves / waiting, instruction counts, busy units), then per-launch means as JSON lines
# in gpurun_out/sq_TAG.jsonl.  Counters only (no trace domains).
set -e
TAG=$1; EXP=$2; KER=$3
OUT=gpurun_out/sq_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM"
P3="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 tools/explore.py "$EXP" > $OUT/p$i.out 2> $OUT/p$i.err
done
python3 - "$OUT" "$KER" > gpurun_out/sq_$TAG.jsonl <<'PY'
import csv, glob, json, sys, collections
out, ker = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ker in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(json.dumps({"kernel": k, "launches": max(len(v) for v in d.values()), **{c: round(sum(v) / len(v), 1) for c, v in sorted(d.items())}}))
PY
cat gpurun_out/sq_$TAG.jsonl
