#!/bin/bash
# instruction counts per launch of the streaming general kernel on uniform reads of length $1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
L=${1:-30000}
OUT=$ROOT/gpurun_out/stream_pmc_$L
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/probes/stream_case.py $L > $OUT/run.log 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "stream" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(open(sys.argv[1] + "/run.log").read().strip())
for k, v in sorted(agg.items()):
    print("%-20s %14.0f per launch (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
