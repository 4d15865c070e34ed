#!/bin/bash
# Kernel-trace durations (last 40 launches of each kernel, after the clocks have settled) of tools/probes/mix_case.py:
# the mixed-length batch segmented / ragged (regrouped on the device), uniform 150 bp beside them.  Any SK_* switch in
# the environment applies.  Usage: [SEQ=1] [SK_SEG_STAGE=0] [SK_SEG_CHUNK_SHIFT=0] bash tools/probes/mix_trace.sh [reads]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-4000000}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/mix_trace_$$
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/probes/mix_case.py $N 40 120 > $OUT/run.log 2> $OUT/err.log || { echo "run failed"; tail -5 $OUT/err.log; tail -3 $OUT/run.log; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
info = dict(zip(*[iter(open(out + "/run.log").read().split())] * 2)) if open(out + "/run.log").read().strip() else {}
algo = float(info.get("algorithmic_bytes", 0))
rows = []
for f in glob.glob(out + "/t/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a "scan" = the sk_ kernels between two uniform launches; group by the case they belong to
d = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if "sk_" in k:
        d[k[-56:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v = v[-40:]
    us = sum(v) / len(v) / 1e3
    print("%-58s %8.1f us (last %d)" % (k, us, len(v)) + ("   %6.0f GB/s if it were the whole scan" % (algo / us / 1e3) if algo and us > 20 else ""), flush=True)
PY
rm -rf $OUT
