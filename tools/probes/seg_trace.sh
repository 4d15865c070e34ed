#!/bin/bash
# Kernel-trace durations (not event brackets: those differ by several per cent between two buffers and with the order
# of the cases) of the uniform kernel and the segmented kernels ON THE SAME BUFFER, one length per run:
# uniform staged / uniform LDS-DMA / segmented staged / segmented LDS-DMA.  Usage: bash tools/probes/seg_trace.sh "150 250"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LENS=${1:-"75 150 250 301"}
cd /tmp && export TMPDIR=/tmp
for L in $LENS; do
  for MODE in "1 1" "0 0"; do
    set -- $MODE
    OUT=$ROOT/gpurun_out/seg_trace_${L}_$1
    rm -rf $OUT; mkdir -p $OUT
    SK_TILE_STAGE=$1 SK_SEG_STAGE=$2 timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/probes/seg_case.py $L 40 120 > $OUT/run.log 2> $OUT/err.log || { echo "$L trace failed"; tail -3 $OUT/err.log; }
    python3 - $OUT $L <<'PY'
import csv, glob, sys
out, L = sys.argv[1], int(sys.argv[2])
st = ((L + 7) // 8 | 1) * 8
n = 1_000_000_000 // st // 64 * 64
import collections
d = collections.defaultdict(list)
for f in glob.glob(out + "/t/*/*kernel_trace.csv"):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        if "sk_scan_" in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(")[0][-52:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v = v[-40:]
    us = sum(v) / len(v) / 1e3
    print("%4d bp  %-52s %7.1f us  %6.0f GB/s raw  %6.0f GB/s algorithmic  (last %d launches)" % (L, k, us, n * st / us / 1e3, n * (L + 8) / us / 1e3, len(v)), flush=True)
PY
    rm -rf $OUT
  done
done
