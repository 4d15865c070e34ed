#!/bin/bash
# Counters of the general kernels on uniform medium reads (default: 600 1000 2000 bp), one kernel per pass:
# the matrix-pipe wave-per-read kernel (sk_band.hip, what the library selects) and round 2's teams of 16 lanes
# (SK_GENERAL=team).  Two --pmc passes per case (instruction mix; where the wave-cycles go), never combined
# with other trace domains.  Usage (via gpurun): bash tools/probes/medium_pmc.sh "600 1000 2000" > out.log
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LENS=${1:-"600 1000 2000"}
cd /tmp && export TMPDIR=/tmp
for L in $LENS; do
  for K in ${KERNELS:-band team}; do
    OUT=$ROOT/gpurun_out/medium_pmc_${K}_$L
    rm -rf $OUT; mkdir -p $OUT
    i=0
    for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM" \
               "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA"; do
      i=$((i+1))
      SK_GENERAL=$K rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/probes/stream_case.py $L > $OUT/run$i.log 2> $OUT/err$i.log || { echo "$K $L pass $i failed"; tail -3 $OUT/err$i.log; }
    done
    python3 - $OUT $K $L <<'PY'
import csv, glob, sys, collections
out, kern, L = sys.argv[1], sys.argv[2], int(sys.argv[3])
agg = collections.defaultdict(list)
name = ""
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sk_scan_" in r["Kernel_Name"] and ("band" in r["Kernel_Name"] or "team" in r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            name = r["Kernel_Name"].split("(")[0]
reads = 1_000_000_000 // L
print("== %s, uniform %d bp, %d reads per launch: %s" % (kern, L, reads, name))
for k, v in sorted(agg.items()):
    print("   %-22s %14.0f per launch  %10.1f per read  (%d launches)" % (k, sum(v) / len(v), sum(v) / len(v) / reads, len(v)))
PY
  done
done
