#!/bin/bash
# Counters of the kernel the library selects for uniform medium reads (tools/probes/stream_case.py L: reads of L bases
# back to back): instruction mix and where the wave-cycles go, two --pmc passes, never combined with other trace domains.
# Usage: bash tools/probes/wide_pmc.sh "600 1000 1500"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LENS=${1:-"600 1000"}
cd /tmp && export TMPDIR=/tmp
for L in $LENS; do
  OUT=$ROOT/gpurun_out/wide_pmc_$L
  rm -rf $OUT; mkdir -p $OUT
  i=0
  for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" \
             "SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT"; do
    i=$((i+1))
    timeout -k 10 180 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/probes/stream_case.py $L > $OUT/run$i.log 2> $OUT/err$i.log || { echo "$L pass $i failed"; tail -2 $OUT/err$i.log | cut -c1-200; }
  done
  python3 - $OUT $L <<'PY'
import csv, glob, sys, collections
out, L = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(list)
name = ""
dur = []
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sk_scan_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            name = r["Kernel_Name"].split("(")[0]
for f in glob.glob(out + "/p1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "sk_scan_" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
reads = 1_000_000_000 // L
print("== uniform %d bp, %d reads per launch: %s, %.1f us per launch under the counters" % (L, reads, name, sum(dur) / max(1, len(dur)) / 1e3))
for k, v in sorted(agg.items()):
    print("   %-28s %14.0f per launch  %10.2f per read  %8.4f per base  (%d launches)" % (k, sum(v) / len(v), sum(v) / len(v) / reads, sum(v) / len(v) / reads / L, len(v)))
PY
  rm -rf $OUT
done
