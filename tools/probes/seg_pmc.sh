#!/bin/bash
# Where the segmented kernel's time goes next to the uniform LDS-DMA kernel ON THE SAME DATA (one length):
# kernel trace, then two --pmc passes (instruction mix; wave-cycles), per kernel name.  Never combined with
# other trace domains.  Usage (via gpurun): bash tools/probes/seg_pmc.sh "250 301" > out.log
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
LENS=${1:-"250 301"}
cd /tmp && export TMPDIR=/tmp
for L in $LENS; do
  OUT=$ROOT/gpurun_out/seg_pmc_$L
  rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/probes/seg_case.py $L 40 > $OUT/run0.log 2> $OUT/err0.log || { echo "$L trace failed"; tail -3 $OUT/err0.log; }
  i=0
  for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" \
             "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_EXP_GDS"; do
    i=$((i+1))
    timeout -k 10 180 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/probes/seg_case.py $L 6 > $OUT/run$i.log 2> $OUT/err$i.log || { echo "$L pass $i failed"; tail -3 $OUT/err$i.log; }
  done
  python3 - $OUT $L <<'PY'
import csv, glob, sys, collections
out, L = sys.argv[1], int(sys.argv[2])
st = ((L + 7) // 8 | 1) * 8
tiles = 1_000_000_000 // st // 64
for f in glob.glob(out + "/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "sk_scan_" in r["Name"]:
            print("trace  %-60s calls %s avg %.1f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sk_scan_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
print("== %d bp, %d tiles per launch; per tile:" % (L, tiles))
for c, d in sorted(agg.items()):
    print("   %-22s" % c + "".join("  %s %12.1f" % (k[-22:], sum(v) / len(v) / tiles) for k, v in sorted(d.items())))
PY
  rm -rf $OUT/p* $OUT/t
done
