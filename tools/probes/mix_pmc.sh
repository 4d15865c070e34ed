#!/bin/bash
# Counters of the kernels of tools/probes/mix_case.py (segmented / regrouped ragged / uniform 150), per launch:
# instruction mix, where the wave-cycles go, HBM traffic -- three --pmc passes, never combined with other trace domains.
# Usage: [SEQ=1] [SK_SEG_STAGE=0] bash tools/probes/mix_pmc.sh [reads]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-4000000}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/mix_pmc_$$
rm -rf $OUT; mkdir -p $OUT
i=0
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/probes/mix_case.py $N 6 0 > $OUT/run$i.log 2> $OUT/err$i.log || { echo "pass $i failed"; tail -3 $OUT/err$i.log; }
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "sk_" in k:
            agg[r["Counter_Name"]][k[-30:]].append(float(r["Counter_Value"]))
names = sorted({k for d in agg.values() for k in d})
print("%-22s" % "per launch" + "".join("%32s" % k for k in names))
for c, d in sorted(agg.items()):
    scale = 2 * 1024 / 1e6 if c == "FETCH_SIZE" else 1024 / 1e6 if c == "WRITE_SIZE" else 1e-6  # KB -> MB; FETCH_SIZE x2: the gfx950 correction of the guide
    print("%-22s" % (c + (" MB" if "SIZE" in c else " M")) + "".join("%32.2f" % (sum(d[k]) / len(d[k]) * scale if k in d else 0) for k in names))
PY
cat $OUT/run1.log
rm -rf $OUT
