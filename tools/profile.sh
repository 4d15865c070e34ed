#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (each in its own run,
# never combined with other trace domains).  Usage (via gpurun): bash tools/profile.sh <tag> [bench args]
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the trace + stats pass runs the SAME command as the plain bench (default steps/warmup, CPU baseline
# included); the counter passes only need a few launches
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --no-e2e $* > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 --settle 20 $*"
echo "trace done"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc${i}_bench.json" 2> "$OUT/pmc$i.err" || { echo "pmc pass $i failed"; tail -3 "$OUT/pmc$i.err"; }
  echo "pmc $i done"
done
# keep only the CSVs that matter small
find "$OUT" -name "*.csv" -size +200M -delete
ls -R "$OUT" | head -50
