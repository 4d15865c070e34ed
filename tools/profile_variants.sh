#!/bin/bash
# Profiles every kernel variant of tools/variants.py on the GPU box (via gpurun): per variant a
# kernel-trace + stats pass and four counter passes (each its own run, --pmc never combined with other
# trace domains), all of `python3 bench.py --variant NAME --steps 30` -- the program itself after `--`.
# Usage: bash tools/profile_variants.sh <tag> [variant ...]      -> gpurun_out/prof_<tag>/<variant>/
set -o pipefail
TAG=${1:-r02}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
VARIANTS=${*:-"n150 u250 u100 seg seg_n seg_scatter packed150 ragged150 ragged_mix u600 u1000 long long30k"}
cd /tmp && export TMPDIR=/tmp
for V in $VARIANTS; do
  OUT=$ROOT/gpurun_out/prof_$TAG/$V
  mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --variant $V --steps 30 > "$OUT/trace_bench.json" 2> "$OUT/trace.err" || { echo "$V: trace pass failed"; tail -3 "$OUT/trace.err"; continue; }
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc$i" -- python3 $ROOT/bench.py --variant $V --steps 5 > "$OUT/pmc${i}_bench.json" 2> "$OUT/pmc$i.err" || { echo "$V: pmc pass $i failed"; tail -3 "$OUT/pmc$i.err"; }
  done
  find "$OUT" -name "*.csv" -size +16M -delete
  echo "$V done: $(cat $OUT/trace_bench.json | tail -1 | cut -c1-200)"
done
