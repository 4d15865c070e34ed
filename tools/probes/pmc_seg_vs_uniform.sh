#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for V in seg150 u150dma; do
  OUT=$ROOT/gpurun_out/pmc_cmp/$V; mkdir -p $OUT
  if [ $V = u150dma ]; then export SK_TILE_STAGE=0; VV=u150; else unset SK_TILE_STAGE; VV=$V; fi
  i=0
  for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 $ROOT/bench.py --variant $VV --steps 5 > $OUT/pmc${i}.json 2> $OUT/pmc$i.err || tail -3 $OUT/pmc$i.err
  done
done
