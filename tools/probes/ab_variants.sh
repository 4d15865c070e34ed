#!/bin/bash
# interleaved A/B of two builds of libsickle_amd.so on the bench variants named: ab_variants.sh old.so new.so variant...
OLD=$1; NEW=$2; shift 2
cp sickle_amd/libsickle_amd.so /tmp/lib_keep.so
for round in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then cp $OLD sickle_amd/libsickle_amd.so; else cp $NEW sickle_amd/libsickle_amd.so; fi
    for v in "$@"; do
      python3 bench.py --variant $v --steps 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $round $which %-10s %.4f ms  %.0f GB/s' % (d['variant'], d['kernel_ms_avg'], d['achieved']))"
    done
  done
done
cp /tmp/lib_keep.so sickle_amd/libsickle_amd.so
