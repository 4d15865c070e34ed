#!/bin/bash
# A/B on one box: an ingest batch through the device whole against in pieces (host/trim.h: piece_reads), -a 1,
# 10 M pairs of 150 bp on tmpfs, wall clock around the process.
D=$(mktemp -d -p /dev/shm)
python3 - "$D" 10000000 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import e2e_bench as eb
eb.write_pair(sys.argv[1], int(sys.argv[2]))
PY
for round in 1 2 3; do
  for piece in 1000000000 2400000 1200000 600000 300000; do
    rm -f $D/o1 $D/o2 $D/os
    S=$(date +%s.%N)
    SICKLE_SUBBATCH_READS=$piece ./sickle_amd/sickle pe -f $D/R1.fastq -r $D/R2.fastq -t sanger -o $D/o1 -p $D/o2 -s $D/os -a 1 2> $D/err >/dev/null
    E=$(date +%s.%N)
    echo "round $round piece $piece: $(python3 -c "print(round($E - $S, 3))") s  md5 $(cat $D/o1 $D/o2 $D/os | md5sum | cut -c1-8)"
  done
done
rm -rf "$D"
