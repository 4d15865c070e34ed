#!/bin/bash
# wall clock around `sickle pe` (10 M pairs of 150 bp in tmpfs, -a 1) with the front process and without (SICKLE_NO_FRONT=1), interleaved
D=$(mktemp -d -p /dev/shm)
python3 - "$D" 10000000 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import e2e_bench as eb
eb.write_pair(sys.argv[1], int(sys.argv[2]))
PY
for round in 1 2 3 4; do
  for nf in 0 1; do
    rm -f $D/o1 $D/o2 $D/os
    S=$(date +%s.%N)
    SICKLE_NO_FRONT=$nf ./sickle_amd/sickle pe -f $D/R1.fastq -r $D/R2.fastq -t sanger -o $D/o1 -p $D/o2 -s $D/os -a 1 > $D/out 2> $D/err
    E=$(date +%s.%N)
    echo "round $round SICKLE_NO_FRONT=$nf: $(python3 -c "print(round($E - $S, 3))") s  rc $?  md5 $(cat $D/o1 $D/o2 $D/os | md5sum | cut -c1-8)  $(grep -c kept $D/out) summary lines"
    sleep 0.5
  done
done
rm -rf "$D"
