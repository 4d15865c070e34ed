#!/bin/bash
# 100 M reads through the CLI with and without the front process (wall clock of the launching shell)
D=$(mktemp -d -p /dev/shm)
python3 - "$D" ${1:-50000000} <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import e2e_bench as eb
eb.write_pair(sys.argv[1], int(sys.argv[2]))
PY
for round in 1 2; do
  for nf in 0 1; do
    rm -f $D/o1 $D/o2 $D/os
    sleep 2
    S=$(date +%s.%N)
    SICKLE_NO_FRONT=$nf SICKLE_STAGE_TIMES=1 ./sickle_amd/sickle pe -f $D/R1.fastq -r $D/R2.fastq -t sanger -o $D/o1 -p $D/o2 -s $D/os -a 1 2> $D/err > $D/out
    E=$(date +%s.%N)
    M=$(grep "wall. main" $D/err | awk '{print $3}'); X=$(grep "wall. exit" $D/err | awk '{print $3}')
    echo "round $round SICKLE_NO_FRONT=$nf: external $(python3 -c "print(round($E - $S, 3))") s = start-up $(python3 -c "print(round($M - $S, 3))") + main..exit $(python3 -c "print(round($X - $M, 3))") + after exit $(python3 -c "print(round($E - $X, 3))"); $(grep -E 'closed' $D/err | sed 's/(cpu.*//' | tr '\n' ' ')"
  done
done
rm -rf "$D"
