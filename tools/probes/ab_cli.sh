#!/bin/bash
# A/B two CLI builds on the same box and input: usage ab_cli.sh <pairs> binA binB
N=$1; A=$2; B=$3
D=$(mktemp -d -p /dev/shm)
python3 - "$D" "$N" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import e2e_bench as eb
eb.write_pair(sys.argv[1], int(sys.argv[2]))
PY
for i in 1 2 3 4; do
  for X in $A $B; do
    S=$(date +%s%N)
    $X pe -f $D/R1.fastq -r $D/R2.fastq -t sanger -o $D/o1 -p $D/o2 -s $D/os -a 1 >/dev/null 2>&1
    E=$(date +%s%N)
    echo "$X PE $(( (E - S) / 1000000 )) ms"
    S=$(date +%s%N)
    $X se -f $D/R1.fastq -t sanger -o $D/o1 -a 1 >/dev/null 2>&1
    E=$(date +%s%N)
    echo "$X SE $(( (E - S) / 1000000 )) ms"
  done
done
rm -rf "$D"
