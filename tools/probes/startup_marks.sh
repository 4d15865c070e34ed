#!/bin/bash
# the first half second of the CLI: every stage mark up to the third written batch (10 M pairs in tmpfs)
D=$(mktemp -d -p /dev/shm)
python3 - "$D" 10000000 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import e2e_bench as eb
eb.write_pair(sys.argv[1], int(sys.argv[2]))
PY
for i in 1 2 3; do
  rm -f $D/o1 $D/o2 $D/os
  SICKLE_STAGE_TIMES=1 ./sickle_amd/sickle pe -f $D/R1.fastq -r $D/R2.fastq -t sanger -o $D/o1 -p $D/o2 -s $D/os -a 1 2>&1 >/dev/null | grep "mark\]" | sed 's/(cpu.*//' | awk '/written/{w++} w<=3' | tr '\n' ';'; echo
done
rm -rf "$D"
