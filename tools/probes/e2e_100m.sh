#!/bin/bash
# BASELINE configs[3] through the CLI: 50 M pairs = 100 M reads of 150 bp (32 GB in tmpfs), this CLI only
# (the reference takes 91-93 s at -a 1 and 71-73 s at -a 16 on this input: round 1, tools/e2e_bench.py)
N=${1:-50000000}
D=$(mktemp -d -p /dev/shm)
python3 - "$D" "$N" <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import e2e_bench as eb
t = time.time(); eb.write_pair(sys.argv[1], int(sys.argv[2])); print("input written in %.1f s" % (time.time() - t), flush=True)
PY
ls -la $D | head -5
for i in 1 2 3; do
  rm -f $D/o1 $D/o2 $D/os
  S=$(date +%s.%N)
  SICKLE_STAGE_TIMES=1 ./sickle_amd/sickle pe -f $D/R1.fastq -r $D/R2.fastq -t sanger -o $D/o1 -p $D/o2 -s $D/os -a 1 2> $D/err > $D/out
  E=$(date +%s.%N)
  echo "run $i: external $(python3 -c "print(round($E - $S, 3))") s; $(grep -E 'closed' $D/err | sed 's/\[mark\]//; s/(cpu/ (cpu/') $(grep -E '^\[stage\]' $D/err | tr '\n' ' ')"
  grep -E "records kept|discarded" $D/out | tr '\n' ' '; echo
done
rm -rf "$D"
