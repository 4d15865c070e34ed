#!/bin/bash
# diagnostic: stage times of the product CLI on a synthetic paired input (GPU box)
N=${1:-2000000}
D=$(mktemp -d -p /dev/shm)
python3 - "$D" "$N" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import e2e_bench as eb
eb.write_pair(sys.argv[1], int(sys.argv[2]))
PY
for i in 1 2; do
  T0=$(date +%s.%N); SICKLE_STAGE_TIMES=1 ./sickle_amd/sickle pe -f $D/R1.fastq -r $D/R2.fastq -t sanger -o $D/o1 -p $D/o2 -s $D/os -a 1 2>&1 >/dev/null | grep -v 'framed\|submitted' | tr '\n' ';' | sed 's/\[mark\]//g; s/  */ /g'; echo
done
echo "SE:"; SICKLE_STAGE_TIMES=1 ./sickle_amd/sickle se -f $D/R1.fastq -t sanger -o $D/se_out -a 1 2>&1 >/dev/null | grep -E "stage|closed|device open" | tr '\n' ';'; echo
rm -rf "$D"
