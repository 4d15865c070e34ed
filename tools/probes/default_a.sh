#!/bin/bash
# probe: the CLI with the default -a (hardware threads: queue-major output order) against -a 1
D=$(mktemp -d -p /dev/shm)
python3 - "$D" 10000000 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import e2e_bench as eb
eb.write_pair(sys.argv[1], int(sys.argv[2]))
PY
run() { local t0=$(date +%s%N); ./sickle_amd/sickle pe -f $D/R1.fastq -r $D/R2.fastq -t sanger -o $D/o1_$1 -p $D/o2_$1 -s $D/os_$1 $2 > /dev/null 2>&1; echo "$1: $(( ($(date +%s%N) - t0) / 1000000 )) ms"; }
run warm "-a 1"; run a1 "-a 1"; run default ""; run a16 "-a 16"; run a1b "-a 1"; run defaultb ""
ls -la $D/o1_a1 $D/o1_default | awk '{print $5}'
rm -rf "$D"
