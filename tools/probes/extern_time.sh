#!/bin/bash
D=$(mktemp -d -p /dev/shm)
python3 - "$D" 10000000 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import e2e_bench as eb
eb.write_pair(sys.argv[1], int(sys.argv[2]))
PY
for i in 1 2 3; do
  rm -f $D/o1 $D/o2 $D/os
  S=$(date +%s.%N)
  SICKLE_STAGE_TIMES=1 ./sickle_amd/sickle pe -f $D/R1.fastq -r $D/R2.fastq -t sanger -o $D/o1 -p $D/o2 -s $D/os -a 1 2> $D/err >/dev/null
  E=$(date +%s.%N)
  echo "run $i: external $(python3 -c "print(round($E - $S, 3))") s; internal: $(grep -E 'closed|device open' $D/err | tr '\n' ' ' | sed 's/\[mark\]//g')"
done
# how long does a trivial HIP program take to start and exit?
S=$(date +%s.%N); ./sickle_amd/sickle --version > /dev/null; E=$(date +%s.%N); echo "sickle --version (loads libamdhip64, no HIP call): $(python3 -c "print(round($E - $S, 3))") s"
S=$(date +%s.%N); python3 -c "import ctypes; l=ctypes.CDLL('sickle_amd/libsickle_amd.so'); import ctypes as C; h=C.c_void_p(); l.sk_create(0,2,C.byref(h)); l.sk_destroy(h)"; E=$(date +%s.%N); echo "python: sk_create + sk_destroy: $(python3 -c "print(round($E - $S, 3))") s"
rm -rf "$D"
