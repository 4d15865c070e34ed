#!/bin/bash
# does run k+1 of the default `sickle pe` pay for run k's teardown?  The front process hands control
# back when the outputs are closed; the worker behind it is still unmapping, unpinning and destroying its HIP context.
# Five DEFAULT runs back to back on the same input (the shape of the reference's trim_all.py loop), total wall / 5,
# against five SICKLE_NO_FRONT=1 runs back to back and against single runs with a pause before each.
N=${1:-10000000}
D=$(mktemp -d -p /dev/shm)
python3 - "$D" $N <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import e2e_bench as eb
eb.write_pair(sys.argv[1], int(sys.argv[2]))
PY
run() { ./sickle_amd/sickle pe -f $D/R1.fastq -r $D/R2.fastq -t sanger -o $D/o1 -p $D/o2 -s $D/os -a 1 > /dev/null 2> $D/err || { echo "run failed"; cat $D/err; }; }
now() { date +%s.%N; }
run; sleep 2   # first touch of the GPU runtime and of the input's pages
for mode in front nofront; do
  if [ $mode = nofront ]; then export SICKLE_NO_FRONT=1; else unset SICKLE_NO_FRONT; fi
  # single runs, the machine at rest before each
  S=0
  for i in 1 2 3; do sleep 2; A=$(now); run; B=$(now); S=$(python3 -c "print($S + $B - $A)"); done
  echo "$mode: single runs with a pause before each: $(python3 -c "print(round($S / 3, 3))") s per run"
  sleep 2
  A=$(now); for i in 1 2 3 4 5; do run; done; B=$(now)
  echo "$mode: five runs back to back: $(python3 -c "print(round(($B - $A) / 5, 3))") s per run"
done
sleep 1
rm -rf "$D"
