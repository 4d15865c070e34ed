#!/bin/bash
# the streaming general kernel at each prefetch depth (blocks of 1 KiB in flight per wave) / wave cap / read cost, same box
for D in 4 3 2; do
  echo "== SK_STREAM_DEPTH=$D"
  SK_GENERAL=stream SK_STREAM_DEPTH=$D timeout -k 10 200 python tools/long_rates.py 2>&1 | grep -E "uniform (1000|5000|10000|30000) \(|ragged"
done
for W in 24 32; do
  echo "== SK_STREAM_DEPTH=2 SK_STREAM_WAVES=$W"
  SK_GENERAL=stream SK_STREAM_DEPTH=2 SK_STREAM_WAVES=$W timeout -k 10 200 python tools/long_rates.py 2>&1 | grep -E "uniform (1000|5000|10000|30000) \(|ragged"
done
for C in 0 2048 8192; do
  echo "== SK_STREAM_READ_COST=$C"
  SK_GENERAL=stream SK_STREAM_READ_COST=$C timeout -k 10 200 python tools/long_rates.py 2>&1 | grep -E "ragged"
done
