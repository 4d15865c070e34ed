#!/bin/bash
# the streaming general kernel at each prefetch depth (blocks of 1 KiB in flight per wave), same box
for D in 3 2 4 8; do
  echo "== SK_STREAM_DEPTH=$D"
  SK_GENERAL=stream SK_STREAM_DEPTH=$D timeout -k 10 200 python tools/long_rates.py 2>&1 | grep -E "uniform (5000|10000|30000|100000) \(|ragged"
done
