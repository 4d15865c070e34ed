#!/bin/bash
# interleaved A/B of two builds of libsickle_amd.so on tools/long_rates.py (streaming kernel forced): ab_long_rates.sh old.so new.so
OLD=$1; NEW=$2
cp sickle_amd/libsickle_amd.so /tmp/lib_keep.so
for round in 1 2; do
  for which in old new; do
    if [ $which = old ]; then cp $OLD sickle_amd/libsickle_amd.so; else cp $NEW sickle_amd/libsickle_amd.so; fi
    echo "== round $round $which"
    SK_GENERAL=stream python3 tools/long_rates.py 2>&1 | grep -E "uniform (1000|2000|5000|10000|30000) \(|ragged 1-30 kb, hint"
  done
done
cp /tmp/lib_keep.so sickle_amd/libsickle_amd.so
