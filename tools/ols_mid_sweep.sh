#!/bin/bash
# Overlap-save transform length by tap count, 64 channels x 2^22 samples: which length the plan should choose.
for nh in 600 1023 1500 2048 2049 3000 4096; do
  for n in 4096 8192 16384 32768; do
    if [ $n -ge $((2 * nh)) ]; then timeout -k 10 60 python tools/ols_one.py $nh $n 64 4194304 || exit 1; fi
  done
done
