#!/bin/bash
# Single-kernel transforms with and without the whole-frame-in-registers kernels, same box.
for r in 0 1; do
  if [ $r = 1 ]; then export DSPMI_FFT_REG3=1; else unset DSPMI_FFT_REG3; fi
  echo "== DSPMI_FFT_REG3=$r"
  timeout -k 10 200 python tools/fft_sizes.py 2048 4096 8192 16384 32768 || exit 1
  timeout -k 10 200 python tools/fft_sizes.py f64 1024 2048 4096 8192 || exit 1
done
