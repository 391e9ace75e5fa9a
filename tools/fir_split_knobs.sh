#!/bin/bash
# f16 FIR kernel: tiles per workgroup (DSPMI_FIR_TPB; 512 channels x 2^20 samples = 131072 tiles), 63, 127 and 305 taps
for tpb in 4 8 16 32 43 57 86 171 342; do
  echo "== DSPMI_FIR_TPB=$tpb"
  DSPMI_FIR_TPB=$tpb timeout -k 10 120 python tools/fir_variants_probe.py 512 63 127 305 | cut -c1-60 || exit 1
done
