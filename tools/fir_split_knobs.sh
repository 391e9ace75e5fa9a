#!/bin/bash
# bf16 FIR kernel: tiles per workgroup and resident workgroups per CU, 31 and 127 taps
for tpb in 1 2 4 8 16 32; do
  echo "== DSPMI_FIR_TPB=$tpb"
  DSPMI_FIR_TPB=$tpb timeout -k 10 120 python tools/fir_variants_probe.py 512 31 127 || exit 1
done
for bpc in 1 2 3; do
  echo "== DSPMI_FIR_BPC=$bpc"
  DSPMI_FIR_BPC=$bpc timeout -k 10 120 python tools/fir_variants_probe.py 512 31 127 || exit 1
done
