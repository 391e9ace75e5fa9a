#!/bin/bash
for d in 0 1 2 3; do
  echo "== DSPMI_FIR_DEBUG=$d (1: no loads/split after the first tiles, 2: no MFMAs)"
  DSPMI_FIR_DEBUG=$d timeout -k 10 120 python tools/fir_variants_probe.py 512 31 127 255 2>&1 | cut -c1-62 || exit 1
done
