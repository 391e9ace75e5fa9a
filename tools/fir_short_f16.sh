#!/bin/bash
# f16 / bf16 / f32 FIR kernels on short and medium filters by tiles per workgroup (DSPMI_FIR_TPB)
for tpb in 3 4 5 6 8 16; do echo "== TPB $tpb"; DSPMI_FIR_TPB=$tpb timeout -k 10 100 python tools/fir_variants_probe.py 512 113 127 145 159 191 | cut -c1-60; done
