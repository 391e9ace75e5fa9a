#!/bin/bash
# f16 FIR kernel by tiles per workgroup (DSPMI_FIR_TPB) and tile walk (DSPMI_FIR_WALK=front: a workgroup's tiles a grid apart)
# usage: tools/fir_walk.sh "<taps...>" "<tpb...>" "<walks>"
for tpb in ${2:-4 16 64}; do for walk in ${3:-rows front}; do
  echo "== TPB $tpb walk $walk"
  DSPMI_FIR_WALK=$walk DSPMI_FIR_TPB=$tpb timeout -k 10 100 python tools/fir_variants_probe.py 512 ${1:-16 49 127 255} | cut -c1-58 || exit 1
done; done
