#!/bin/bash
# usage: tools/fft_gridx.sh "variants" "bpcs" "grid multipliers"
for v in $1; do for b in $2; do for g in $3; do
  DSPMI_FFT_VARIANT=$v DSPMI_FFT_BPC=$b DSPMI_GRIDX=$g timeout -k 10 120 python bench.py --workload fft1024 --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/sweep.log 2>&1 || { tail -5 gpurun_out/sweep.log; exit 1; }
  tail -1 gpurun_out/sweep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('v=$v bpc=$b gridx=$g', d['config']['kernel'], d['roofline']['achieved'], 'GB/s', d['roofline']['launch_ms'], 'ms')"
done; done; done
