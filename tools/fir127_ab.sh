#!/bin/bash
# 127-tap workload, two library builds alternating on one box: tools/fir127_ab.sh <other.so> [reps]
other=$1; reps=${2:-3}
mkdir -p gpurun_out
for r in $(seq $reps); do for v in other new; do
  if [ $v = other ]; then export DSPMI_LIB=$other; else unset DSPMI_LIB; fi
  timeout -k 10 200 python3 bench.py --workload fir127 --no-cpu-baseline --no-secondary > gpurun_out/fir_ab.json 2>gpurun_out/fir_ab.err || { tail -5 gpurun_out/fir_ab.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/fir_ab.json')); print('fir127 $v', d['value'], 'Gsamples/s', d['ms_per_step'], 'ms', d['roofline']['launch_ms_median'])"
done; done
