#!/bin/bash
# chain workload, two library builds alternating on one box: tools/chain_ab.sh <other.so> [reps] [taps]
other=$1; reps=${2:-3}; taps=${3:-127}
mkdir -p gpurun_out
for r in $(seq $reps); do for v in other new; do
  if [ $v = other ]; then export DSPMI_LIB=$other; else unset DSPMI_LIB; fi
  timeout -k 10 200 python3 bench.py --workload chain --chain-taps $taps --no-cpu-baseline > gpurun_out/chain_ab.json 2>gpurun_out/chain_ab.err || { tail -5 gpurun_out/chain_ab.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/chain_ab.json')); print('chain $taps $v', d['value'], 'Gsamples/s', d['ms_per_step'], 'ms')"
done; done
