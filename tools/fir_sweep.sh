#!/bin/bash
# usage: tools/fir_sweep.sh "labels" "bpcs" [workload]   (labels only tag the output lines)   (on the GPU box, from the repo root)
w=${3:-fir127}
for m in $1; do for b in $2; do
  DSPMI_FIR_BPC=$b timeout -k 10 120 python bench.py --workload $w --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/sweep.log 2>&1 || { tail -5 gpurun_out/sweep.log; exit 1; }
  tail -1 gpurun_out/sweep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$m bpc=$b', r['achieved'], r['unit'], r['launch_ms'], 'ms')"
done; done
