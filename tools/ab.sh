#!/bin/bash
# usage: tools/ab.sh "workloads" [reps]  -- time tools/ab/libdspmi_base.so against dsp_amd/libdspmi.so, alternating
reps=${2:-3}
for w in $1; do for r in $(seq $reps); do for v in base new; do
  if [ $v = base ]; then export DSPMI_LIB=$PWD/tools/ab/libdspmi_base.so; else unset DSPMI_LIB; fi
  timeout -k 10 120 python bench.py --workload $w --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
  tail -1 gpurun_out/ab.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w $v', r['achieved'], r['unit'], r['launch_ms'], 'ms')"
done; done; done
