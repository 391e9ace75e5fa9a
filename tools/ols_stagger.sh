#!/bin/bash
# 8191-tap overlap-save on 2^26 samples by start-time spread (DSPMI_OLS_STAGGER, clock ticks) and by who is
# spread (DSPMI_OLS_STAGGER_ALL: 1 = every workgroup, 0 = only those that stride over one pair less)
for r in 1 2; do for all in 0 1; do for v in ${1:-60000 80000}; do
  DSPMI_OLS_STAGGER_ALL=$all DSPMI_OLS_STAGGER=$v timeout -k 10 120 python bench.py --workload ols8191 --steps 10 --no-cpu-baseline --no-secondary ${2:-} > gpurun_out/stag.log 2>&1 || { tail -5 gpurun_out/stag.log; exit 1; }
  tail -1 gpurun_out/stag.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('all $all stagger $v', r['achieved'], r['unit'], r['launch_ms'], 'ms')"
done; done; done
