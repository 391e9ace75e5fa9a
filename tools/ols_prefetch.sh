#!/bin/bash
# 8191-tap overlap-save: next pair's input requested ahead of the stores (default) against DSPMI_OLS_PREFETCH=0, alternating
for r in 1 2 3; do for v in 0 1; do
  DSPMI_OLS_PREFETCH=$v timeout -k 10 120 python bench.py --workload ols8191 --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/pf.log 2>&1 || { tail -5 gpurun_out/pf.log; exit 1; }
  tail -1 gpurun_out/pf.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('prefetch $v', r['achieved'], r['unit'], r['launch_ms'], 'ms')"
done; done
