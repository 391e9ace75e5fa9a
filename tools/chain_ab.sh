#!/bin/bash
# chain workload: FIR + spectrum in one kernel against the two-kernel form
for flag in "" "--chain-two-kernels"; do
  timeout -k 10 120 python bench.py --workload chain --steps 20 --warmup 3 --no-cpu-baseline --no-secondary $flag 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$flag', d['value'], d['unit'], d['ms_per_step'], 'ms; fir+fft alone', round(d['roofline'].get('fir_fft_ms', d['config'].get('fir_fft_ms', 0)) or 0, 3), d['config']['kernel'])" || exit 1
done
