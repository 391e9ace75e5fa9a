for c in 4 16 32 64 128; do
  timeout -k 10 120 python bench.py --workload chain --chunks $c --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunks $c', d['value'], d['unit'], d['ms_per_step'], 'fir+fft ms', d['config'].get('fir_fft_ms'), d['roofline'].get('fir_fft_ms'))" || exit 1
done
