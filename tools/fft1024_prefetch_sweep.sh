#!/bin/bash
# N = 1024 c2c: the prefetching kernel (DSPMI_FFT_VARIANT=5) launched with a grid the dispatcher refills
# (DSPMI_FFT_ITERS frame-blocks per workgroup) and 2 or 3 workgroups per CU, against the default kernel.
run() { python3 bench.py --workload fft1024 --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms %.0f GB/s %s' % (d['ms_per_step'], d['roofline']['achieved'], d['config']['kernel']))"; }
echo "default: $(run)"
for bpc in 2 3; do for it in 0 3 6 12 24; do
  if [ $it = 0 ]; then echo "variant 5 bpc $bpc persistent: $(DSPMI_FFT_VARIANT=5 DSPMI_FFT_BPC=$bpc run)"; else
  echo "variant 5 bpc $bpc iters $it: $(DSPMI_FFT_VARIANT=5 DSPMI_FFT_BPC=$bpc DSPMI_FFT_ITERS=$it run)"; fi
done; done
echo "default: $(run)"
