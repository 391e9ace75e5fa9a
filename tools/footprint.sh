for f in 65536 262144 524288 1048576; do
  timeout -k 10 120 python bench.py --workload fft1024 --frames $f --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/fp.log 2>&1 || exit 1
  tail -1 gpurun_out/fp.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('frames $f', r['achieved'], r['unit'], r['launch_ms'], 'ms')"
done
for c in 512 2048; do
  timeout -k 10 120 python bench.py --workload fir16 --channels $c --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/fp.log 2>&1 || exit 1
  tail -1 gpurun_out/fp.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('fir16 channels $c', r['achieved'], r['unit'], r['launch_ms'], 'ms')"
done
