# one line per box: the default transform launch against the prefetching persistent variant, and the FIR workloads
hostname
bash tools/fft_gridx.sh "0 5 0 5" "0" "0" | awk '{print $1, $(NF-3), $(NF-2)}' | tr '\n' ';'; echo
for w in fir16 fir127; do timeout -k 10 100 python bench.py --workload $w --steps 10 --no-cpu-baseline --no-secondary | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', r['achieved'], r['unit'], r['launch_ms'])"; done
