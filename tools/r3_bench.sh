#!/bin/bash
# Measurement pass: the whole GPU test suite, the default bench, the same under
# rocprofv3 --kernel-trace --stats, and the HBM-traffic counter passes of the default run.
# Usage: tools/r3_bench.sh <tag>
set -o pipefail
tag=${1:-r3}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $out/pytest_gpu.log 2>&1 || { tail -20 $out/pytest_gpu.log; exit 1; }
tail -2 $out/pytest_gpu.log
timeout -k 10 400 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -5 $out/bench_default.err; exit 1; }
rm -rf $out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --steps 20 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err || { tail -5 $out/rocprof.err; exit 1; }
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats_default.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_$c.log 2>&1 || { tail -5 $out/pmc_$c.log; exit 1; }
  echo "# rocprofv3 --pmc $c (KiB) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline" >> $out/pmc_hbm_traffic.txt
  python3 tools/pmc_summary.py $(find $out/pmc_$c -name "*counter_collection.csv" | head -1) anonymous >> $out/pmc_hbm_traffic.txt
done
python3 - <<PY
import json
d = json.load(open("$out/bench_default.json"))
print("headline", d["value"], d["unit"], d["ms_per_step"], "ms", d["roofline"]["frac"])
for s in d["secondary"]:
    print("  %-70s %8.2f Gs/s %8.4f ms  frac %.4f" % (s["workload"][:70], s["gsamples_per_s"], s["ms_per_step"], s["roofline"]["frac"]))
PY
