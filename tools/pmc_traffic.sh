#!/bin/bash
# HBM traffic per launch of the default bench's kernels: two separate counter passes.
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmct_$c
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmct_$c -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmct_$c.log 2>&1 || { tail -5 gpurun_out/pmct_$c.log; exit 1; }
  echo "# rocprofv3 --pmc $c (KiB)"
  python3 tools/pmc_summary.py $(find gpurun_out/pmct_$c -name "*counter_collection.csv" | head -1) anonymous
done
