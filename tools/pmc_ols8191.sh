#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the 8191-tap overlap-save kernel on the 2^26-sample signal.
export TMPDIR=/tmp
out=gpurun_out/${1:-pmc_ols}
mkdir -p $out
: > $out/pmc_ols8191.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o p -- python3 bench.py --workload ols8191 --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_$c.log 2>&1 || { tail -5 $out/pmc_$c.log; exit 1; }
  echo "# rocprofv3 --pmc $c (KiB), bench.py --workload ols8191" | tee -a $out/pmc_ols8191.txt
  python3 tools/pmc_summary.py $(find $out/pmc_$c -name "*counter_collection.csv" | head -1) ols_ | tee -a $out/pmc_ols8191.txt
done
