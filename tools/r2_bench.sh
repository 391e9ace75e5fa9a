#!/bin/bash
# Measurement pass: default bench, the same under rocprofv3 --kernel-trace --stats,
# the HBM-traffic counter passes for the 8191-tap overlap-save kernel, and a 2-rank rehearsal of
# the split-signal workload (gloo; both ranks on the one GPU).  Usage: tools/r2_bench.sh <tag>
set -o pipefail
tag=${1:-r2}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -5 $out/bench_default.err; exit 1; }
tail -c 600 $out/bench_default.json; echo
rm -rf $out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --steps 20 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err || { tail -5 $out/rocprof.err; exit 1; }
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats_default.csv
head -9 $out/kernel_stats_default.csv | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o p -- python3 bench.py --workload ols8191 --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_$c.log 2>&1 || { tail -5 $out/pmc_$c.log; exit 1; }
  echo "# rocprofv3 --pmc $c (KiB), bench.py --workload ols8191" | tee -a $out/pmc_ols8191.txt
  python3 tools/pmc_summary.py $(find $out/pmc_$c -name "*counter_collection.csv" | head -1) ols_ | tee -a $out/pmc_ols8191.txt
done
DSPMI_BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --workload ols8191_split --steps 10 --warmup 2 > $out/bench_split_2rank_gloo.json 2> $out/bench_split.err || { tail -5 $out/bench_split.err; exit 1; }
tail -c 900 $out/bench_split_2rank_gloo.json; echo
