#!/bin/bash
# Per-kernel times of one bench workload: tools/kstats.sh <tag> <bench args...>
# -> gpurun_out/<tag>/kernel_stats.csv (rocprofv3 --kernel-trace --stats) and the bench line under the profiler
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
rm -rf $out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py "$@" --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err || { tail -5 $out/rocprof.err; exit 1; }
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
cut -c1-150 $out/kernel_stats.csv | head -12
