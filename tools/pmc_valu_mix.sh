#!/bin/bash
# Vector-ALU instruction mix of one kernel of a bench workload: tools/pmc_valu_mix.sh <workload> <kernel filter>
w=$1; k=$2
export TMPDIR=/tmp
rm -rf gpurun_out/pmcv_$w
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 --output-format csv -d gpurun_out/pmcv_$w -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmcv_$w.log 2>&1 || { tail -5 gpurun_out/pmcv_$w.log; exit 1; }
python3 tools/pmc_summary.py $(find gpurun_out/pmcv_$w -name "*counter_collection.csv" | head -1) $k
