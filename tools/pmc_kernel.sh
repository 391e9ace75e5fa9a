#!/bin/bash
# SQ counters of one kernel of a bench workload: tools/pmc_kernel.sh <workload> <kernel-name-filter> [pass]
w=$1; k=$2; pass=${3:-1}
export TMPDIR=/tmp
if [ $pass = 1 ]; then ctr="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAVES"
else ctr="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; fi
rm -rf gpurun_out/pmck_$w_$pass
timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmck_${w}_$pass -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmck_${w}_$pass.log 2>&1 || { tail -5 gpurun_out/pmck_${w}_$pass.log; exit 1; }
f=$(find gpurun_out/pmck_${w}_$pass -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $f $k
