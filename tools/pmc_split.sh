#!/bin/bash
# SQ counters of the FIR kernel the plan picks on the bench's 127-tap workload, three passes
export TMPDIR=/tmp
w=${1:-fir127}
p1="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES"
p2="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
p3="GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH"
i=0
for ctr in "$p1" "$p2" "$p3"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcs_$i
  timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmcs_$i -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmcs_$i.log 2>&1 || { tail -5 gpurun_out/pmcs_$i.log; exit 1; }
  f=$(find gpurun_out/pmcs_$i -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py $f fir_
done
