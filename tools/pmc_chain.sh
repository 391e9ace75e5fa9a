#!/bin/bash
# Counters of the chain workload's kernels, each pass a run of its own:
#   tools/pmc_chain.sh <tag> [extra bench args]   -> gpurun_out/<tag>/pmc_chain.txt
# pass 1, 2, 3: SQ instruction counts / busy and wait cycles / MFMA; then FETCH_SIZE and WRITE_SIZE
# (KiB; FETCH_SIZE counts a 128-byte read request as 64 bytes on gfx950: doubled in the summary line).
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
res=$out/pmc_chain.txt
: > $res
run_pass() {
  name=$1; shift
  rm -rf $out/pmc_$name
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o p -- python3 bench.py --workload chain --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $out/pmc_$name.log 2>&1 || { tail -5 $out/pmc_$name.log; exit 1; }
  echo "# rocprofv3 --pmc $* -- python3 bench.py --workload chain --steps 3 --warmup 1 $EXTRA" >> $res
  python3 tools/pmc_summary.py $(find $out/pmc_$name -name "*counter_collection.csv" | head -1) anonymous | grep -v "tonegen_range\|tonegen_finish" >> $res
}
EXTRA="$*"
run_pass sq1 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAVES || exit 1
run_pass sq2 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
run_pass sq3 GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_ANY || exit 1
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE || exit 1
cat $res
