#!/bin/bash
# SQ / GRBM counters of the FIR MFMA kernel for a workload: tools/pmc_fir.sh fir127
w=$1
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_$w
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_$w -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_$w.log 2>&1 || { tail -5 gpurun_out/pmc_$w.log; exit 1; }
f=$(find gpurun_out/pmc_$w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $f fir_mfma
grep -v "^[WEI]2" gpurun_out/pmc_$w.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', r['achieved'], r['unit'], r['launch_ms'], 'ms')"
