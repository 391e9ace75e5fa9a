# direct MFMA FIR: tiles per workgroup x resident workgroups per CU
for w in $1; do for b in $2; do for t in $3; do
  DSPMI_FIR_VARIANT=mfma DSPMI_FIR_BPC=$b DSPMI_FIR_TPB=$t timeout -k 10 120 python bench.py --workload $w --steps 10 --no-cpu-baseline --no-secondary > gpurun_out/sweep.log 2>&1 || { tail -5 gpurun_out/sweep.log; exit 1; }
  tail -1 gpurun_out/sweep.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w bpc=$b tpb=$t', r['achieved'], r['unit'], r['launch_ms'], 'ms')"
done; done; done
