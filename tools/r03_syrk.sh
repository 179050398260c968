ulimit -c 0
D=gpurun_out/r03d; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu > $D/tests.log 2>&1; tail -4 $D/tests.log
for mode in mfma scalar; do
  if [ $mode = scalar ]; then export LSFM_SN_SCALAR_UPDATE=1; else unset LSFM_SN_SCALAR_UPDATE; fi
  timeout 600 python bench.py --config synth16k --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 > $D/b16k_$mode.log 2>/dev/null
  timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bnc_$mode.log 2>/dev/null
  python - <<PY
import json
for f in ("b16k_$mode","bnc_$mode"):
    l=[x for x in open("$D/"+f+".log") if x.startswith("{")]
    d=json.loads(l[0]); print(f, round(d["value"],2), round(d["resolve_ms"],2), {k:round(v,2) for k,v in d["device_breakdown_ms"].items()}, d["max_rel_residual"], d["not_converged"])
PY
done
