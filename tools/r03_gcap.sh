ulimit -c 0
D=gpurun_out/r03f; mkdir -p $D
timeout 600 python bench.py --config synth16k --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 > $D/b16k.log 2>/dev/null
timeout 300 python bench.py --config rs468 --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/b468.log 2>/dev/null
python - <<PY
import json
for f in ("b16k","b468"):
    l=[x for x in open("$D/"+f+".log") if x.startswith("{")]
    d=json.loads(l[0]); print(f, round(d["value"],2), round(d["resolve_ms"],2), {k:round(v,2) for k,v in d["device_breakdown_ms"].items()}, d["kernels"]["trf"], d["max_rel_residual"], d["not_converged"])
PY
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "mono or aerial or synth16k" 2>&1 | tail -3
