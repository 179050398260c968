ulimit -c 0
D=gpurun_out/${1:-r04m}; mkdir -p $D
timeout 600 python bench.py --config synth16k --steps 4 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_synth16k.log 2>/dev/null
timeout 300 python bench.py --config rs468 --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_rs468.log 2>/dev/null
timeout 300 python bench.py --cpu-baseline 0 --extras 0 --steps 10 > $D/bench_default.log 2>/dev/null
python - <<PY
import json
for f in ("synth16k","rs468","default"):
    try:
        l=[x for x in open("$D/bench_%s.log" % f) if x.startswith("{")]
        d=json.loads(l[0]); print(f, round(d["value"],2), round(d["resolve_ms"],2), {k: round(v,2) for k,v in d["device_breakdown_ms"].items()}, d["max_rel_residual"], d["not_converged"])
    except Exception as e: print(f, "ERR", e)
PY
