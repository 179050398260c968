# round 3: the whole GPU suite + the default bench line (+ the Mono and 16k lines)
ulimit -c 0
D=gpurun_out/r03g; mkdir -p $D
timeout 1800 python -m pytest tests -x -q -m gpu --durations=8 > $D/gpu_tests.log 2>&1; tail -14 $D/gpu_tests.log
timeout 500 python bench.py > $D/bench_default.log 2> $D/bench_default.err
timeout 300 python bench.py --config rs468 --steps 10 --warmup 2 > $D/bench_rs468.log 2>/dev/null
timeout 300 python bench.py --config rs90 --steps 10 --warmup 2 > $D/bench_rs90.log 2>/dev/null
timeout 600 python bench.py --config synth16k --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_synth16k.log 2>/dev/null
python - <<'PY'
import json
for f in ("default","rs468","rs90","synth16k"):
    try:
        l=[x for x in open(f"gpurun_out/r03g/bench_{f}.log") if x.startswith("{")]
        d=json.loads(l[0]); print(f, round(d["value"],2), round(d["resolve_ms"],2), round(d["first_run_ms"],1), round(d["roofline"]["frac"],4), {k:round(v,2) for k,v in d["device_breakdown_ms"].items()}, (d.get("cpu_baseline") or {}).get("pose_param_max_rel_err_vs_oracle"))
    except Exception as e: print(f, "ERR", e)
PY
