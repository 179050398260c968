# round 3, first GPU call: GPU tests, default bench line, rocprofv3 kernel stats of the same command
ulimit -c 0
D=gpurun_out/r03a; mkdir -p $D
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > $D/gpu_tests.log 2>&1; tail -25 $D/gpu_tests.log
timeout 500 python bench.py > $D/bench_default.log 2> $D/bench_default.err; tail -c 1500 $D/bench_default.err
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o run -- python3 bench.py --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_prof.log 2>/dev/null
python - <<'PY'
import json
for f in ("bench_default","bench_prof"):
    try:
        l=[x for x in open(f"gpurun_out/r03a/{f}.log") if x.startswith("{")]
        d=json.loads(l[0]); print(f, d["value"], d.get("resolve_ms"), d.get("first_run_ms"), d["roofline"]["frac"], d["device_breakdown_ms"], (d.get("cpu_baseline") or {}).get("pose_param_max_rel_err_vs_oracle"))
    except Exception as e: print(f, "ERR", e)
PY
