# round 6 baseline: the GPU suite and the default bench line of the tree as it stands
ulimit -c 0
D=gpurun_out/r06base; mkdir -p $D
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $D/gpu_tests.log 2>&1; tail -3 $D/gpu_tests.log
timeout 500 python bench.py --cpu-baseline 0 --extras 0 --steps 20 --warmup 3 > $D/bench_default.log 2> $D/bench_default.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r06base/bench_default.log") if x.startswith("{")]
d=json.loads(l[0]); print(round(d["value"],2), round(d["resolve_ms"],2), d["roofline"]["frac"], d["device_breakdown_ms"])
PY
