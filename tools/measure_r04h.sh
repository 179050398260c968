# transform change: parity subset + bench lines
ulimit -c 0
D=gpurun_out/${1:-r04h}; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or tree_stereo or tree_mono or full_size" > $D/gpu_tests.log 2>&1; tail -3 $D/gpu_tests.log
timeout 500 python bench.py --cpu-baseline 0 --extras 0 --steps 10 > $D/bench_default.log 2> $D/bench_default.err
timeout 300 python bench.py --config rs468 --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_rs468.log 2>/dev/null
python - <<PY
import json
for f in ("default","rs468"):
    try:
        l=[x for x in open("$D/bench_%s.log" % f) if x.startswith("{")]
        d=json.loads(l[0]); print(f, round(d["value"],2), round(d["resolve_ms"],2), {k: round(v,2) for k,v in d["device_breakdown_ms"].items()}, "trf", d["kernels"]["trf"]["avg_launch_ms"], d["kernels"]["trf"]["frac_of_hbm_peak"], "schur", d["kernels"]["schur"]["avg_launch_ms"])
    except Exception as e: print(f, "ERR", e)
PY
