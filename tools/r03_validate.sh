ulimit -c 0
D=gpurun_out/r03m; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_sharded.py -x -q -m gpu -k "not synth16k_mono_full and not 16384" > $D/gpu_tests.log 2>&1; tail -5 $D/gpu_tests.log
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench.json 2> $D/bench.err
LSFM_TIMELINE=1 timeout 300 python bench.py --steps 2 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_tl.json 2> $D/timeline.txt
python - <<'PY'
import json
for f in ("bench",):
    d=json.loads([l for l in open(f"gpurun_out/r03m/{f}.json") if l.startswith("{")][0])
    print(f, "value", round(d["value"],2), "resolve", round(d["resolve_ms"],2), {k:round(v,2) for k,v in d["device_breakdown_ms"].items()}, "its", d["pcg_iterations_per_step"], d["max_rel_residual"], d["roofline"]["frac"])
PY
sed -n 28,41p $D/timeline.txt | cut -c1-330
