D=gpurun_out/r03f; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "early or tree_stereo or repeated or baseline_configuration or full_size_properties_without or error_paths" > $D/gpu_tests.log 2>&1; tail -5 $D/gpu_tests.log
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_occ3.json 2> /dev/null
timeout 300 python bench.py --config synth16k --maps 4096 --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench16k_occ3.json 2> /dev/null
touch linearsfm_amd/csrc/lsfm_schur_panel.hip; make -s -C linearsfm_amd/csrc K9_OCC16=2 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_occ2.json 2> /dev/null
python - <<'PY'
import json
for f in ("bench_occ3","bench16k_occ3","bench_occ2"):
    d=json.loads([l for l in open(f"gpurun_out/r03f/{f}.json") if l.startswith("{")][0])
    print(f, "value", round(d["value"],2), "resolve", round(d["resolve_ms"],2), {k:round(v,2) for k,v in d["device_breakdown_ms"].items()}, "K9", d["kernels"]["schur"]["ms_per_step"], d["roofline"]["frac"])
PY
