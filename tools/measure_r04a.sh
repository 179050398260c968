# round 4, first GPU pass: the suite, determinism / stability of the fixed-point factorisation, the headline bench
ulimit -c 0
D=gpurun_out/r04a; mkdir -p $D
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $D/gpu_tests.log 2>&1; tail -3 $D/gpu_tests.log
export LSFM_FACTOR_DIGEST=1
timeout 300 python tools/stability_16k.py 20 nc3500 > $D/stab_nc3500.txt 2>&1; tail -2 $D/stab_nc3500.txt
timeout 300 python tools/stability_16k.py 20 rs468 > $D/stab_rs468.txt 2>&1; tail -2 $D/stab_rs468.txt
timeout 300 python tools/stability_16k.py 20 nc3500 plans 200 > $D/stab_nc200.txt 2>&1; tail -2 $D/stab_nc200.txt
timeout 900 python tools/stability_16k.py 30 synth16k > $D/stab_16k.txt 2>&1; tail -2 $D/stab_16k.txt
unset LSFM_FACTOR_DIGEST
timeout 500 python bench.py --cpu-baseline 0 --extras 0 --steps 10 > $D/bench_default.log 2> $D/bench_default.err
timeout 600 python bench.py --config synth16k --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_synth16k.log 2>/dev/null
python - <<'PY'
import json
for f in ("default","synth16k"):
    try:
        l=[x for x in open(f"gpurun_out/r04a/bench_{f}.log") if x.startswith("{")]
        d=json.loads(l[0]); print(f, round(d["value"],2), round(d["resolve_ms"],2), d["device_breakdown_ms"], d["max_rel_residual"], d["not_converged"])
    except Exception as e: print(f, "ERR", e)
PY
