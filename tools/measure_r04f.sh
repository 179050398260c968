# the sharded tests (several ranks on one GPU over gloo), then the one-GPU logic check of bench.py --gpus N (starts its own ranks)
ulimit -c 0
D=gpurun_out/${1:-r04f}; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_cli.py -x -q -m gpu --durations=5 > $D/gpu_tests.log 2>&1; tail -5 $D/gpu_tests.log
export LSFM_BENCH_ONE_GPU=1
for n in 2 8; do
timeout 900 python bench.py --gpus $n --steps 2 --warmup 1 > $D/bench_onegpu$n.log 2> $D/bench_onegpu$n.err
python - <<PY
import json
try:
    l=[x for x in open("$D/bench_onegpu$n.log") if x.startswith("{")]
    d=json.loads(l[0]); print("$n", d["n_gpus"], round(d["value"],1), round(d["resolve_ms"],1), [round(v,1) for v in d["per_rank_device_ms"]], d["per_rank_phases_ms"], d["max_rel_residual"], d["not_converged"])
except Exception as e:
    print("$n ERR", e); print(open("$D/bench_onegpu$n.err").read()[-2500:])
PY
done
