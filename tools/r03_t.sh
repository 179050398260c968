ulimit -c 0
D=gpurun_out/r03n; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_sharded.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -3
for c in rs468 rs90 aerial; do
timeout 300 python bench.py --config $c --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/b_$c.log 2>/dev/null
python - <<PY
import json
l=[x for x in open("$D/b_$c.log") if x.startswith("{")]
d=json.loads(l[0]); print("$c", round(d["value"],2), round(d["resolve_ms"],2), d["max_rel_residual"], d["not_converged"])
PY
done
