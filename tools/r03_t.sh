ulimit -c 0
D=gpurun_out/r03k; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -3
for mode in worker noworker; do
  if [ $mode = noworker ]; then export LSFM_NO_WORKER=1; else unset LSFM_NO_WORKER; fi
  for rep in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bnc_$mode$rep.log 2>/dev/null
  python - <<PY
import json
l=[x for x in open("$D/bnc_$mode$rep.log") if x.startswith("{")]
d=json.loads(l[0]); print("$mode", round(d["value"],2), round(d["resolve_ms"],2), round(d["roofline"]["frac"],4), round(d["kernels"]["schur"]["avg_launch_ms"],3), {k:round(v,2) for k,v in d["device_breakdown_ms"].items()}, d["max_rel_residual"], d["not_converged"])
PY
  done
done
unset LSFM_NO_WORKER
LSFM_TIMELINE=1 timeout 300 python bench.py --steps 2 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_tl.json 2> $D/timeline.txt
grep -n "^\[tl\]" $D/timeline.txt | sed -n 44,52p | cut -c1-300
