ulimit -c 0
D=gpurun_out/r03i; mkdir -p $D
for mode in late early; do
  if [ $mode = early ]; then export LSFM_PREFETCH_EARLY=1; else unset LSFM_PREFETCH_EARLY; fi
  for rep in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bnc_$mode$rep.log 2>/dev/null
  python - <<PY
import json
l=[x for x in open("$D/bnc_$mode$rep.log") if x.startswith("{")]
d=json.loads(l[0]); print("$mode", round(d["value"],2), round(d["resolve_ms"],2), round(d["roofline"]["frac"],4), round(d["kernels"]["schur"]["avg_launch_ms"],3), {k:round(v,2) for k,v in d["device_breakdown_ms"].items()}, d["max_rel_residual"], d["not_converged"])
PY
  done
done
