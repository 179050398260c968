ulimit -c 0
D=gpurun_out/r03u; mkdir -p $D

T0=$(date +%s); timeout 500 python bench.py > $D/bench_default.log 2> $D/bench_default.err; echo "bench wall s: $(( $(date +%s) - T0 ))"
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r03u/bench_default.log") if x.startswith("{")]
d=json.loads(l[0]); cb=d["cpu_baseline"]; print(d["value"], d["resolve_ms"], cb["value"], {k:v for k,v in cb["sample_legs"].items() if k!="note"}, cb["pose_param_max_rel_err_vs_oracle"])
PY
