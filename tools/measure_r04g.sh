ulimit -c 0
D=gpurun_out/${1:-r04g}; mkdir -p $D
timeout 600 python bench.py --cpu-baseline 0 --steps 5 > $D/bench_extras.log 2> $D/bench_extras.err
python - <<PY
import json
l=[x for x in open("$D/bench_extras.log") if x.startswith("{")]
d=json.loads(l[0]); print(d["value"], d["e2e_cli"])
PY
