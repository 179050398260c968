ulimit -c 0
D=gpurun_out/r03m; mkdir -p $D
timeout 900 python tools/random_parity_sweep.py 30 7 > $D/sweep.jsonl 2> $D/sweep.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r03m/sweep.jsonl") if l.startswith("{")]
bad=[r for r in rows if "case" in r and (r.get("error") or not r.get("same_structure") or r.get("rc") or r.get("state_max_rel_err",0)>1e-6)]
print(len(rows)-1, "cases; not clean:", len(bad)); 
for r in bad[:8]: print(r)
print(rows[-1])
PY
LSFM_TIMELINE=1 timeout 300 python bench.py --config rs468 --steps 2 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_tl.json 2> $D/timeline.txt
grep -n "^\[tl\]" $D/timeline.txt | sed -n 24,30p | cut -c1-420
