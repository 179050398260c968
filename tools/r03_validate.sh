ulimit -c 0
D=gpurun_out/r03s; mkdir -p $D
LSFM_TIMELINE=1 timeout 300 python bench.py --steps 2 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_tl.json 2> $D/timeline.txt
sed -n 41,54p $D/timeline.txt | cut -c1-300
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D/trace_cold -o run -- python3 bench.py --steps 3 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_prof_cold.log 2>/dev/null
python - <<'PY'
import csv, glob
f=glob.glob("gpurun_out/r03s/trace_cold/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
print(len(rows), rows[0].keys())
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# take the last 30 ms window before the final analysing run ends: find last k_tr_entries sequences
import collections
names=[r["Kernel_Name"].replace("void lsfm::","").split("(")[0][:28] for r in rows]
# dump queue ids and a compact timeline of one full cold tree: find indices of k_tr_find (level starts)
idx=[i for i,n in enumerate(names) if n.startswith("k_tr_find")]
print("levels found", len(idx))
# choose the 3rd tree's levels (13 transforms per tree): tree t covers idx[13*t : 13*(t+1)]
t=3
lo, hi = idx[13*t+9], idx[13*t+11]
t0=int(rows[lo]["Start_Timestamp"])
for r,n in zip(rows[lo:hi], names[lo:hi]):
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print(f"{s:9.1f} {e-s:8.1f} q{r.get('Queue_Id','?')} {n}")
PY
