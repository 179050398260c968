ulimit -c 0
D=gpurun_out/r03s; mkdir -p $D
LSFM_TIMELINE=1 timeout 300 python bench.py --steps 2 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_tl.json 2> $D/timeline.txt
grep -n "^\[tl\]" $D/timeline.txt | tail -14 | cut -c1-400
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D/trace_cold -o run -- python3 bench.py --steps 3 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_prof_cold.log 2>/dev/null
python - <<'PY'
import csv, glob, collections
f=glob.glob("gpurun_out/r03s/trace_cold/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
mc=glob.glob("gpurun_out/r03s/trace_cold/*memory_copy_trace.csv")
cp=list(csv.DictReader(open(mc[0]))) if mc else []
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("void lsfm::","").replace("lsfm::","").split("(")[0][:30],r.get("Queue_Id","?")) for r in rows]
ev+= [(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY_"+r.get("Direction","")[:12],"c") for r in cp]
ev.sort()
idx=[i for i,e in enumerate(ev) if e[2].startswith("k_tr_find")]
print("transform starts", len(idx))
# tree t (13 transforms per tree); pick the 4th tree = an analysing timed step
t=3
lo,hi=idx[13*t], idx[13*(t+1)] if 13*(t+1)<len(idx) else len(ev)
t0=ev[lo][0]
print("tree wall ms", (ev[hi-1][1]-t0)/1e6)
# busy time (union of intervals) and per-level wall / busy
def union(iv):
    iv=sorted(iv); tot=0; cs,ce=iv[0]
    for s,e in iv[1:]:
        if s>ce: tot+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    return tot+ce-cs
for L in range(13):
    a=idx[13*t+L]; b=idx[13*t+L+1] if 13*t+L+1<len(idx) else hi
    seg=ev[a:b]
    wall=(seg[-1][1]-seg[0][0])/1e6
    busy=union([(s,e) for s,e,_,_ in seg])/1e6
    # largest idle gaps
    gaps=[]; ce=seg[0][1]
    for s,e,n,q in seg[1:]:
        if s>ce: gaps.append(((s-ce)/1e3,n))
        ce=max(ce,e)
    gaps.sort(reverse=True)
    print(f"level {L:2d}: wall {wall:6.2f} ms busy {busy:6.2f} ms launches {len(seg):4d}  top gaps(us): "+", ".join(f"{g:.0f}->{n}" for g,n in gaps[:4]))
PY
