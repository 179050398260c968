# Where the main stream of an analysing run idles: rocprofv3 kernel trace of bench.py, per tree level the wall time, the time the main
# stream is busy and the largest gaps.  usage (GPU box):  bash tools/level_gaps.sh [tag]   (LSFM_NO_WORKER=1 in the environment: the
# symbolic analysis on the enqueuing thread, as before the helper thread) -> gpurun_out/level_gaps_<tag>.txt
ulimit -c 0
TAG=${1:-worker}
D=gpurun_out/gaps_$TAG; mkdir -p $D
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D/trace -o run -- python3 bench.py --steps 3 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench.log 2>/dev/null
python - "$D" > gpurun_out/level_gaps_$TAG.txt <<'PY'
import csv, glob, sys
D=sys.argv[1]
f=glob.glob(D+"/trace/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
ev=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].replace("void lsfm::","").replace("lsfm::","").split("(")[0][:26],r.get("Stream_Id","?")) for r in rows]
ev.sort()
idx=[i for i,e in enumerate(ev) if e[2].startswith("k_tr_find")]
main=max(set(e[3] for e in ev), key=lambda s: sum(1 for e in ev if e[3]==s))
def union(iv):
    iv=sorted(iv); tot=0; cs,ce=iv[0]
    for s,e in iv[1:]:
        if s>ce: tot+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    return tot+ce-cs
print("# bench.py --steps 3 --warmup 2 under rocprofv3 --kernel-trace (the profiler slows the host: gaps are larger than in an unprofiled run)")
for t in (3,4):   # timed analysing trees
    lo=idx[13*t]; hi=idx[13*(t+1)]
    print("tree", t, "wall ms %.2f" % ((ev[hi-1][1]-ev[lo][0])/1e6))
    for L in range(13):
        a=idx[13*t+L]; b=idx[13*t+L+1]
        seg=ev[a:b]; m=[e for e in seg if e[3]==main]
        wall=(seg[-1][1]-seg[0][0])/1e6
        busy=union([(s,e) for s,e,_,_ in m])/1e6
        gaps=[]; ce=m[0][1]
        for s,e,n,q in m[1:]:
            if s>ce+20000: gaps.append(((s-ce)/1e3,n))
            ce=max(ce,e)
        gaps.sort(reverse=True)
        print(f" level {L:2d}: wall {wall:6.2f} ms, main stream busy {busy:6.2f}, idle {wall-busy:5.2f}, {len(m):4d} launches | gaps > 20 us before: "+", ".join(f"{g:.0f}->{n}" for g,n in gaps[:4]))
PY
cat gpurun_out/level_gaps_$TAG.txt | head -16
