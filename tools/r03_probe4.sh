ulimit -c 0
D=gpurun_out/r03e; mkdir -p $D
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats16k -o run -- python3 bench.py --config synth16k --plans --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench16k_prof.log 2>/dev/null
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r03e/stats16k/run_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    n=r["Name"].replace("void ","").replace("lsfm::","").split("(")[0][:50]
    print(f"{n:50s} calls {int(r['Calls']):6d} total ms {int(r['TotalDurationNs'])/1e6:9.2f} avg us {float(r['AverageNs'])/1e3:9.1f} max us {int(r['MaxNs'])/1e3:9.1f} {100*int(r['TotalDurationNs'])/tot:5.1f}%")
# the syrk launches of one tree, by duration
tr=list(csv.DictReader(open("gpurun_out/r03e/stats16k/run_kernel_trace.csv")))
sy=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r["Grid_Size_X"] if "Grid_Size_X" in r else "", r.get("Grid_Size_Y","")) for r in tr if "k_sn_syrk" in r["Kernel_Name"]]
sy.sort(reverse=True)
print("syrk launches", len(sy), "top:", [(round(d/1e3),gx,gy) for d,gx,gy in sy[:12]])
print(list(tr[0].keys()))
PY
