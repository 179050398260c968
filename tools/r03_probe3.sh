ulimit -c 0
D=gpurun_out/r03c; mkdir -p $D
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats16k -o run -- python3 bench.py --config synth16k --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench16k_prof.log 2>/dev/null
LSFM_DEBUG=1 timeout 600 python bench.py --config synth16k --steps 1 --warmup 1 --cpu-baseline 0 --extras 0 > $D/dbg16k.json 2> $D/dbg16k.txt
grep "solve M=" $D/dbg16k.txt | tail -16 | cut -c1-250
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r03c/stats16k/run_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    n=r["Name"].replace("void ","").replace("lsfm::","").split("(")[0][:50]
    print(f"{n:50s} calls {int(r['Calls']):6d} total ms {int(r['TotalDurationNs'])/1e6:9.2f} avg us {float(r['AverageNs'])/1e3:9.1f} {100*int(r['TotalDurationNs'])/tot:5.1f}%")
PY
