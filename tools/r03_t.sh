ulimit -c 0
D=gpurun_out/r03h; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --config synth16k --plans --steps 4 --warmup 1 --cpu-baseline 0 --extras 0 > $D/b16k.log 2>/dev/null
timeout 300 python bench.py --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bnc.log 2>/dev/null
python - <<PY
import json
for f in ("b16k","bnc"):
    l=[x for x in open("$D/"+f+".log") if x.startswith("{")]
    d=json.loads(l[0]); print(f, round(d["value"],2), round(d["analysing_run_ms"],2), round(d["resolve_ms"],2), {k:round(v,2) for k,v in d["device_breakdown_ms"].items()}, d["max_rel_residual"], d["not_converged"])
PY
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats16k -o run -- python3 bench.py --config synth16k --plans --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench16k_prof.log 2>/dev/null
grep "k_sn_syrk\|k_sn_panel\|k_tr_entries" $D/stats16k/run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,200-400
