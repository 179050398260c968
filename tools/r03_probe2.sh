ulimit -c 0
D=gpurun_out/r03b; mkdir -p $D
LSFM_DEBUG=1 timeout 300 python bench.py --plans --steps 1 --warmup 2 --cpu-baseline 0 --extras 0 > $D/debug_plans.json 2> $D/debug_plans.txt
timeout 900 python bench.py --config synth16k --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_synth16k.log 2> $D/bench_synth16k.err
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_FETCH_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/pmc_FETCH_SIZE.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_WRITE_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/pmc_WRITE_SIZE.log 2>&1
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r03b/bench_synth16k.log") if x.startswith("{")]
d=json.loads(l[0]); print("synth16k", d["value"], d["resolve_ms"], d["first_run_ms"], d["device_breakdown_ms"], d["max_rel_residual"], d["not_converged"])
PY
tail -40 $D/debug_plans.txt | cut -c1-260
ls $D/pmc_FETCH_SIZE | head
