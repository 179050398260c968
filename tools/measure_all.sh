# Everything the profiles/ directory is refreshed from, in one gpurun call:  gpurun --timeout 2400 -- bash tools/measure_all.sh
# then:  python tools/refresh_profiles.py gpurun_out/r02u r02 nc3500 3   (3 = trees in a --steps 2 --warmup 1 PMC run)
D=gpurun_out/r02u; mkdir -p $D
timeout 900 python -m pytest tests -x -q -m gpu > $D/gpu_tests.log 2>&1; tail -2 $D/gpu_tests.log
timeout 400 python bench.py > $D/bench_default.log 2>/dev/null
timeout 300 python bench.py --config rs468 --steps 10 --warmup 2 > $D/bench_rs468.log 2>/dev/null
timeout 300 python bench.py --config rs90 --steps 10 --warmup 2 > $D/bench_rs90.log 2>/dev/null
timeout 300 python bench.py --mixed --steps 10 --warmup 2 --cpu-baseline 0 > $D/bench_mixed.log 2>/dev/null
timeout 300 python bench.py --config spmv-stream --steps 20 --warmup 3 > $D/bench_spmv_stream.log 2>/dev/null
timeout 300 python tools/spmv_bench.py > $D/spmv_bench.jsonl 2>/dev/null
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o run -- python3 bench.py --steps 5 --warmup 2 --cpu-baseline 0 > $D/bench_prof.log 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_FETCH_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 > $D/pmc_FETCH_SIZE.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_WRITE_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 > $D/pmc_WRITE_SIZE.log 2>&1
timeout 600 python tools/full_parity.py nc3500 > $D/full_parity_nc3500.json 2> $D/full_parity.err
timeout 600 python tools/full_parity.py rs468 > $D/full_parity_rs468.json 2>> $D/full_parity.err
for f in default rs468 rs90 mixed spmv_stream; do python -c "
import json,sys
l=[x for x in open('$D/bench_$f.log') if x.startswith('{')]
d=json.loads(l[0]); print('$f', d['value'], d.get('first_run_ms'), d['roofline']['frac'], (d.get('cpu_baseline') or {}).get('pose_param_max_rel_err_vs_oracle'))"; done
ls $D $D/stats $D/pmc_FETCH_SIZE | head -30
