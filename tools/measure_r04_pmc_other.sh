# HBM traffic (PMC, separate passes) of a synth-16k tree and kernel stats of the aerial block -- the configurations beside NC3500:
#   gpurun --timeout 2400 -- bash tools/measure_r04_pmc_other.sh ; python tools/refresh_profiles.py gpurun_out/r04s r04 synth16k 3
ulimit -c 0
D=gpurun_out/r04s; mkdir -p $D
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_FETCH_SIZE -o run -- python3 bench.py --config synth16k --steps 1 --warmup 1 --cpu-baseline 0 --extras 0 > $D/pmc_FETCH_SIZE.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_WRITE_SIZE -o run -- python3 bench.py --config synth16k --steps 1 --warmup 1 --cpu-baseline 0 --extras 0 > $D/pmc_WRITE_SIZE.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_aerial -o run -- python3 bench.py --config aerial --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_prof_aerial.log 2>/dev/null
ls $D $D/pmc_FETCH_SIZE | head -20
grep -c k_backsub $D/pmc_FETCH_SIZE/run_counter_collection.csv
