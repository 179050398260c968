# cold (plans-off) run of the NC3500-like tree: per-level debug timings, kernel / copy / HIP API statistics
D=gpurun_out/r03a; mkdir -p $D
LSFM_DEBUG=1 timeout 300 python bench.py --no-plans --steps 3 --warmup 1 --cpu-baseline 0 > $D/cold_bench.json 2> $D/cold_debug.txt
timeout 300 python bench.py --no-plans --steps 5 --warmup 1 --cpu-baseline 0 > $D/cold_bench_nodebug.json 2> /dev/null
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 500 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --stats --output-format csv -d $D/cold_stats -o run -- python3 bench.py --no-plans --steps 3 --warmup 1 --cpu-baseline 0 > $D/cold_prof.log 2>/dev/null
ls $D/cold_stats | head; tail -c 600 $D/cold_bench_nodebug.json | head -c 300
