ulimit -c 0
D=gpurun_out/${1:-r04p}; mkdir -p $D
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o run -- python3 bench.py --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_prof.log 2>/dev/null
ls $D/stats/*/ 2>/dev/null | head; find $D/stats -name "*kernel_stats.csv" | head -2
