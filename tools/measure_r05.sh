# round 5: everything profiles/r05_* is refreshed from, in one gpurun call:  gpurun --timeout 3300 -- bash tools/measure_r05.sh
# then:  python tools/refresh_profiles.py gpurun_out/r05u r05 nc3500 7   (7 = trees in a --steps 2 --warmup 1 PMC run)
ulimit -c 0
D=gpurun_out/r05fin; mkdir -p $D
timeout 1800 python -m pytest tests -x -q -m gpu --durations=10 > $D/gpu_tests.log 2>&1; tail -3 $D/gpu_tests.log
timeout 500 python bench.py > $D/bench_default.log 2> $D/bench_default.err
timeout 300 python bench.py --config rs468 --steps 10 --warmup 2 > $D/bench_rs468.log 2>/dev/null
timeout 300 python bench.py --config rs90 --steps 10 --warmup 2 > $D/bench_rs90.log 2>/dev/null
timeout 300 python bench.py --config aerial --steps 10 --warmup 2 > $D/bench_aerial.log 2>/dev/null
timeout 600 python bench.py --config synth16k --steps 6 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_synth16k.log 2>/dev/null
timeout 600 python bench.py --config synth64k --maps 16384 --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_synth64k_16k.log 2>/dev/null
export LSFM_FACTOR_DIGEST=1
timeout 600 python tools/stability_16k.py 100 nc3500 > $D/stab.txt 2>&1
timeout 600 python tools/stability_16k.py 100 rs468 >> $D/stab.txt 2>&1
timeout 1200 python tools/stability_16k.py 40 synth16k >> $D/stab.txt 2>&1
unset LSFM_FACTOR_DIGEST
timeout 300 python tools/small_levels.py nc3500 4 analysing 5 > $D/small_levels.txt 2>&1
timeout 300 python tools/small_levels.py nc3500 4 analysing 16 >> $D/small_levels.txt 2>&1
timeout 300 python tools/small_levels.py rs468 3 analysing 5 >> $D/small_levels.txt 2>&1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o run -- python3 bench.py --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_prof.log 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_synth16k -o run -- python3 bench.py --config synth16k --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_prof_synth16k.log 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats_rs468 -o run -- python3 bench.py --config rs468 --steps 5 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_prof_rs468.log 2>/dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_FETCH_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/pmc_FETCH_SIZE.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_WRITE_SIZE -o run -- python3 bench.py --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/pmc_WRITE_SIZE.log 2>&1
# the kernel traces are tens of MB: the per-kernel statistics are what is kept
rm -f $D/stats*/run_kernel_trace.csv
python - <<'PY'
import json
for f in ("default","rs468","rs90","aerial","synth16k","synth64k_16k","prof","prof_synth16k","prof_rs468"):
    try:
        l=[x for x in open(f"gpurun_out/r05fin/bench_{f}.log") if x.startswith("{")]
        d=json.loads(l[0]); print(f, round(d["value"],2), round(d["resolve_ms"],2), round(d["first_run_ms"],1), round(d["roofline"]["frac"],4), d["max_rel_residual"], d["not_converged"], (d.get("cpu_baseline") or {}).get("pose_param_max_rel_err_vs_oracle"), (d.get("e2e_cli") or {}).get("e2e_cli_s"))
    except Exception as e: print(f, "ERR", e)
PY
cat $D/stab.txt | grep -v "^Traceback\|^  File"
cat $D/small_levels.txt
