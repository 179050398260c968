# Everything profiles/<tag>_* is made from, in ONE gpurun call (about 45 GPU-minutes):
#     gpurun --timeout 3300 -- bash tools/measure.sh r06
# then, here:  python tools/collect_profiles.py gpurun_out/$TAG r06
# Parts: the GPU suite; the bench line of every configuration; rocprofv3 --kernel-trace --stats of the bench command (nc3500, synth16k,
# rs468); HBM traffic from SEPARATE --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the same command (never combined with a trace domain);
# run-to-run stability with LSFM_FACTOR_DIGEST=1; the multi-rank logic runs on ONE GPU over gloo (structure, not times); the Gauss-Newton
# polish; the phase clocks of K9 / k_tr_entries / k_sn_panel from a profiling build made ON THE BOX (nothing of it is committed).
ulimit -c 0
TAG=${1:-r06}
D=gpurun_out/$TAG; mkdir -p $D
timeout 1800 python -m pytest tests -x -q -m gpu --durations=10 > $D/gpu_tests.log 2>&1; tail -3 $D/gpu_tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_default.log 2> $D/bench_default.err   # (the driver's own command line)
timeout 300 python bench.py --config rs468 --steps 10 --warmup 2 > $D/bench_rs468.log 2>/dev/null
timeout 300 python bench.py --config rs90 --steps 10 --warmup 2 > $D/bench_rs90.log 2>/dev/null
timeout 300 python bench.py --config aerial --steps 10 --warmup 2 > $D/bench_aerial.log 2>/dev/null
timeout 600 python bench.py --config synth16k --steps 6 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_synth16k.log 2>/dev/null
timeout 600 python bench.py --config synth64k --maps 16384 --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_synth64k_16k.log 2>/dev/null
timeout 300 python tools/stage_compare.py nc3500 12 > $D/stage_compare_nc3500.txt 2>&1
for c in "nc3500 3" "rs468 4" "rs90 6"; do set -- $c; timeout 300 python tools/gn_bench.py $1 $2 > $D/gn_polish_$1.json 2>/dev/null; done
export LSFM_FACTOR_DIGEST=1
timeout 600 python tools/stability_16k.py 60 nc3500 > $D/stab.txt 2>&1
timeout 600 python tools/stability_16k.py 60 rs468 >> $D/stab.txt 2>&1
timeout 900 python tools/stability_16k.py 20 synth16k >> $D/stab.txt 2>&1
unset LSFM_FACTOR_DIGEST
export LSFM_BENCH_ONE_GPU=1
for spec in "nc3500 2" "nc3500 8" "synth16k 8"; do set -- $spec
timeout 1500 python bench.py --gpus $2 --config $1 --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_onegpu$2_$1.log 2> $D/bench_onegpu$2_$1.err
done
unset LSFM_BENCH_ONE_GPU
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in nc3500 synth16k rs468; do
  if [ $c = nc3500 ]; then a="--steps 5 --warmup 2"; elif [ $c = synth16k ]; then a="--steps 2 --warmup 1"; else a="--steps 5 --warmup 2"; fi
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/$c/stats -o run -- python3 bench.py --config $c $a --cpu-baseline 0 --extras 0 > $D/bench_prof_$c.log 2>/dev/null
  rm -f $D/$c/stats/*kernel_trace.csv $D/$c/stats/*/*kernel_trace.csv   # tens of MB: the per-kernel statistics are what is kept
  # (synth16k: one timed step -- the counter collection of a longer run of it crashed the profiler on this pool)
  if [ $c = synth16k ]; then ps=1; trees=5; else ps=2; trees=7; fi
  for m in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $m --output-format csv -d $D/$c/pmc_$m -o run -- python3 bench.py --config $c --steps $ps --warmup 1 --cpu-baseline 0 --extras 0 > $D/pmc_${m}_$c.log 2>&1
  done
  # the summaries are made HERE (trees in a --steps 2 --warmup 1 run: first + 1 warm-up + 2 timed + 3 of the other mode = 7; --steps 1: 5);
  # the raw per-dispatch counter files (5-10 MB each) travel back for nc3500 only
  if [ $c = nc3500 ]; then kr=1; else kr=0; fi
  PROFILES_DIR=$D/profiles KEEP_RAW=$kr python tools/refresh_profiles.py $D/$c $TAG $c $trees > $D/refresh_$c.txt 2>&1
  rm -rf $D/$c/pmc_FETCH_SIZE $D/$c/pmc_WRITE_SIZE
  if [ $c = nc3500 ]; then
    # the matrix cores' busy share per kernel (counters alone, like the passes above)
    timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $D/$c/pmc_mfma -o run -- python3 bench.py --config $c --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/pmc_mfma_$c.log 2>&1
    PROFILES_DIR=$D/profiles python tools/mfma_busy.py $(ls $D/$c/pmc_mfma/*counter_collection.csv $D/$c/pmc_mfma/*/*counter_collection.csv 2>/dev/null | head -1) $TAG $c > $D/mfma_busy_$c.txt 2>&1
    rm -rf $D/$c/pmc_mfma
  fi
done
touch linearsfm_amd/csrc/lsfm_pcg.hip linearsfm_amd/csrc/lsfm_schur_panel.hip linearsfm_amd/csrc/lsfm_transform.hip
make -s -C linearsfm_amd/csrc K9_TIMING=1 -j16 > $D/build_timing.log 2>&1
timeout 600 python tools/tr_phase_times.py > $D/tr_phase.txt 2>&1
timeout 600 python tools/k9_phase_times.py > $D/k9_phase.txt 2>&1
timeout 600 python tools/sn_phase_times.py nc3500 > $D/sn_phase_nc3500.txt 2>&1
python - <<PY
import json
for f in ("default","rs468","rs90","aerial","synth16k","synth64k_16k","prof_nc3500","prof_synth16k","prof_rs468"):
    try:
        l=[x for x in open("$D/bench_%s.log" % f) if x.startswith("{")]
        d=json.loads(l[0]); print(f, round(d["value"],2), round(d["resolve_ms"],2), round(d["first_run_ms"],1), round(d["roofline"]["frac"],4), d["max_rel_residual"], d["not_converged"], (d.get("cpu_baseline") or {}).get("pose_param_max_rel_err_vs_oracle"), (d.get("e2e_cli") or {}).get("e2e_cli_s"))
    except Exception as e: print(f, "ERR", e)
PY
grep -v "^Traceback\|^  File" $D/stab.txt | tail -12
