# quick loop for the factorisation work: parity subset, bench lines, then the phase clocks of the panel kernel (profiling build on the box)
ulimit -c 0
D=gpurun_out/${1:-r04c}; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or dense or tree_stereo or tree_mono or mixed or inverse" > $D/gpu_tests.log 2>&1; tail -3 $D/gpu_tests.log
timeout 500 python bench.py --cpu-baseline 0 --extras 0 --steps 10 > $D/bench_default.log 2> $D/bench_default.err
timeout 600 python bench.py --config synth16k --steps 3 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_synth16k.log 2>/dev/null
timeout 300 python bench.py --config rs468 --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 > $D/bench_rs468.log 2>/dev/null
python - <<PY
import json
for f in ("default","synth16k","rs468"):
    try:
        l=[x for x in open("$D/bench_%s.log" % f) if x.startswith("{")]
        d=json.loads(l[0]); print(f, round(d["value"],2), round(d["resolve_ms"],2), {k: round(v,2) for k,v in d["device_breakdown_ms"].items()}, d["max_rel_residual"], d["not_converged"], d["pcg_iterations_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
touch linearsfm_amd/csrc/lsfm_pcg.hip linearsfm_amd/csrc/lsfm_schur_panel.hip linearsfm_amd/csrc/lsfm_transform.hip
make -s -C linearsfm_amd/csrc K9_TIMING=1 -j8 > $D/build.log 2>&1
timeout 600 python tools/sn_phase_times.py nc3500 > $D/sn_phase_nc3500.txt 2>&1
cat $D/sn_phase_nc3500.txt
