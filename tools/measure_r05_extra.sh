# round 5, second call: the one-GPU logic runs of bench.py --gpus N (all ranks on cuda:0 over gloo: structure, not times) and the phase
# clocks of k_tr_entries / K9 / k_sn_panel (profiling build made ON THE BOX; nothing of it is committed)
#   gpurun --timeout 2400 -- bash tools/measure_r05_extra.sh  ->  profiles/r05_bench_onegpu*_logic_*.json, r05_{tr,k9,sn_panel}_phase_times.txt
ulimit -c 0
D=gpurun_out/r05x; mkdir -p $D
export LSFM_BENCH_ONE_GPU=1
for spec in "nc3500 2" "nc3500 8" "synth16k 8"; do
set -- $spec; cfg=$1; n=$2
timeout 1500 python bench.py --gpus $n --config $cfg --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_onegpu${n}_$cfg.log 2> $D/bench_onegpu${n}_$cfg.err
python - <<PY
import json
try:
    l=[x for x in open("$D/bench_onegpu${n}_$cfg.log") if x.startswith("{")]
    d=json.loads(l[0]); print("$cfg $n", d["n_gpus"], round(d["value"],1), round(d["resolve_ms"],1), [round(v,1) for v in d["per_rank_device_ms"]], d["distributed_solve"], d["max_rel_residual"], d["not_converged"])
except Exception as e:
    print("$cfg $n ERR", e); print(open("$D/bench_onegpu${n}_$cfg.err").read()[-2500:])
PY
done
unset LSFM_BENCH_ONE_GPU
touch linearsfm_amd/csrc/lsfm_pcg.hip linearsfm_amd/csrc/lsfm_schur_panel.hip linearsfm_amd/csrc/lsfm_transform.hip
make -s -C linearsfm_amd/csrc K9_TIMING=1 -j8 > $D/build.log 2>&1
timeout 600 python tools/tr_phase_times.py > $D/tr_phase.txt 2>&1
timeout 600 python tools/k9_phase_times.py > $D/k9_phase.txt 2>&1
timeout 600 python tools/sn_phase_times.py nc3500 > $D/sn_phase_nc3500.txt 2>&1
cat $D/tr_phase.txt $D/k9_phase.txt $D/sn_phase_nc3500.txt
