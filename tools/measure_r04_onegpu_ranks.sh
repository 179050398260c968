# the one-GPU logic check of bench.py --gpus N (it starts its own ranks; all ranks on cuda:0 over gloo): n_gpus, per-rank times, the
# replicated share of the distributed factorisations.  Not a measurement of scaling.
ulimit -c 0
D=gpurun_out/${1:-r04f}; mkdir -p $D
export LSFM_BENCH_ONE_GPU=1
for cfg in nc3500 synth16k; do
for n in 2 8; do
timeout 1500 python bench.py --gpus $n --config $cfg --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 > $D/bench_onegpu${n}_$cfg.log 2> $D/bench_onegpu${n}_$cfg.err
python - <<PY
import json
try:
    l=[x for x in open("$D/bench_onegpu${n}_$cfg.log") if x.startswith("{")]
    d=json.loads(l[0]); print("$cfg $n", d["n_gpus"], round(d["value"],1), round(d["resolve_ms"],1), [round(v,1) for v in d["per_rank_device_ms"]], d["distributed_solve"], d["max_rel_residual"], d["not_converged"])
except Exception as e:
    print("$cfg $n ERR", e); print(open("$D/bench_onegpu${n}_$cfg.err").read()[-2500:])
PY
done; done
