# development aid: the multi-rank logic of bench.py on ONE GPU (all ranks on cuda:0, gloo): 2 and 8 ranks, both top modes
ulimit -c 0
D=gpurun_out/r03p2; mkdir -p $D
export LSFM_BENCH_ONE_GPU=1
for n in 2 8; do
for top in shard merge; do
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus $n --steps 2 --warmup 1 --top $top > $D/bench_onegpu${n}_$top.log 2> $D/bench_onegpu${n}_$top.err
python - <<PY
import json
try:
    l=[x for x in open("$D/bench_onegpu${n}_$top.log") if x.startswith("{")]
    d=json.loads(l[0]); print("$n $top", round(d["value"],1), round(d["resolve_ms"],1), [round(v,1) for v in d["per_rank_device_ms"]], d["per_rank_phases_ms"], d["max_rel_residual"], d["not_converged"], d["config"]["workload"][-60:])
except Exception as e:
    print("$n $top ERR", e); print(open("$D/bench_onegpu${n}_$top.err").read()[-1500:])
PY
done; done
