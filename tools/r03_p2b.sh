ulimit -c 0
D=gpurun_out/r03p2; mkdir -p $D
export LSFM_BENCH_ONE_GPU=1
for top in shard merge; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 3 --warmup 1 --top $top > $D/bench_onegpu4_$top.log 2> $D/bench_onegpu4_$top.err
tail -3 $D/bench_onegpu4_$top.err
python - <<PY
import json
l=[x for x in open("$D/bench_onegpu4_$top.log") if x.startswith("{")]
d=json.loads(l[0]); print("$top", d["value"], d["resolve_ms"], d["first_run_ms"], d["per_rank_device_ms"], d["rank0_phases_ms"], d["max_rel_residual"], d["not_converged"], d["config"]["sharding"][:80])
PY
done
