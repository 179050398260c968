ulimit -c 0
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
for c in rs468 synth16k nc3500; do
  if [ $c = synth16k ]; then st=4; else st=12; fi
  timeout 600 python bench.py --config $c --steps $st --warmup 2 --cpu-baseline 0 --extras 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); b = d['device_breakdown_ms']
        print('$c', 'analysing %.2f repeat %.2f | trf %.2f join %.2f schur %.2f pcg %.2f' % (d['value'], d['resolve_ms'], b['t_transform_ms'], b['t_join_ms'], b['t_schur_ms'], b['t_pcg_ms']))
"
done
