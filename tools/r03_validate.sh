ulimit -c 0
D=gpurun_out/r03q; mkdir -p $D
b() { timeout 300 python bench.py "$@" --steps 10 --warmup 2 --cpu-baseline 0 --extras 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('value', round(d['value'],2), 'resolve', round(d['resolve_ms'],2), {k:round(v,2) for k,v in d['device_breakdown_ms'].items()}, d['pcg_iterations_per_step'], 'K9', round(d['kernels']['schur']['ms_per_step'],2))"; }
echo "nc3500 PF6=4"; b
echo "rs468"; b --config rs468
touch linearsfm_amd/csrc/lsfm_schur_panel.hip; make -s -C linearsfm_amd/csrc K9_PF6=3 2>&1 | tail -2
echo "nc3500 PF6=3"; b
echo "nc3500 PF6=3 plans"; b --plans
touch linearsfm_amd/csrc/lsfm_schur_panel.hip; make -s -C linearsfm_amd/csrc K9_OCC16=2 2>&1 | tail -2
echo "nc3500 OCC2 plans"; b --plans
