# A/B of run-time switches on ONE box, same build:  gpurun -- bash tools/ab_env.sh "<label>|<VAR=val ...>" ...   [CONFIG=<name> STEPS=<n>]
ulimit -c 0
D=gpurun_out/ab; mkdir -p $D
CONFIG=${CONFIG:-nc3500}; STEPS=${STEPS:-20}
for rep in $(seq 1 ${REPS:-2}); do
for spec in "$@"; do
  label="${spec%%|*}"; envs="${spec#*|}"
  env $envs timeout 600 python bench.py --config $CONFIG --cpu-baseline 0 --extras 0 --steps $STEPS --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); b = d['device_breakdown_ms']
        print('$label', 'analysing %.2f repeat %.2f | K9 %.3f ms/level frac %.4f | trf %.2f join %.2f schur %.2f pcg %.2f backsub %.2f small %.2f | resid %.1e nc %s' % (d['value'], d['resolve_ms'], d['kernels']['schur']['avg_launch_ms'], d['kernels']['schur'].get('frac_of_f64_mfma_peak', 0), b['t_transform_ms'], b['t_join_ms'], b['t_schur_ms'], b['t_pcg_ms'], b['t_backsub_ms'], b['t_small_ms'], d['max_rel_residual'], d['not_converged']))
"
done
done 2>&1 | tee $D/abenv_$(date +%H%M%S).txt
python - <<'PY'
import glob, re, collections
f = sorted(glob.glob("gpurun_out/ab/abenv_*.txt"))[-1]
acc = collections.defaultdict(list)
for l in open(f):
    m = re.match(r"(\S+) analysing ([\d.]+) repeat ([\d.]+)", l)
    if m: acc[m.group(1)].append((float(m.group(2)), float(m.group(3))))
for k, v in acc.items():
    a = sorted(x[0] for x in v); r = sorted(x[1] for x in v)
    print(f"MEDIAN {k}: analysing {a[len(a)//2]:.2f} (min {a[0]:.2f}) repeat {r[len(r)//2]:.2f} (min {r[0]:.2f}) over {len(v)}")
PY
