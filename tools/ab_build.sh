# A/B of build variants on ONE box:  gpurun -- bash tools/ab_build.sh "<label>|<make flags>" ...   (first entry usually "default|")
#   [CONFIG=<name> STEPS=<n> REPS=<n>]
# every variant: rebuild the library in place, 2 x `bench.py --steps 20 --warmup 3` (analysing / repeat ms, K9 ms per level, stage times)
ulimit -c 0
D=gpurun_out/ab; mkdir -p $D
for spec in "$@"; do
  label="${spec%%|*}"; flags="${spec#*|}"
  touch linearsfm_amd/csrc/lsfm_solve.hpp linearsfm_amd/csrc/lsfm_symbolic.hpp linearsfm_amd/csrc/lsfm_small.hip
  ( cd linearsfm_amd/csrc && make -s -j16 $flags > /dev/null 2>&1 ) || { echo "$label: build failed"; continue; }
  for rep in $(seq 1 ${REPS:-2}); do
    timeout 600 python bench.py --config ${CONFIG:-nc3500} ${MAPS:+--maps $MAPS} --cpu-baseline 0 --extras 0 --steps ${STEPS:-20} --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); b = d['device_breakdown_ms']
        print('$label', 'analysing %.2f repeat %.2f | K9 %.3f ms/level frac %.4f | trf %.2f join %.2f schur %.2f pcg %.2f backsub %.2f small %.2f | resid %.1e nc %s' % (d['value'], d['resolve_ms'], d['kernels']['schur']['avg_launch_ms'], d['kernels']['schur'].get('frac_of_f64_mfma_peak', 0), b['t_transform_ms'], b['t_join_ms'], b['t_schur_ms'], b['t_pcg_ms'], b['t_backsub_ms'], b['t_small_ms'], d['max_rel_residual'], d['not_converged']))
"
  done
done 2>&1 | tee $D/ab_$(date +%H%M%S).txt
# leave the default build behind
touch linearsfm_amd/csrc/lsfm_solve.hpp linearsfm_amd/csrc/lsfm_symbolic.hpp linearsfm_amd/csrc/lsfm_small.hip; ( cd linearsfm_amd/csrc && make -s -j16 > /dev/null 2>&1 )
