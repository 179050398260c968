# round 3, first validation on the GPU: the whole -m gpu suite (new configuration tests included), then the default bench line
D=gpurun_out/r03b; mkdir -p $D
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > $D/gpu_tests.log 2>&1; tail -25 $D/gpu_tests.log
timeout 900 python bench.py > $D/bench_default.log 2> $D/bench_default.err; tail -c 1500 $D/bench_default.log; tail -5 $D/bench_default.err
