# round 3 validation on the GPU: the whole -m gpu suite (new configuration tests included), then the default bench line
D=gpurun_out/r03c; mkdir -p $D
timeout 2400 python -m pytest tests -q -m gpu --durations=15 > $D/gpu_tests.log 2>&1; tail -40 $D/gpu_tests.log
