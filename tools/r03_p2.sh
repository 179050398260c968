ulimit -c 0
D=gpurun_out/r03p2; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_sharded.py -x -q --durations=20 > $D/sharded.log 2>&1; tail -40 $D/sharded.log
