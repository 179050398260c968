ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_cli.py tests/test_gpu_configs.py -x -q -m gpu -k "cli or aerial" 2>&1 | tail -1
for i in 1 2 3 4 5; do
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "aerial or synth16k" -rP 2>&1 | grep -a "attempts per run\|passed\|failed\|AssertionError: {" | cut -c1-300
done
