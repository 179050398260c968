ulimit -c 0
timeout 600 python tools/k9_phase_times.py nc3500 2>&1 | tail -40
