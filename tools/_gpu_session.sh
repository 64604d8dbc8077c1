timeout 300 python tools/experiments/f_sync_check.py 2>&1 | tail -2
