timeout 900 python -m pytest tests/test_hip_synthesis.py tests/test_dropin_reference.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do timeout 300 python tools/profile_f.py 2>/dev/null | tail -1; done
timeout 300 python tools/experiments/f_sync_check.py 2>&1 | tail -1
