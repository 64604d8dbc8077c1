timeout 600 python -m pytest tests/test_hip_synthesis.py -x -q -m gpu -k "prepared_conditioning" 2>&1 | grep -v "^$" | tail -30
