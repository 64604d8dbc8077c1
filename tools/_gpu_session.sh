timeout 900 python -m pytest tests/test_hip_synthesis.py -x -q -m gpu -k "style_plan_memo" 2>&1 | tail -12
