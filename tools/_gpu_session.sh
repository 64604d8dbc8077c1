O=gpurun_out/r03j; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_synthesis.py tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -3
python tools/graph_backbone.py 2>/dev/null | tail -1
python tools/profile_f.py 2>/dev/null | tail -1
python tools/generate_subject.py 2>/dev/null | tail -3
