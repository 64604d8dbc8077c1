mkdir -p gpurun_out/r03n
timeout 900 python -m pytest tests/test_hip_synthesis.py tests/test_dropin_reference.py -x -q -m gpu > gpurun_out/r03n/pytest_syn.log 2>&1; echo "pytest rc $?"
tail -4 gpurun_out/r03n/pytest_syn.log
for i in 1 2; do timeout 300 python tools/profile_f.py 2>/dev/null | tail -1; done
timeout 300 python tools/generate_subject.py 2>/dev/null | tail -1
