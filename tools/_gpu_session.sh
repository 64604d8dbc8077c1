mkdir -p gpurun_out/r03l
timeout 900 python -m pytest tests/test_hip_synthesis.py tests/test_dropin_reference.py tests/test_c_abi_host.py -x -q -m gpu > gpurun_out/r03l/pytest_syn.log 2>&1; echo "pytest rc $?"
tail -15 gpurun_out/r03l/pytest_syn.log
