O=gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_mcubes.py tests/test_hip_bench_contract.py -m gpu -q -x 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
