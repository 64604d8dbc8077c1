set -x
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r03a/pytest.log
tail -15 gpurun_out/r03a/pytest.log
timeout 600 python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc $?"
tail -c 300 gpurun_out/r03a/bench.err
