mkdir -p gpurun_out/r03t
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03t/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r03t/pytest.log
bash tools/secondary_benchmarks.sh > gpurun_out/r03t/secondary.txt 2>&1; echo "secondary rc $?"
timeout 300 python tools/graph_backbone.py 2>&1 | tail -1
