mkdir -p gpurun_out/r03s
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03s/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r03s/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/secondary_benchmarks.sh > gpurun_out/r03s/secondary.txt 2>&1; echo "secondary rc $?"
