mkdir -p gpurun_out/r03k
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03k/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r03k/pytest.log
bash tools/collect_profile.sh r03 > gpurun_out/r03k/collect.log 2>&1; echo "collect rc $?"
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/r03k/bench.json 2> gpurun_out/r03k/bench.err; echo "bench rc $?"
timeout 600 python bench.py --fast > gpurun_out/r03k/bench_fast.json 2> gpurun_out/r03k/bench_fast.err; echo "bench fast rc $?"
timeout 600 python bench.py --scene surface > gpurun_out/r03k/bench_surface.json 2> gpurun_out/r03k/bench_surface.err; echo "bench surface rc $?"
