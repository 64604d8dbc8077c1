mkdir -p gpurun_out/r03m
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03m/pytest.log 2>&1; echo "pytest rc $?"
tail -3 gpurun_out/r03m/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/collect_profile.sh r03 > gpurun_out/r03m/collect.log 2>&1; echo "collect rc $?"
cd $GRAFT_REPO_ROOT
bash tools/pmc_backbone.sh r03 > gpurun_out/r03m/pmc_backbone.log 2>&1; echo "pmc_backbone rc $?"
cd $GRAFT_REPO_ROOT
bash tools/secondary_benchmarks.sh > gpurun_out/r03m/secondary.txt 2>&1; echo "secondary rc $?"
