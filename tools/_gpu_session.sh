O=gpurun_out/r03i; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; tail -4 $O/pytest.log; grep "fast-vs-oracle\|tolerance vs exact" $O/pytest.log | sort | tail -40
