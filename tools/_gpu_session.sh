O=gpurun_out/r03n; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err; echo "bench rc $?"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
