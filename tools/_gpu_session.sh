O=gpurun_out/r03m; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/pmc_backbone.sh r03 > $O/pmc_backbone.log 2>&1
bash tools/collect_profile.sh r03 > $O/collect.log 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 600 python bench.py --fast --no-cpu-baseline --no-table > $O/bench_fast.json 2>> $O/bench.err
timeout 600 python bench.py --scene surface --no-cpu-baseline --no-table > $O/bench_surface.json 2>> $O/bench.err
bash tools/secondary_benchmarks.sh > $O/secondary.txt 2>&1
for c in "--crop 0.1" "--crop 0.1 --fast" "--fast"; do echo "== bench_c5 $c"; timeout 300 python tools/bench_c5.py $c 2>/dev/null | tail -1; done >> $O/secondary.txt
echo "== sweep360 --exact"; timeout 300 python tools/sweep360.py --exact 2>/dev/null | tail -1 >> $O/secondary.txt
