mkdir -p gpurun_out/r03r
timeout 600 python bench.py > gpurun_out/r03r/bench.json 2> gpurun_out/r03r/bench.err; echo "bench rc $?"
timeout 600 python bench.py --fast > gpurun_out/r03r/bench_fast.json 2> gpurun_out/r03r/bench_fast.err; echo "bench fast rc $?"
timeout 600 python bench.py --scene surface > gpurun_out/r03r/bench_surface.json 2> gpurun_out/r03r/bench_surface.err; echo "bench surface rc $?"
