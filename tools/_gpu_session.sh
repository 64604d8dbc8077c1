mkdir -p gpurun_out/r03r
bash tools/secondary_benchmarks.sh > gpurun_out/r03r/secondary.txt 2>&1; echo "secondary rc $?"
timeout 300 python tools/graph_backbone.py 2>&1 | tail -1
