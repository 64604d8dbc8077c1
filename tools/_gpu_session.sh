mkdir -p gpurun_out/r03n
bash tools/secondary_benchmarks.sh > gpurun_out/r03n/secondary.txt 2>&1; echo "secondary rc $?"
