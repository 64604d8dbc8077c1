#!/bin/bash
# Run ON THE GPU BOX (gpurun): the round's closing validation of the shipped sources.
#   part 1 (always): the GPU test suite, the driver's bench command, the secondary benchmarks, the host profile of a G.f call and a
#                    kernel trace of the same calls (kernel time per view).
#   part 2 (FULL=1): per-layer convolution times, per-launch counters of a backbone / super-resolution pass, the renderer's kernel
#                    statistics + PMC passes — only needed when csrc/ changed (profiles are keyed on the source hashes).
# Condense afterwards (build container): python tools/summarize_prof.py r04; python tools/summarize_conv_pmc.py r04
python -c "import panic3d_amd as P; assert not P._build.needs_build()" || exit 9
O=gpurun_out/r04final
mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/pytest_gpu.txt 2>&1
cat $O/pytest_gpu.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 900 bash tools/secondary_benchmarks.sh > $O/secondary_benchmarks.txt 2>&1
timeout 120 python tools/host_profile.py > $O/host_profile.txt 2>&1
timeout 120 python tools/host_profile.py --paste > $O/host_profile_paste.txt 2>&1
timeout 200 python tools/graph_backbone.py > $O/passes.txt 2>/dev/null; cat $O/passes.txt
R=$(pwd); ( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/view_trace -o r -- python $R/tools/host_profile.py --paste > $R/$O/view_trace.txt 2>&1 )
find $O/view_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/view_kernel_stats.csv
head -3 $O/host_profile.txt | tail -2; head -3 $O/host_profile_paste.txt | tail -2
if [ "${FULL:-0}" = 1 ]; then
  timeout 200 python tools/conv_layers_time.py --n 30 > $O/conv_layers.txt 2>/dev/null
  bash tools/pmc_backbone.sh r04 > $O/pmc_backbone.log 2>&1
  bash tools/collect_profile.sh r04 > $O/collect.log 2>&1
  tail -40 $O/collect.log
fi
