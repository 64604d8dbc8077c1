#!/bin/bash
# Run ON THE GPU BOX (gpurun): round-5 session 3 — GPU suite on the final renderer sources, bench, G.f host profile (plain / pasted view),
# the renderer's kernel statistics + PMC passes (tools/collect_profile.sh, surface scene) and the convolution launches' counters.
TAG=${1:-r05d}
R=$(pwd); O=$R/gpurun_out/$TAG
mkdir -p $O
python -c "import panic3d_amd as P; assert not P._build.needs_build(), 'stale .so'" || exit 9
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 120 python tools/host_profile.py > $O/host_profile.txt 2>&1; head -1 $O/host_profile.txt
timeout 120 python tools/host_profile.py --paste > $O/host_profile_paste.txt 2>&1; head -1 $O/host_profile_paste.txt
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/view_trace -o r -- python $R/tools/host_profile.py --paste > $O/view_trace.txt 2>&1 )
find $O/view_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/view_kernel_stats.csv
if [ "${2:-}" = full ]; then
  bash tools/collect_profile.sh r05 "surface" > $O/collect.log 2>&1; tail -5 $O/collect.log
  bash tools/pmc_backbone.sh r05 > $O/pmc_backbone.log 2>&1; tail -3 $O/pmc_backbone.log
fi
