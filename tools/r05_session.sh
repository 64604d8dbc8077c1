#!/bin/bash
# Run ON THE GPU BOX (gpurun): round-5 validation of the shipped sources — GPU test suite, the driver's bench command, the small-view A/B
# (8 rays x 4 samples exact kernel at one wave per SIMD vs the 16 x 2 kernel), kernel statistics of the bench command.
#   usage: bash tools/r05_session.sh <tag> [full]
TAG=${1:-r05b}
O=gpurun_out/$TAG
mkdir -p $O
python -c "import panic3d_amd as P; assert not P._build.needs_build(), 'stale .so'" || exit 9
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
( timeout 100 python tools/bench_small_view.py 128 96; timeout 100 python tools/bench_small_view.py 128 48; timeout 100 python tools/bench_small_view.py 64 96 ) > $O/small_view.txt 2>&1; cat $O/small_view.txt
if [ "${2:-}" = full ]; then
  R=$(pwd); ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o r -- python $R/bench.py --no-cpu-baseline --no-verify --no-table --no-pipeline --roofline-steps 0 --steps 30 --warmup 3 > $R/$O/stats.json 2> $R/$O/stats.log )
  find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; head -6 $O/kernel_stats.csv
fi
