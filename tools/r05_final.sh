#!/bin/bash
# Run ON THE GPU BOX (gpurun): the round's closing validation of the shipped sources — GPU suite, smoke, the driver's bench command
# (default arguments and the short form), counters of the 128^2-ray launch; summaries made on the box, raw rocprofv3 trees pruned.
TAG=${1:-r05final}
R=$(pwd); O=$R/gpurun_out/$TAG
mkdir -p $O/summary
python -c "import panic3d_amd as P; assert not P._build.needs_build(), 'stale .so'" || exit 9
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 ) > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench (defaults) rc $?"; grep real $O/bench_default.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
bash tools/pmc_small_view.sh r05 > $O/pmc_small.log 2>&1
python tools/summarize_small_pmc.py r05 > $O/summary/summarize_small_pmc.txt 2>&1
cp profiles/r05_small_view_pmc.json $O/summary/ 2>/dev/null
rm -rf gpurun_out/r05_smallpmc; find gpurun_out -size +4M -delete
du -sh gpurun_out; tail -5 $O/summary/summarize_small_pmc.txt | cut -c1-300
