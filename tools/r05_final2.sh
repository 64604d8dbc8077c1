#!/bin/bash
# Run ON THE GPU BOX (gpurun): closing validation of the FINAL renderer sources — GPU suite, smoke, the renderer's kernel statistics + PMC
# passes (both scenes; summarised on the box), then the driver's bench command (which then finds the capture of these sources).
TAG=${1:-r05final2}
R=$(pwd); O=$R/gpurun_out/$TAG
mkdir -p $O/summary
python -c "import panic3d_amd as P; assert not P._build.needs_build(), 'stale .so'" || exit 9
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
bash tools/collect_profile.sh r05 "surface canonical" > $O/collect.log 2>&1
python tools/summarize_prof.py r05 > $O/summary/summarize_prof.txt 2>&1
cp profiles/r05_*kernel_stats.csv profiles/r05_pmc.json profiles/r05_bench_under_rocprof.json profiles/pmc_latest.json $O/summary/ 2>/dev/null
rm -rf gpurun_out/r05; find gpurun_out -size +4M -delete
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench (defaults) rc $?"; grep real $O/bench_default.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
du -sh gpurun_out; cut -c1-200 $O/summary/summarize_prof.txt
