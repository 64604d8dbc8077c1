#!/bin/bash
# Run ON THE GPU BOX: the renderer's kernel statistics + PMC passes only (collect_profile.sh with the pipeline block off), summarised on the box.
TAG=${1:-r05f}
R=$(pwd); O=$R/gpurun_out/$TAG
mkdir -p $O/summary
python -c "import panic3d_amd as P; assert not P._build.needs_build(), 'stale .so'" || exit 9
bash tools/collect_profile.sh r05 "surface canonical" > $O/collect.log 2>&1
python tools/summarize_prof.py r05 > $O/summary/summarize_prof.txt 2>&1
cp profiles/r05_*kernel_stats.csv profiles/r05_pmc.json profiles/r05_bench_under_rocprof.json profiles/pmc_latest.json $O/summary/ 2>/dev/null
rm -rf gpurun_out/r05; find gpurun_out -size +4M -delete
du -sh gpurun_out; cat $O/summary/summarize_prof.txt | cut -c1-400
