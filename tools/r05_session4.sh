#!/bin/bash
# Run ON THE GPU BOX (gpurun): bench, G.f host profile (plain / pasted view) + its kernel statistics, the renderer's kernel statistics +
# PMC passes (tools/collect_profile.sh, surface scene), the convolution launches' counters — SUMMARISED ON THE BOX (the raw rocprofv3
# trees can exceed the 64 MiB that gpurun copies back: session 3 of this round lost its outputs that way), raw trees pruned.
TAG=${1:-r05e}
R=$(pwd); O=$R/gpurun_out/$TAG
mkdir -p $O
python -c "import panic3d_amd as P; assert not P._build.needs_build(), 'stale .so'" || exit 9
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 120 python tools/host_profile.py > $O/host_profile.txt 2>&1; grep -m1 ms_per_call_synced $O/host_profile.txt
timeout 120 python tools/host_profile.py --paste > $O/host_profile_paste.txt 2>&1; grep -m1 ms_per_call_synced $O/host_profile_paste.txt
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/view_trace -o r -- python $R/tools/host_profile.py --paste > $O/view_trace.txt 2>&1 )
find /tmp/view_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/view_kernel_stats.csv
bash tools/collect_profile.sh r05 "surface" > $O/collect.log 2>&1
bash tools/pmc_backbone.sh r05 > $O/pmc_backbone.log 2>&1
mkdir -p $O/summary
python tools/summarize_prof.py r05 > $O/summary/summarize_prof.txt 2>&1
python tools/summarize_conv_pmc.py r05 > $O/summary/summarize_conv_pmc.txt 2>&1
cp profiles/r05_* profiles/pmc_latest.json $O/summary/ 2>/dev/null
du -sh gpurun_out/* | sort -h | tail -5
rm -rf gpurun_out/r05 gpurun_out/r05_convpmc          # the raw rocprofv3 trees: summarised above
find gpurun_out -size +4M -delete
du -sh gpurun_out; ls $O/summary
