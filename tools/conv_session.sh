#!/bin/bash
# Run ON THE GPU BOX: per-layer times of the convolution kernels (HIP events + kernel trace) and whole-pass times.
#   usage: bash tools/r04_conv_session.sh <tag>
set -u
TAG=${1:-r04a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$REPO'); import panic3d_amd as P; print(P._build.source_hash())" > "$OUT/kernel_src_sha.txt"
timeout 300 python $REPO/tools/conv_layers_time.py --n 30 > "$OUT/conv_layers.txt" 2> "$OUT/conv_layers.log"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_conv" -o r -- python $REPO/tools/conv_layers_time.py --n 10 \
   > "$OUT/stats_conv.txt" 2> "$OUT/stats_conv.log"
timeout 300 python $REPO/tools/graph_backbone.py > "$OUT/passes.txt" 2> "$OUT/passes.log"
cat "$OUT/conv_layers.txt" "$OUT/passes.txt"
find "$OUT/stats_conv" -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -30 {}'
