#!/bin/bash
# Run ON THE GPU BOX (gpurun): the round's closing validation of the shipped sources — the GPU test suite, per-layer convolution
# times, whole-pass times, per-launch counters of a backbone / super-resolution pass, and the renderer's kernel statistics + PMC
# passes.  Condense afterwards (build container): python tools/summarize_prof.py r04; python tools/summarize_conv_pmc.py r04
python -c "import panic3d_amd as P; assert not P._build.needs_build()" || exit 9
mkdir -p gpurun_out/r04final
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r04final/pytest_gpu.txt 2>&1
cat gpurun_out/r04final/pytest_gpu.txt
timeout 200 python tools/conv_layers_time.py --n 30 > gpurun_out/r04final/conv_layers.txt 2>/dev/null
timeout 200 python tools/graph_backbone.py > gpurun_out/r04final/passes.txt 2>/dev/null; cat gpurun_out/r04final/passes.txt
bash tools/pmc_backbone.sh r04 > gpurun_out/r04final/pmc_backbone.log 2>&1
bash tools/collect_profile.sh r04 > gpurun_out/r04final/collect.log 2>&1
tail -40 gpurun_out/r04final/collect.log
