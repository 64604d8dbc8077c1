#!/bin/bash
# One gpurun lease of round 6:  tools/r06_session.sh <tag> <step> [<step> ...]   (outputs under gpurun_out/<tag>/)
# steps: up4test | up4ab | synth | gputests | bench | convpmc | ...   (each a function below)
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
up4test() { timeout 900 python -m pytest tests/test_hip_synthesis.py -x -q -k "up4 or fir_pass_inside or activation_image" > "$OUT/up4test.log" 2>&1; tail -5 "$OUT/up4test.log"; }
up4ab() { timeout 600 python tools/up4_ab.py --n 30 > "$OUT/up4ab.jsonl" 2> "$OUT/up4ab.err"; cat "$OUT/up4ab.jsonl"; tail -3 "$OUT/up4ab.err"; }
synth() { timeout 1500 python -m pytest tests/test_hip_synthesis.py -x -q ${SYNTH_K:+-k "$SYNTH_K"} > "$OUT/synth.log" 2>&1; tail -5 "$OUT/synth.log"; }
gputests() { timeout 2400 python -m pytest tests -x -q -m gpu > "$OUT/gputests.log" 2>&1; tail -8 "$OUT/gputests.log"; }
benchtest() { timeout 1500 python -m pytest tests/test_hip_bench_contract.py -x -q > "$OUT/benchtest.log" 2>&1; tail -8 "$OUT/benchtest.log"; }
bench() { timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 3000 "$OUT/bench.json"; tail -3 "$OUT/bench.err"; }
convpmc() { bash tools/pmc_backbone.sh "$TAG" > "$OUT/convpmc.log" 2>&1; python tools/summarize_conv_pmc.py "$TAG" > "$OUT/convpmc_summary.txt" 2>&1; tail -60 "$OUT/convpmc_summary.txt"; }
rgbtest() { timeout 900 python -m pytest tests/test_hip_synthesis.py -x -q -s -k "torgb or superresolution or fullsize or generator_vs" > "$OUT/rgbtest.log" 2>&1; grep -i "ToRGB on\|passed\|failed\|Error" "$OUT/rgbtest.log" | tail -15; }
torgbtime() { timeout 600 python tools/torgb_time.py > "$OUT/torgb_time.jsonl" 2> "$OUT/torgb_time.err"; cat "$OUT/torgb_time.jsonl"; tail -3 "$OUT/torgb_time.err"; }
trace() { bash tools/trace_passes.sh "$TAG" > "$OUT/trace.txt" 2>&1; grep -v "^$" "$OUT/trace.txt" | cut -c1-110 | tail -70; }
graphbb() { timeout 600 python tools/graph_backbone.py > "$OUT/graph_backbone.txt" 2>&1; tail -5 "$OUT/graph_backbone.txt"; }
renderprof() { bash tools/collect_profile.sh "$TAG" "surface canonical" > "$OUT/renderprof.log" 2>&1; ls "$OUT" | head -50; }
secondary() { bash tools/secondary_benchmarks.sh > "$OUT/secondary.txt" 2>&1; cat "$OUT/secondary.txt" | cut -c1-400; }
for step in "$@"; do echo "== $step"; $step; done
