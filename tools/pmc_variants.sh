#!/bin/bash
# Development aid (GPU box): the same bench under rocprofv3 --pmc for several builds of the library (P3D_LIB override).
#   bash tools/pmc_variants.sh <tag> "<lib1> <lib2> ..." "<bench flags>"
set -u
TAG=$1; LIBS=$2; FLAGS=${3:-"--scene surface --no-early-out"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-verify --roofline-steps 0 --steps 6 --warmup 2 $FLAGS"
declare -A G
G[ta]="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
G[td]="GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum TA_BUFFER_READ_WAVEFRONTS_sum"
G[tcp1]="GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
G[tcp2]="GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum"
G[sq]="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD"
for lib in $LIBS; do
  name=$(basename $lib .so)
  for g in ta td tcp1 tcp2 sq; do
    P3D_LIB=$REPO/$lib timeout 180 rocprofv3 --pmc ${G[$g]} --output-format csv -d "$OUT/${name}_$g" -o r -- $B > "$OUT/${name}_$g.json" 2> "$OUT/${name}_$g.log" || echo "$name $g failed" >> "$OUT/failures.txt"
  done
done
python - <<PY
import csv, glob, collections, os
out = "$OUT"
res = collections.defaultdict(dict)
for fn in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    name = os.path.relpath(fn, out).split(os.sep)[0].rsplit("_", 1)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if "k_render<" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[name][k] = sum(v) / len(v)
keys = sorted({k for d in res.values() for k in d})
print("%-42s" % "counter" + "".join("%16s" % n for n in sorted(res)))
for k in keys:
    print("%-42s" % k + "".join("%16.4g" % res[n].get(k, float("nan")) for n in sorted(res)))
PY
cat "$OUT/failures.txt" 2>/dev/null
