#!/bin/bash
# Run ON THE GPU BOX: ordered kernel trace (durations) of the last backbone pass and the last super-resolution pass.  usage: bash tools/trace_passes.sh <tag>
TAG=${1:-r04t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for what in bb sr; do
  FL=""; [ $what = sr ] && FL="--sr"
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$what -o r -- python $R/tools/profile_backbone.py --passes 2 $FL > $O/trace_$what.log 2>&1
done
python - <<PY
import csv,glob
for what in ("bb","sr"):
    f=glob.glob("$O/trace_%s/**/*kernel_trace.csv"%what,recursive=True)[0]
    rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r["Start_Timestamp"]))
    idx=[i for i,r in enumerate(rows) if "k_demod_plan" in r["Kernel_Name"]]
    seg=rows[idx[-1]:]
    t0=int(seg[0]["Start_Timestamp"]); tot=0
    print(what, "pass: %d launches, span %.1f us"%(len(seg),(int(seg[-1]["End_Timestamp"])-t0)/1e3))
    for r in seg:
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; tot+=d
        n=r["Kernel_Name"].replace("void ","").split("(")[0][:40]
        print("  %-40s %9s x%3s x%3s  start %8.1f  dur %7.1f"%(n,r["Grid_Size_X"],r["Grid_Size_Y"],r["Grid_Size_Z"],(int(r["Start_Timestamp"])-t0)/1e3,d))
    print("  sum of kernel durations %.1f us"%tot)
PY
