#!/bin/bash
# Run ON THE GPU BOX: PMC counters of one convolution shape, per kernel name (round 4).  usage: bash tools/pmc_conv_w3.sh <tag> I O H up
TAG=${1:-r04pmc}; I=${2:-256}; O=${3:-256}; H=${4:-256}; UP=${5:-1}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
for grp in a b; do
case $grp in
 a) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE";;
 b) C="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE";;
esac
timeout 120 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$grp -o r -- python $R/tools/bench_conv_one.py $I $O $H $UP 3 --f16x2 > /dev/null 2> $OUT/pmc_$grp.log
done
python - <<PY
import csv,glob,collections,os,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob("$OUT/pmc_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].replace("void ","").split("(")[0]
        if k.startswith("k_modconv") or k.startswith("k_fir"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res={}
for k,c in acc.items():
    m={n:sum(v)/len(v) for n,v in c.items()}
    act=m.get("GRBM_GUI_ACTIVE",0)/8.0
    d={"launches":len(next(iter(c.values()))),"active_clocks":act}
    if act:
        d["mfma_busy_frac"]=m.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(act*1024)
        wc=m.get("SQ_WAVE_CYCLES",0)
        if wc:
            d["wait_any_frac"]=m.get("SQ_WAIT_ANY",0)/wc; d["wait_inst_frac"]=m.get("SQ_WAIT_INST_ANY",0)/wc; d["active_inst_frac"]=m.get("SQ_ACTIVE_INST_ANY",0)/wc
        if m.get("SQ_LDS_IDX_ACTIVE"): d["lds_conflict_frac"]=m.get("SQ_LDS_BANK_CONFLICT",0)/m["SQ_LDS_IDX_ACTIVE"]; d["lds_active_frac_of_clocks"]=m["SQ_LDS_IDX_ACTIVE"]/(act*256)
        d["insts_mfma"]=m.get("SQ_INSTS_MFMA"); d["insts_lds"]=m.get("SQ_INSTS_LDS"); d["insts_vmem_rd"]=m.get("SQ_INSTS_VMEM_RD"); d["insts_valu"]=m.get("SQ_INSTS_VALU"); d["insts_salu"]=m.get("SQ_INSTS_SALU")
        d["wait_inst_lds_frac"]=m.get("SQ_WAIT_INST_LDS",0)/wc if wc else None
    res[k]=d
json.dump({"shape":[$I,$O,$H,$UP],"kernels":res},open("$OUT/pmc_summary.json","w"),indent=1)
print(json.dumps(res,indent=1))
PY
