cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in a b c; do
case $grp in
 a) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE";;
 b) C="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE";;
 c) C="SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE";;
esac
timeout 120 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_conv_$grp -o r -- python $R/tools/bench_conv_one.py 256 256 256 1 3 --f16x2 > /dev/null 2>&1
done
python - <<'PY'
import csv,glob,collections,os
acc=collections.defaultdict(list)
for fn in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_conv_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_modconv_h" in r["Kernel_Name"] or "k_modconv_w2" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()): print(k, sum(v)/len(v), len(v))
PY
