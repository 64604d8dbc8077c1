#!/bin/bash
# Run ON THE GPU BOX: per-launch PMC counters of one whole backbone pass and one super-resolution pass with the SHIPPED default
# convolution kernels (two-term f16 operands), plus an ordered kernel trace with durations.  Condense afterwards with
#   python tools/summarize_conv_pmc.py <tag>          -> profiles/<tag>_mfma_util.json
# usage: bash tools/pmc_backbone.sh <tag>
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_convpmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$R'); import panic3d_amd as P; print(P._build.source_hash())" > $O/kernel_src_sha.txt
python -c "import sys; sys.path.insert(0, '$R'); import panic3d_amd as P; print(P._build.synthesis_source_hash())" > $O/synthesis_src_sha.txt
for what in bb sr; do
  FL=""; [ $what = sr ] && FL="--sr"
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$what -o r -- python $R/tools/profile_backbone.py --passes 2 $FL > $O/trace_$what.log 2>&1
  for grp in a b; do
    case $grp in
      a) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE";;
      b) C="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE";;
    esac
    timeout 200 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${what}_$grp -o r -- python $R/tools/profile_backbone.py --passes 1 $FL > $O/pmc_${what}_$grp.log 2>&1 || echo "pmc $what $grp failed" >> $O/failures.txt
  done
done
ls $O
