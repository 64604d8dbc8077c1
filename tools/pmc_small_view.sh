#!/bin/bash
# Run ON THE GPU BOX: counters of the pipeline's own renderer launch (128^2 rays x (96+96), k_render_quad, tolerance mode; and the
# exact mode) — separate rocprofv3 --pmc passes, one counter group each, plus a kernel trace.  Condense: python tools/summarize_small_pmc.py <tag>
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_smallpmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$R'); import panic3d_amd as P; print(P._build.render_source_hash())" > $O/render_src_sha.txt
for mode in tol exact; do
  FL=""; [ $mode = exact ] && FL="--exact"
  [ -z "${GROUPS_ONLY:-}" ] && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$mode -o r -- python $R/tools/small_view_loop.py --n 30 $FL > $O/stats_$mode.json 2> $O/stats_$mode.log
  for grp in ${GROUPS_ONLY:-sq tcp lds}; do
    case $grp in
      sq)  C="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" ;;  # (8 SQ slots)
      tcp) C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" ;;
      lds) C="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT GRBM_GUI_ACTIVE" ;;
    esac
    timeout 120 rocprofv3 --pmc $C --output-format csv -d $O/pmc_${mode}_$grp -o r -- python $R/tools/small_view_loop.py --n 6 $FL > $O/pmc_${mode}_$grp.json 2> $O/pmc_${mode}_$grp.log || echo "pass $mode $grp failed" >> $O/failures.txt
  done
done
ls $O
