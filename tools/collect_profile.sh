#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel-trace statistics and PMC passes of bench.py for both bench scenes, in the exact-contract
# mode (the default since round 3) and the tolerance mode (--fast), with the exact early-outs on and off.  Outputs under gpurun_out/<tag>/; condense them afterwards (build container) with
#   python tools/summarize_prof.py <tag>
# PMC passes are separate runs with --pmc only (no trace domains), one counter group per pass (SQ has 8 slots; FETCH_SIZE
# and WRITE_SIZE do not fit one TCC pass) — /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots".
#   usage: bash tools/collect_profile.sh <tag> [scenes="canonical surface"] [extra bench flags]
set -u
TAG=${1:-r02}
SCENES=${2:-"canonical surface"}
EXTRA=${3:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$REPO'); import panic3d_amd as P; print(P._build.source_hash())" > "$OUT/kernel_src_sha.txt"
python -c "import sys; sys.path.insert(0, '$REPO'); import panic3d_amd as P; print(P._build.render_source_hash())" > "$OUT/render_src_sha.txt"
B="python $REPO/bench.py --no-cpu-baseline --no-verify --no-table --no-pipeline --roofline-steps 0 $EXTRA"
for scene in $SCENES; do
  for mode in early noearly exact_early exact_noearly; do
    FL="--scene $scene"
    case $mode in   # early / noearly = the tolerance mode (bench.py --fast); exact_* = the default since round 3
      early) FL="$FL --fast" ;;
      noearly) FL="$FL --fast --no-early-out" ;;
      exact_early) ;;
      exact_noearly) FL="$FL --no-early-out" ;;
    esac
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_${scene}_${mode}" -o r -- $B $FL --steps 30 --warmup 3 \
      > "$OUT/stats_${scene}_${mode}.json" 2> "$OUT/stats_${scene}_${mode}.log"
    [ "${PMC:-1}" = 0 ] && continue
    [ "$mode" != early ] && [ "$mode" != exact_early ] && [ "$mode" != exact_noearly ] && continue   # counters: both timed launches, and the exact kernel with every sample decoded (the roofline figure)
    for grp in sq tcp fetch write; do
      case $grp in
        sq)    C="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" ;;
        tcp)   C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" ;;
        fetch) C="FETCH_SIZE GRBM_GUI_ACTIVE" ;;
        write) C="WRITE_SIZE GRBM_GUI_ACTIVE" ;;
      esac
      timeout 180 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${scene}_${mode}_${grp}" -o r -- $B $FL --steps 6 --warmup 2 \
        > "$OUT/pmc_${scene}_${mode}_${grp}.json" 2> "$OUT/pmc_${scene}_${mode}_${grp}.log" || echo "pass $scene $mode $grp failed (rc $?)" >> "$OUT/failures.txt"
    done
  done
done
ls "$OUT"
