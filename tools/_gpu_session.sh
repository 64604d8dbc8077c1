python tools/experiments/ablate/time_layers.py 2>/dev/null | tail -1
for t in base w2_nox w2_now nostore up_nox up_now; do P3D_LIB=$GRAFT_REPO_ROOT/tools/experiments/ablate/lib_$t.so python tools/experiments/ablate/time_layers.py 2>/dev/null | tail -1; done
