for T in 64 128 256; do echo ks$T; P3D_LIB=$GRAFT_REPO_ROOT/tools/experiments/lib_ks$T.so python tools/graph_backbone.py 2>/dev/null | tail -1; done
echo default; python tools/graph_backbone.py 2>/dev/null | tail -1
