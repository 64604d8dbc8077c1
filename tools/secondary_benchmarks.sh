for t in bench_c2 sweep360 bench_c5 bench_backbone generate_subject profile_f bench_small_view; do
  echo "== $t (default)"; timeout 300 python tools/$t.py 2>/dev/null | tail -2
done
for t in bench_c2 bench_backbone generate_subject profile_f; do
  echo "== $t (P3D_CONV_MMA=f32)"; P3D_CONV_MMA=f32 timeout 300 python tools/$t.py 2>/dev/null | tail -2
done
