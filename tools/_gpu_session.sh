timeout 120 python tools/experiments/fir_img_time.py 2>/dev/null | tail -1
for p in 44 48 52 56; do P3D_LIB=$PWD/build/lib_fir$p.so timeout 120 python tools/experiments/fir_img_time.py 2>/dev/null | tail -1; done
