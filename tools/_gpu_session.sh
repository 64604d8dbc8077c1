timeout 600 python tools/experiments/img_conv_check.py 2>&1 | tail -7
echo "== w4 off"
P3D_CONV_W4=0 timeout 600 python tools/experiments/img_conv_check.py 2>&1 | tail -7 | cut -c1-200
