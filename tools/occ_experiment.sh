for lib in build/lib_occ2_d4.so build/lib_occ3_d1.so build/lib_occ3_d2.so; do
  for s in "32 32" "48 48"; do set -- $s
    echo "== $lib $1+$2"
    P3D_LIB=$PWD/$lib python tools/fast_color_check.py --sc $1 --sf $2 2>&1 | grep -E "^(canonical|surface)" | python -c "
import sys, json
for l in sys.stdin:
    name, js = l.split(' ', 1); d = json.loads(js)
    print(name, ' '.join(f'{k[3:]}={d[k]:.3f}' for k in d if k.startswith('ms_')))"
  done
done
