#!/bin/bash
# Development aid (GPU box): kernel times of several builds of the library (P3D_LIB override), exact / fast, early-out on / off.
for lib in "$@"; do
  echo "== $lib"
  P3D_LIB=$PWD/$lib python tools/fast_color_check.py 2>&1 | grep -E "^(canonical|surface)" | python -c "
import sys, json
for l in sys.stdin:
    name, js = l.split(' ', 1); d = json.loads(js)
    print(name, ' '.join(f'{k[3:]}={d[k]:.3f}' for k in d if k.startswith('ms_')))"
  [ -n "${C5:-}" ] && P3D_LIB=$PWD/$lib python tools/bench_c5.py --grid 512 2>&1 | tail -1 | cut -c1-160
done
