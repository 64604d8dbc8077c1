#!/bin/bash
# Run ON THE GPU BOX (gpurun): round-5 session 2 — GPU suite, bench, blocked-vs-row-major tile order (kernel time + L2 / fabric counters),
# split-K target sweep of the image-fed convolution kernels.
TAG=${1:-r05c}
R=$(pwd); O=$R/gpurun_out/$TAG
mkdir -p $O
python -c "import panic3d_amd as P; assert not P._build.needs_build(), 'stale .so'" || exit 9
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-verify --no-table --no-pipeline --roofline-steps 0"
for ord in 0 1; do
  P3D_TILE_ORDER=$ord timeout 120 $B --steps 40 --warmup 5 > $O/order${ord}_bench.json 2> $O/order${ord}_bench.err
  for grp in fetch write tcp; do
    case $grp in
      fetch) C="FETCH_SIZE GRBM_GUI_ACTIVE" ;;
      write) C="WRITE_SIZE GRBM_GUI_ACTIVE" ;;
      tcp)   C="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" ;;
    esac
    P3D_TILE_ORDER=$ord timeout 180 rocprofv3 --pmc $C --output-format csv -d $O/pmc_order${ord}_$grp -o r -- $B --steps 6 --warmup 2 > $O/pmc_order${ord}_$grp.json 2> $O/pmc_order${ord}_$grp.log || echo "pmc $ord $grp failed" >> $O/failures.txt
  done
done
cd $R
for t in 128 256 384 512 768 1024; do
  echo "== P3D_KSPLIT_TARGET_IMG=$t"; P3D_KSPLIT_TARGET_IMG=$t timeout 120 python tools/bench_backbone.py 2>/dev/null | tail -1
done > $O/ksplit_sweep.txt 2>&1
cat $O/ksplit_sweep.txt | cut -c1-400
python - <<PY
import csv, glob, json, collections
for o in (0, 1):
    res = {}
    for grp in ("fetch", "write", "tcp"):
        for fn in glob.glob("$O/pmc_order%d_%s/**/*counter_collection.csv" % (o, grp), recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(fn)):
                if "k_render<" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                res[k] = sum(v) / len(v)
    try:
        b = json.loads([l for l in open("$O/order%d_bench.json" % o) if l.startswith("{")][-1])
        res["kernel_ms"] = b["roofline"]["kernel_ms"]; res["ms_per_step"] = b["ms_per_step"]
    except Exception as e:
        res["bench_error"] = str(e)
    print("order", o, json.dumps(res))
PY
