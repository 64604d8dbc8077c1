mkdir -p gpurun_out/r03k
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -x -q -m gpu -k "out_of_order or other_sampling or in_kernel or 96 or rates" > gpurun_out/r03k/pytest_tcg.log 2>&1; echo "pytest rc $?"
tail -15 gpurun_out/r03k/pytest_tcg.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03k/bench.json 2> gpurun_out/r03k/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03k/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("results", "eval_faithful"):
    print(k, json.dumps(d.get(k))[:1500])
PY
