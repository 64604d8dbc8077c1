"""GPU: the driver's contract with bench.py — ONE JSON line with the metric of BASELINE.json, the roofline and cpu_baseline
objects, a frame verified against the oracle — for the plain 1-GPU invocation and for the launch the driver uses for N > 1
(`python -m torch.distributed.run ... bench.py --gpus N`, here with one rank: RCCL communicator, streamed frame gather, barriers,
max-over-ranks timing)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")]
SMALL = ["--steps", "6", "--warmup", "2", "--roofline-steps", "3"]


def _bench(args, launched):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable]
    if launched:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29561"]
    r = subprocess.run(cmd + [os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # exactly one JSON line
    return json.loads(lines[0])


def _check_line(d, steps):
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].startswith("rendered rays/sec at 512") and base["metric"].startswith("rendered rays/sec at 512")
    assert d["unit"] == "rays/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["vs_baseline"] is None and d["value"] > 1e6
    assert abs(d["value"] - 512 * 512 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]  # whole-job rays / wall time of the K steps
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0 and "traffic" in r
    # achieved = algorithmic bytes per launch / the every-sample kernel's time (HIP events)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_no_early_out"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert d["verify"]["ok"] is True


def test_bench_default_invocation_prints_the_contract_line():
    d = _bench(SMALL, launched=False)
    _check_line(d, 6)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "rays/s" and c["sample"]
    assert "per_rank" not in d


def test_bench_under_torch_distributed_run_streams_the_frames():
    d = _bench(SMALL + ["--no-cpu-baseline"], launched=True)
    _check_line(d, 6)
    assert len(d["per_rank"]["ms_per_step_render"]) == 1 and len(d["per_rank"]["gather_ms"]) == 1
    assert "slices sent while the next frames render" in d["config"]["workload"]
    e = _bench(SMALL + ["--no-cpu-baseline", "--gather", "end", "--exact"], launched=True)
    _check_line(e, 6)
    assert e["dtype"] == "f32" and "ONE gather" in e["config"]["workload"]
