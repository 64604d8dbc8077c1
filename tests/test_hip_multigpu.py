"""GPU, needs >= 2 devices (skipped on the 1-GPU test box; the driver's 8-GPU node runs it): BASELINE configs c4 / c5 and
bench.py under torch.distributed.run with RCCL at world size 2, compared with the 1-process run.

c4 (tools/sweep360.py --check): every view draws from its own seeded generator and keeps its own depth-clamp range, so the
gathered [views,4,res,res] tensor must hash identically for world size 1 and 2 (and for batched launches).  c5
(tools/bench_c5.py --check): the gathered sigma grid must hash identically.  bench.py --gpus 2 must print its JSON line with
per-rank timings and a verified frame."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL at world size 2)")]


def _run(script, args, nproc, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, script)] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, script)] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[-1])


def test_c4_sweep_same_frames_for_world_size_1_and_2():
    a = _run("tools/sweep360.py", ["--views", "8", "--res", "256", "--check"], 1, 0)
    b = _run("tools/sweep360.py", ["--views", "8", "--res", "256", "--check"], 2, 29551)
    c = _run("tools/sweep360.py", ["--views", "8", "--res", "256", "--check", "--batch", "2"], 2, 29552)
    assert b["n_gpus"] == 2 and a["sha256"] == b["sha256"] == c["sha256"]


def test_c5_grid_same_sigma_for_world_size_1_and_2():
    a = _run("tools/bench_c5.py", ["--grid", "192", "--check"], 1, 0)
    b = _run("tools/bench_c5.py", ["--grid", "192", "--check"], 2, 29553)
    assert b["n_gpus"] == 2 and a["sha256"] == b["sha256"] and a["verts"] == b["verts"]


def test_bench_two_ranks_prints_per_rank_times_and_verifies():
    out = _run("bench.py", ["--gpus", "2", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-table"], 2, 29554)
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["verify"]["ok"]
    assert len(out["per_rank"]["ms_per_step_render"]) == 2 and len(out["per_rank"]["gather_ms"]) == 2


def test_unwrapped_gpus_flag_starts_its_own_ranks():
    """The driver's plain form `python bench.py --gpus 2` (no torch.distributed.run around it) and the tools' `--gpus 2`."""
    out = _run("bench.py", ["--gpus", "2", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-table", "--no-verify"], 1, 0)
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and len(out["per_rank"]["gather_ms"]) == 2
    a = _run("tools/sweep360.py", ["--views", "8", "--res", "256", "--check"], 1, 0)
    b = _run("tools/sweep360.py", ["--views", "8", "--res", "256", "--check", "--gpus", "2"], 1, 0)
    assert b["n_gpus"] == 2 and a["sha256"] == b["sha256"]
    c = _run("tools/bench_c5.py", ["--grid", "128", "--check", "--gpus", "2"], 1, 0)
    assert c["n_gpus"] == 2
