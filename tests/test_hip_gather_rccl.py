"""GPU, ONE device: the path's only collective — the gather of final RGBA frames to rank 0 — on RCCL itself (backend 'nccl' on
ROCm) in a one-rank group with force=True, so that what the 1-GPU test box can exercise of it IS exercised there: communicator
creation, `FrameGather`'s collective mode decision, the asynchronous `dist.gather` on the collective's stream writing in place
into the result, the batched point-to-point form, and `gather_frames`.  (World size 2 on RCCL: tests/test_hip_multigpu.py, which
needs two devices; world size 2 / 3 on gloo: tests/test_host_cpu.py.)  The rank runs in a subprocess under torch.distributed.run —
the driver's own launch form — with a timeout: a collective that hangs fails the test instead of the session."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")]

RANK_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import panic3d_amd as P
from panic3d_amd import sharding
assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0", "sharding must have set the dmabuf IPC mode before HIP initialises"
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
res, K = 64, 8
g = torch.Generator(device=dev).manual_seed(5)
frames = torch.rand((K, res, res, 4), device=dev, generator=g)
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "p2p_env": os.environ.get("P3D_GATHER_P2P", "0")}
fg = sharding.FrameGather(frames, K, dst=0, force=True)
out["p2p"] = bool(fg.p2p)
side = torch.cuda.Stream()
for lo in range(0, K, 2):          # slices pushed while "rendering" goes on on another stream
    fg.push(lo, lo + 2)
    with torch.cuda.stream(side):
        torch.rand((256, 256), device=dev).sum()
got = fg.finish()
torch.cuda.synchronize()
out["streamed_equal"] = bool(torch.equal(got, frames)) and got.data_ptr() != frames.data_ptr()
out["pending_left"] = len(fg.pending)
one = sharding.gather_frames(frames, counts=[K], dst=0, force=True)
torch.cuda.synchronize()
out["end_equal"] = bool(torch.equal(one, frames))
out["plain_is_local"] = sharding.gather_frames(frames) is frames  # world 1 without force: no collective
try:
    sharding.FrameGather(frames, K + 1, dst=0, force=True)
    out["bad_count_raises"] = False
except RuntimeError:
    out["bad_count_raises"] = True
dist.barrier()
dist.destroy_process_group()
print(json.dumps(out))
'''


@pytest.mark.parametrize("p2p", ["0", "1"])
def test_frame_gather_on_rccl_in_a_one_rank_group(tmp_path, p2p):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}  # the package has to set it itself
    env.update(MASTER_ADDR="127.0.0.1", P3D_GATHER_P2P=p2p)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(29570 + int(p2p)), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["p2p"] == (p2p == "1")  # the mode is decided collectively, from the environment of every rank
    assert out["streamed_equal"] and out["end_equal"] and out["plain_is_local"] and out["bad_count_raises"] and out["pending_left"] == 0


def test_bench_plain_and_launched_forms_take_the_same_path(tmp_path):
    """`python bench.py --gpus 1` under torch.distributed.run is the multi-GPU code path with one rank (RCCL communicator, streamed
    gather, per-rank report); the environment it needs is set by bench.py / sharding themselves, not inherited."""
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    env.update(MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", "29573",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-table", "--no-verify",
           "--roofline-steps", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_ranks_seen"] == 1 and d["gather"]["mode"].startswith("streamed") and d["gather"]["backend"].startswith("nccl")
    assert d["gather"]["bytes_into_rank0"] == 0 and len(d["per_rank"]["gather_ms"]) == 1 and d["gather"]["exposed_ms_max"] >= 0
