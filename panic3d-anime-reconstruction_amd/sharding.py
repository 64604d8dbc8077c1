"""Multi-GPU layer of the hot path: independent views (or density-grid slabs) are sharded over ranks, one process per
GPU; the ONLY collective is one gather of final RGBA frames to rank 0 (RCCL over xGMI: backend 'nccl' on ROCm; 'gloo'
in the CPU tests).  The reference's eval path is single-GPU (_scripts/eval/generate.py:16); its view loop
(generate.py:108-117) and the 360-degree sweep (_train/eg3dc/util/eg3dc_v0.py:64-87 quickspin) are what gets sharded.
"""
import os
import socket
import sys

import torch
import torch.distributed as dist

# RCCL between processes needs dmabuf IPC on this driver (hipIpcGetMemHandle fails with "invalid argument" otherwise), and the HSA
# runtime reads HSA_ENABLE_IPC_MODE_LEGACY when it initialises — at the first HIP call of the process.  The variable is therefore set
# only where RANKS come to life (not for every importer of the package — a single-process user keeps the runtime's
# default IPC mode, e.g. for torch.multiprocessing tensor sharing):
#   * a process that a launcher started as a rank (RANK and MASTER_ADDR in its environment: torch.distributed.run, the driver's
#     form, ensure_ranks' own re-exec) gets it here, at import, which precedes any HIP call this package makes;
#   * ensure_ranks puts it into the environment of the ranks it starts;
#   * a rank whose HIP runtime was initialised without it is TOLD so when it first touches the collective (_check_ipc_env) instead of
#     failing later inside RCCL with "invalid argument".  A value the user exported always wins.
def _is_launched_rank():
    return "RANK" in os.environ and "MASTER_ADDR" in os.environ


if _is_launched_rank():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def prepare_rank_env(env=None):
    """For launchers that create ranks themselves — `torch.multiprocessing.spawn`, a process that sets RANK / MASTER_ADDR after
    importing this package: put HSA_ENABLE_IPC_MODE_LEGACY=0 into `env` (default: os.environ, which spawned children
    inherit) unless the user exported a value.  Call it BEFORE the first HIP call of every process that will be a rank — in the
    parent before `spawn`, or first thing in the child; afterwards the runtime's IPC mode is fixed and only `_check_ipc_env`'s warning is
    left.  Returns True when the variable is (now) set, False — with a RuntimeWarning — when HIP is already up in this process without it."""
    env = os.environ if env is None else env
    if env.get("HSA_ENABLE_IPC_MODE_LEGACY") is not None:
        return True
    if env is os.environ and torch.cuda.is_initialized():
        import warnings
        warnings.warn("panic3d_amd.sharding.prepare_rank_env: the HIP runtime of this process is already initialised without "
                      "HSA_ENABLE_IPC_MODE_LEGACY=0; RCCL between processes will fail with 'invalid argument' on this driver — "
                      "call prepare_rank_env() (or export the variable) before the first torch.cuda / panic3d_amd device call", RuntimeWarning, stacklevel=2)
        return False
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return True


def _check_ipc_env():
    """Called where a multi-rank RCCL group is first used: the IPC mode cannot be changed any more (HIP is up), so say what to do."""
    if (dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() > 1
            and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") is None):
        import warnings
        warnings.warn("panic3d_amd.sharding: HSA_ENABLE_IPC_MODE_LEGACY is not set in this rank's environment; on this driver RCCL between "
                      "processes needs HSA_ENABLE_IPC_MODE_LEGACY=0 BEFORE the first HIP call (export it, start the ranks through "
                      "sharding.ensure_ranks / torch.distributed.run, or call sharding.prepare_rank_env() before spawning / before the first device call)",
                      RuntimeWarning, stacklevel=3)


def ensure_ranks(gpus, argv=None):
    """`--gpus N` of bench.py / tools/sweep360.py / tools/bench_c5.py: make sure N ranks exist, one process per GPU.

    * under a launcher (RANK in the environment — `python -m torch.distributed.run ...`, the driver's form for N > 1):
      WORLD_SIZE must equal N (N = None: accept whatever the launcher started), else SystemExit;
    * no launcher and N > 1: this process re-executes itself under `python -m torch.distributed.run --nnodes=1
      --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>` with the same command line (does not return) — the
      reference's multi-process entry point spawns its own ranks too (_train/eg3dc/trainers/train_eclustrousC.py:39-50,109-114);
    * no launcher and N in (None, 1): single process, returns.
    Device-count checks are the caller's (they need torch.cuda, which a CPU dry launch must not touch)."""
    if "RANK" in os.environ:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if gpus is not None and world != int(gpus):
            raise SystemExit(f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks")
        return
    if gpus is None or int(gpus) <= 1:
        return
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = list(sys.argv if argv is None else argv)
    env = dict(os.environ, P3D_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(gpus)}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + argv
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(sys.executable, cmd, env)


def partition(n_items, world_size, rank):
    """Contiguous block partition: rank r gets items [lo, hi).  First (n % world) ranks get one extra item."""
    base, rem = divmod(int(n_items), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def frames_rgba(feat, wsum, res, out=None, channels_last=False):
    """Final RGBA frame of one render: RGB = first 3 feature channels (training/triplane.py:223 image_raw), A = weights
    sum (generate.py:143-146).  feat [N,R,32], wsum [N,R,1] -> [N,4,res,res], or [N,res,res,4] with channels_last (no
    transposition: two strided copies; the consumer permutes once after the gather).  `out`: preallocated destination."""
    N = feat.shape[0]
    if channels_last:
        if out is None:
            out = torch.empty((N, res, res, 4), dtype=feat.dtype, device=feat.device)
        flat = out.view(N, res * res, 4)
        flat[..., :3] = feat[..., :3]
        flat[..., 3:] = wsum
        return out
    rgba = torch.cat([feat[..., :3], wsum], dim=-1).permute(0, 2, 1).reshape(N, 4, res, res)
    if out is None:
        return rgba.contiguous()
    out.copy_(rgba)
    return out


_COMM_READY = set()  # (backend, world) whose default communicator has seen a group-wide collective in this process


def _ensure_communicator(device):
    """RCCL/NCCL build a communicator lazily, and PyTorch requires EVERY rank to take part in the first batched point-to-point
    operation of a group — ranks without frames skip the P2P batch, so one tiny group-wide all_reduce goes first (once per process)."""
    key = (dist.get_backend(), dist.get_world_size())
    if key not in _COMM_READY:
        _check_ipc_env()
        dist.all_reduce(torch.zeros(1, device=device))
        _COMM_READY.add(key)


def gather_frames(local, counts=None, dst=0, force=False):
    """Gather per-rank frame stacks [n_r, ...] to rank `dst` in rank order: a TRUE gather — every rank sends its own frames
    to dst once and nothing else moves (north_star: "RCCL gather over xGMI of final RGBA only").  Issued as one batch of
    point-to-point operations (RCCL groups them: rank dst receives over its 7 xGMI links concurrently, each message exactly
    counts[r] frames, written straight into its slice of the result — no padding, no staging copy).  A rank with no frames
    (counts[r] == 0, local of shape [0, ...]) takes part without sending.  Returns [sum n_r, ...] on dst, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if counts is None:
        c = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
        allc = [torch.zeros_like(c) for _ in range(world)]
        dist.all_gather(allc, c)
        counts = [int(x.item()) for x in allc]
    if local.shape[0] != counts[rank]:
        raise RuntimeError(f"rank {rank} holds {local.shape[0]} frames but counts[{rank}] = {counts[rank]}")
    local = local.contiguous()
    _ensure_communicator(local.device)
    if rank != dst:
        if counts[rank] > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, dst)]):
                w.wait()
        return None
    out = torch.empty((sum(counts),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    offs = [sum(counts[:r]) for r in range(world)]
    ops_ = [dist.P2POp(dist.irecv, out[offs[r]:offs[r] + counts[r]], r) for r in range(world) if r != dst and counts[r] > 0]
    reqs = dist.batch_isend_irecv(ops_) if ops_ else []
    out[offs[dst]:offs[dst] + counts[dst]].copy_(local)
    for w in reqs:
        w.wait()
    return out


class FrameGather:
    """The same true gather, streamed: every rank renders `per_rank` frames into its own stack and hands finished slices
    [lo, hi) to `push` while it keeps rendering — the transfers run on the collective's own stream (RCCL over xGMI) under the
    next slices' kernels, and `finish` only waits for what is still in flight.  Rank `dst` ends with [world * per_rank, ...]
    in rank order (`out`), written in place (no staging copy); every slice is one asynchronous `gather` (only dst receives);
    every rank must push the same [lo, hi) sequence.  With 8 ranks at 512^2 x RGBA fp32 this hides 5.9 GB of inbound frames per 200-view sweep
    behind rendering instead of paying them after it."""

    def __init__(self, local, per_rank, dst=0, force=False):
        """force: go through the collective also when the group has ONE rank (tests on a 1-GPU box: the communicator, the
        asynchronous gather and the in-place receive buffers are exercised on RCCL; default: a plain copy)."""
        self.local, self.per_rank, self.dst, self.force = local, int(per_rank), int(dst), bool(force)
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.active else 1
        self.rank = dist.get_rank() if self.active else 0
        if local.shape[0] != self.per_rank:
            raise RuntimeError(f"rank {self.rank}: the local stack holds {local.shape[0]} frames, per_rank = {per_rank}")
        self.out = None
        if self.rank == self.dst:
            self.out = torch.empty((self.world * self.per_rank,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        self.pending = []
        # gather() or batched point-to-point: decided ONCE, collectively (a rank that switched on its own — e.g. after an
        # exception only dst sees — would leave the others inside gather()): P3D_GATHER_P2P=1 or a backend without gather(),
        # then the maximum over ranks, which is also the group-wide collective that has to precede the first P2P batch
        want = os.environ.get("P3D_GATHER_P2P", "0") == "1" or (self.active and dist.get_backend() not in ("nccl", "gloo"))
        if self.active and (self.world > 1 or self.force):
            _check_ipc_env()
            flag = torch.tensor([1.0 if want else 0.0], device=local.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            want = bool(flag.item() > 0)
            _COMM_READY.add((dist.get_backend(), self.world))
        self.p2p = want

    def push(self, lo, hi):
        """Frames [lo, hi) of every rank's stack are final: start moving them."""
        if not (0 <= lo < hi <= self.per_rank):
            raise ValueError(f"slice [{lo}, {hi}) outside [0, {self.per_rank})")
        if not self.active or (self.world == 1 and not self.force):
            if self.out is not None:
                self.out[lo:hi].copy_(self.local[lo:hi])
            return
        # one `gather` per slice on the default communicator (RCCL: the root receives, every other rank sends its slice once),
        # asynchronous: it runs on the collective's stream, this rank's render stream goes on
        dests = [self.out[r * self.per_rank + lo:r * self.per_rank + hi] for r in range(self.world)] if self.rank == self.dst else None
        if not self.p2p:
            self.pending.append(dist.gather(self.local[lo:hi], gather_list=dests, dst=self.dst, async_op=True))
            return
        if self.rank != self.dst:
            self.pending += dist.batch_isend_irecv([dist.P2POp(dist.isend, self.local[lo:hi], self.dst)])
            return
        recvs = [dist.P2POp(dist.irecv, dests[r], r) for r in range(self.world) if r != self.dst]
        if recvs:  # (a one-rank group has nobody to receive from)
            self.pending += dist.batch_isend_irecv(recvs)
        dests[self.dst].copy_(self.local[lo:hi])

    def finish(self):
        """Wait for the transfers still in flight; returns the gathered stack on dst, None elsewhere."""
        for w in self.pending:
            w.wait()
        self.pending = []
        return self.out


def render_views_sharded(render_one, n_views, res, dst=0):
    """Render views [0, n_views) sharded over the ranks; `render_one(view_index) -> (feat, wsum)` for one view.
    Returns [n_views,4,res,res] on rank dst.  Ranks beyond n_views render nothing and still take part in the gather."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = partition(n_views, world, rank)
    frames = []
    for v in range(lo, hi):
        feat, wsum = render_one(v)
        frames.append(frames_rgba(feat, wsum, res))
    counts = [partition(n_views, world, r)[1] - partition(n_views, world, r)[0] for r in range(world)]
    if frames:
        local = torch.cat(frames)
    else:  # more ranks than views: an empty stack of the right trailing shape (device: the process's current accelerator, if any)
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        local = torch.empty((0, 4, res, res), dtype=torch.float32, device=dev)
    return gather_frames(local, counts, dst)
