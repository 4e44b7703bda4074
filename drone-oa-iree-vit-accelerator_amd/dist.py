"""Multi-GPU layer: one process per GPU, frames sharded as independent streams.

The reference has no distributed code at all (IREE reports `Collectives: 0`,
output/compilation_info.txt:11); the only exchange this engine adds is the one the task
names: a gather of the (frames, 3) velocity outputs.  Each stream's LSTM state (h, c) stays
on the GPU that owns the stream and is never communicated.  The message is 12 bytes per frame
(12 KiB per rank at 1024 frames) -- latency-bound on xGMI, so it is one all-gather per step
(a single hop on the fully connected fabric), issued asynchronously so that it overlaps the
next step's kernels.

Works with backend "nccl" (= RCCL on ROCm) on GPU tensors and "gloo" on CPU tensors (tests).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of `total` streams for `rank`; sizes differ by at most one."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError("bad rank/world/total")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init(backend: Optional[str] = None, device_index: Optional[int] = None):
    """Initialises torch.distributed from the environment when WORLD_SIZE > 1."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world == 1 or dist.is_initialized():
        return rank, local_rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", local_rank if device_index is None else device_index)
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


class VelocityGather:
    """Double-buffered asynchronous all-gather of per-rank (frames, 3) velocity tensors.

    Equal shards (`frames_per_rank` on every rank: bench.py's weak-scaling form) or, with `total`, the uneven
    contiguous shards of shard_range(total, rank, world) (strong scaling: a global batch cut over the ranks): every
    rank then sends max-shard rows (its tail rows are padding) and `result` returns the `total` valid rows in stream
    order.  `sizes` gives the rows of every rank explicitly (a group of several steps of uneven shards: rank r sends
    steps x shard_r rows).  `start` issues the collective for this step and returns immediately; `result` waits for it.

    `stage=True`: `start` first copies the caller's tensor into one of this object's own two send buffers (a
    stream-ordered device copy), so the caller may overwrite its tensor as soon as `start` has returned -- for callers
    with ONE velocity buffer (a captured HIP graph writes the same buffer on every replay).  Without it the caller
    ping-pongs two buffers and calls `ready` before reusing one."""

    def __init__(self, frames_per_rank: int, world: int, device, dtype=None, total: Optional[int] = None,
                 sizes=None, stage: bool = False):
        import torch
        self.world = world
        dtype = dtype or torch.float32
        if sizes is not None:
            if len(sizes) != world or min(sizes) < 0:
                raise ValueError("sizes: one non-negative row count per rank")
            self.sizes = [int(s) for s in sizes]
        elif total is None:
            self.sizes = [frames_per_rank] * world
        else:
            self.sizes = [hi - lo for lo, hi in (shard_range(total, r, world) for r in range(world))]
        self.n = max(self.sizes) if self.sizes else 0
        self.even = all(sz == self.n for sz in self.sizes)
        self.stage = bool(stage)
        self.bufs = [torch.empty((world * self.n, 3), dtype=dtype, device=device) for _ in range(2)]
        self.pad = (None if self.even and not self.stage
                    else [torch.zeros((self.n, 3), dtype=dtype, device=device) for _ in range(2)])
        self.handles = [None, None]
        self.i = 0

    def ready(self):
        """Call before overwriting the `vel` tensor that was passed to `start` two steps ago (callers that
        ping-pong two velocity buffers in step with this object): its all-gather may still be reading it."""
        if self.handles[self.i] is not None:
            self.handles[self.i].wait()
            self.handles[self.i] = None

    def start(self, vel):
        import torch.distributed as dist
        i = self.i
        if self.handles[i] is not None:       # buffer about to be reused: its collective must be done
            self.handles[i].wait()
        if vel.shape[0] != self.n or self.stage:   # a short shard: send it padded to the longest one (stage: always copy)
            if self.pad is None or vel.shape[0] > self.n:
                raise ValueError(f"velocity tensor has {vel.shape[0]} rows, the gather was built for {self.sizes}")
            self.pad[i][:vel.shape[0]].copy_(vel)
            vel = self.pad[i]
        if self.world == 1:
            self.bufs[i].copy_(vel)
            self.handles[i] = None
        elif dist.get_backend() == "gloo" and vel.is_cuda:
            # rehearsal only (several ranks sharing one GPU, where RCCL refuses duplicate devices):
            # stage through host memory, synchronously
            import torch
            host_out = torch.empty(self.bufs[i].shape, dtype=vel.dtype)
            dist.all_gather_into_tensor(host_out, vel.cpu().contiguous())
            self.bufs[i].copy_(host_out)
            self.handles[i] = None
        else:
            self.handles[i] = dist.all_gather_into_tensor(self.bufs[i], vel.contiguous(), async_op=True)
        self.i ^= 1
        return i

    def result(self, i: int):
        if self.handles[i] is not None:
            self.handles[i].wait()
            self.handles[i] = None
        if self.even:
            return self.bufs[i]
        import torch
        return torch.cat([self.bufs[i][r * self.n:r * self.n + sz] for r, sz in enumerate(self.sizes)], 0)

    def finish(self):
        for i in (0, 1):
            if self.handles[i] is not None:
                self.handles[i].wait()
                self.handles[i] = None
