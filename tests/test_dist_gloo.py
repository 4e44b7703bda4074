"""world_size-2 `gloo` rehearsal of the multi-GPU layer on CPU: frames are sharded across ranks
as independent streams, every rank runs the hot path on its shard (here: the CPU oracle stands in
for the GPU, it is the checker), and the velocity all-gather must reproduce the single-process
result row for row.  This is the N > 1 code path of bench.py minus the HIP kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, golden_files
from drone_oa_iree_vit_accelerator_amd import dist as itadist


def test_shard_range_partitions():
    for total in (0, 1, 7, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [itadist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        itadist.shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, steps, q):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from drone_oa_iree_vit_accelerator_amd import dist as itd, params, synth
    from oracle import oracle
    r, lr, w = itd.init("gloo")
    assert (r, w) == (rank, world)
    d = params.load_fixture(golden_files("vitlstm_E64_seed0_B2.npz")[0])
    blob = params.blob_from_record(d, synth.float_params(0), E=64)
    fr = synth.frames(99, total)
    lo, hi = itd.shard_range(total, rank, world)
    n = hi - lo
    # strong-scaling form: a global batch `total` cut into contiguous shards that may differ by one frame
    gather = itd.VelocityGather(n, world, torch.device("cpu"), total=total)
    h = np.zeros((3, n, 128), np.float32)
    c = np.zeros((3, n, 128), np.float32)
    outs = []
    pending = None
    for s in range(steps):                       # state stays on the owning rank, only velocities travel
        vel, h, c = oracle.forward(blob, fr["img_u8"][lo:hi], fr["desvel"][lo:hi] + s, fr["quat"][lo:hi], h, c)
        if pending is not None:
            outs.append(gather.result(pending).clone())
        pending = gather.start(torch.from_numpy(vel))
    outs.append(gather.result(pending).clone())
    gather.finish()
    dist.barrier()
    if rank == 0:
        q.put([o.numpy() for o in outs])
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5], ids=["even_shards", "uneven_shards"])
def test_two_rank_gather_matches_single_process(oracle, total):
    from drone_oa_iree_vit_accelerator_amd import params, synth
    steps, world = 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = params.load_fixture(golden_files("vitlstm_E64_seed0_B2.npz")[0])
    blob = params.blob_from_record(d, synth.float_params(0), E=64)
    fr = synth.frames(99, total)
    h = np.zeros((3, total, 128), np.float32)
    c = np.zeros((3, total, 128), np.float32)
    for s in range(steps):
        vel, h, c = oracle.forward(blob, fr["img_u8"], fr["desvel"] + s, fr["quat"], h, c)
        np.testing.assert_array_equal(got[s], vel)       # rank order == stream order, bit for bit


def _group_worker(rank, world, port, total, ngroup, groups, q):
    """the graph / --gather-every 8 schedule of bench.py on CPU: every rank owns ONE velocity buffer of ngroup steps x its
    shard that it overwrites for every group (as a captured HIP graph does), the gather is built with explicit per-rank
    sizes (uneven shards padded to the longest) and stage=True (its own ping-pong send buffers)"""
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from drone_oa_iree_vit_accelerator_amd import dist as itd
    itd.init("gloo")
    shard = [hi - lo for lo, hi in (itd.shard_range(total, r, world) for r in range(world))]
    lo, hi = itd.shard_range(total, rank, world)
    n = hi - lo
    gather = itd.VelocityGather(ngroup * n, world, torch.device("cpu"), sizes=[ngroup * s for s in shard], stage=True)
    vel = torch.empty((ngroup, n, 3))                       # the one buffer
    outs, pending = [], None
    for g in range(groups):
        for k in range(ngroup):                             # "replay": overwrite the buffer while the last all-gather may run
            step = g * ngroup + k
            vel[k] = (torch.arange(lo, hi, dtype=torch.float32).reshape(n, 1) * 1000 + step) + torch.tensor([0.0, 0.25, 0.5])
        if pending is not None:
            outs.append(gather.result(pending).clone())
        pending = gather.start(vel.reshape(ngroup * n, 3))
    outs.append(gather.result(pending).clone())
    gather.finish()
    dist.barrier()
    if rank == 0:
        q.put([o.numpy() for o in outs])
    dist.destroy_process_group()


@pytest.mark.parametrize("total,world", [(5, 2), (8, 2), (7, 3)], ids=["uneven2", "even2", "uneven3"])
def test_grouped_staged_gather_with_uneven_shards(total, world):
    ngroup, groups = 4, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_worker, args=(r, world, port, total, ngroup, groups, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == groups
    for g in range(groups):
        rows = []
        for r in range(world):                              # rank-major, then step, then stream: what each rank sent
            lo, hi = itadist.shard_range(total, r, world)
            for k in range(ngroup):
                step = g * ngroup + k
                rows.append(np.arange(lo, hi, dtype=np.float32).reshape(-1, 1) * 1000 + step + np.array([0.0, 0.25, 0.5], np.float32))
        np.testing.assert_array_equal(got[g], np.concatenate(rows, 0))


def test_gather_rejects_bad_sizes():
    with pytest.raises(ValueError):
        itadist.VelocityGather(4, 2, torch.device("cpu"), sizes=[4])
