"""world_size-2 gloo tests (CPU) of the N > 1 path: stream sharding, the
benchmark's barrier + MAX-over-ranks timing reduction, and the warm-start
broadcast of a background model.  The data path itself has no collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from pygofpid_amd import distributed as gd


def test_shard_streams_partition():
    for n, world in [(8, 1), (8, 2), (8, 8), (33, 8), (3, 8), (0, 4)]:
        parts = [gd.shard_streams(n, world, r) for r in range(world)]
        flat = sorted(s for p in parts for s in p)
        assert flat == list(range(n))                      # every stream exactly once
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
        for r, p in enumerate(parts):
            assert all(s % world == r for s in p)
    with pytest.raises(ValueError):
        gd.shard_streams(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w = gd.init(backend="gloo")
    assert (r, w) == (rank, world)
    gd.barrier()
    # timing reduction: the slowest rank defines the step time
    slow = gd.max_over_ranks(1.0 + rank)
    total = gd.sum_over_ranks(10.0 * (rank + 1))
    # warm start: rank 0 owns a "trained" model, the others start from zeros
    rng = np.random.default_rng(5)
    model = rng.integers(0, 256, 4096, dtype=np.uint8)
    state = torch.from_numpy(model.copy() if rank == 0 else np.zeros_like(model))
    n = gd.broadcast_model(state, 137 if rank == 0 else 0, src=0)
    mine = gd.shard_streams(5, world, rank)
    out.put((rank, slow, total, n, bool(np.array_equal(state.numpy(), model)), mine))
    gd.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, slow, total, n, same, mine in res:
        assert slow == 2.0 and total == 30.0
        assert n == 137 and same
        assert mine == list(range(rank, 5, world))


def test_single_process_helpers_are_noops():
    assert gd.max_over_ranks(3.5) == 3.5
    t = torch.arange(4, dtype=torch.uint8)
    assert gd.broadcast_model(t, 9) == 9
    gd.barrier()
