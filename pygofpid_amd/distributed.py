"""Multi-GPU layout of the foreground path: shard by STREAM, one process per GPU.

MOG2 and frame differencing are temporal recurrences (frame t needs the state
left by frame t-1), so the frames of one video cannot be split across GPUs;
independent videos can -- the reference already builds one GOFPID per video
(examples/dataset_ilids.py:126-140).  Stream s goes to rank s mod world_size
and there is NO collective on the steady-state data path.  torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) is used
only for
  * the barrier / MAX-over-ranks of the benchmark timing, and
  * an optional warm start: broadcasting one trained background model to the
    replicas that will watch the same scene (one contiguous buffer, ~1.7 GB for
    8 x 1080p streams -> a single large broadcast, sized for xGMI's per-link
    bandwidth rather than many small ones).
"""
import os


def shard_streams(n_streams, world_size, rank):
    """Global stream ids owned by `rank` (round-robin, as SURVEY.md 8(e))."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank %r of %r" % (rank, world_size))
    return list(range(rank, n_streams, world_size))


def env_rank():
    """(rank, world_size, local_rank) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None, device=None):
    """Join the process group if WORLD_SIZE > 1.  Returns (rank, world)."""
    rank, world, _ = env_rank()
    if world == 1 and not os.environ.get("GOF_DIST_FORCE"):
        return 0, 1      # (GOF_DIST_FORCE=1: a one-rank group, to rehearse the RCCL calls on one GPU)
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            import torch
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX of a python float over all ranks (the benchmark's elapsed time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_model(state, nframes, src=0):
    """Warm start: replicate rank `src`'s background model.

    `state` is a 1-D uint8 torch tensor holding the whole model (device tensor
    filled by ForegroundPipeline.state_to_buffer on the GPU box; a CPU tensor in
    the gloo tests).  It is overwritten in place on the other ranks.  Returns
    the source's frame counter, which sets the learning rate 1/min(2t, history).
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return int(nframes)
    dist.broadcast(state, src=src)
    n = torch.tensor([int(nframes)], dtype=torch.int64, device=state.device)
    dist.broadcast(n, src=src)
    return int(n.item())


def warm_start(pipe, src=0):
    """Broadcast the model of `pipe` (a MOG2 ForegroundPipeline) from rank src.

    Every rank must hold a pipeline of the same geometry and nmixtures, built against the same
    library ABI (the raw model layout is the library's): these are compared over the group first (a mismatch would hang or corrupt the
    broadcast) and ValueError is raised on every rank if they differ.  With one
    process this is a copy of the model through the staging buffer."""
    import torch
    import torch.distributed as dist
    _, nbytes = pipe.mog2_device_state()
    dev = torch.device("cuda", pipe.device)
    if dist.is_available() and dist.is_initialized():
        from . import _lib
        # (the library's ABI version too: the raw model layout belongs to it)
        mine = torch.tensor([nbytes, pipe.streams, pipe.h, pipe.w, pipe.c, pipe.nmixtures,
                             _lib.load().gof_abi_version()], dtype=torch.int64, device=dev)
        every = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        if any(not torch.equal(e, every[src]) for e in every):
            raise ValueError("warm_start: ranks hold different models: %r"
                             % [e.tolist() for e in every])
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pipe.state_to_buffer(buf.data_ptr())
    n = broadcast_model(buf, pipe.nframes(), src=src)
    pipe.state_from_buffer(buf.data_ptr(), n)
    return n
