"""Multi-GPU launcher logic: one process per GPU, each owning whole camera streams.

The hot path shards at stream granularity with NO data-path collective: a frame is
far too small to split, so rank r simply owns streams [r*k, (r+1)*k).  torch.distributed is used
only to align the timed window (barrier) and to take the MAX of the per-rank elapsed time.  The
backend is "nccl" (= RCCL over xGMI) on GPUs and "gloo" in the CPU tests.
"""
import os
import time


def owned_streams(rank, world, streams_per_rank=1):
    """ids of the camera streams rank `rank` owns (weak scaling: fixed work per rank)"""
    if not (0 <= rank < world) or streams_per_rank < 1:
        raise ValueError("bad rank/world")
    return list(range(rank * streams_per_rank, (rank + 1) * streams_per_rank))


def stream_seed(stream_id, base=1000, stride=97):
    """seed of the synthetic generator for one stream (distinct streams -> distinct images)"""
    return base + stride * stream_id


def init(backend, rank=None, world=None, device_id=None):
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0)) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", 1)) if world is None else world
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {"device_id": device_id} if device_id is not None else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def timed_steps(step, steps, local_sync, world, device=None):
    """barrier + local sync, run `steps` steps, local sync + barrier; returns MAX-over-ranks seconds"""
    import torch
    import torch.distributed as dist

    def fence():
        local_sync()
        if world > 1:
            dist.barrier()
        local_sync()

    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def aggregate_rate(units_per_rank_per_step, steps, world, elapsed_max):
    """whole-job throughput: all units processed by all ranks / slowest rank's time"""
    return units_per_rank_per_step * steps * world / elapsed_max
