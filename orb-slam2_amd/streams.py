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


def _collective(world):
    """collectives run when there is more than one rank -- or when a process group exists at all (a one-rank group made on purpose: the
    rehearsal of the "nccl" = RCCL branch on a one-GPU box, ORBX_BENCH_FORCE_DIST=1)"""
    if world > 1:
        return True
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def init(backend, rank=None, world=None, device_id=None, force=False):
    """force: make the process group even for a single rank (every collective below then really runs through the backend)"""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0)) if rank is None else rank
    world = int(os.environ.get("WORLD_SIZE", 1)) if world is None else world
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {"device_id": device_id} if device_id is not None else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def timed_steps(step, steps, local_sync, world, device=None, detail=None):
    """barrier + local sync, run `steps` steps, local sync + barrier; returns MAX-over-ranks seconds.
    `detail` (a dict) additionally receives this rank's own seconds under "own" and, for world > 1, every rank's under "per_rank"."""
    import torch
    import torch.distributed as dist

    coll = _collective(world)

    def fence():
        local_sync()
        if coll:
            dist.barrier()
        local_sync()

    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    local_sync()
    own = time.perf_counter() - t0          # this rank's own work, before it waits for the others: a straggler shows in the per-rank list
    fence()
    elapsed = time.perf_counter() - t0
    if detail is not None:
        detail["own"] = own
        detail["per_rank"] = gather_floats(own, world, device)
    if coll:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def gather_floats(x, world, device=None):
    """every rank's value of `x`, in rank order, on every rank (all-gather of one float64)"""
    if not _collective(world):
        return [float(x)]
    import torch
    import torch.distributed as dist
    mine = torch.tensor([x], dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def ranks_seen(world, device=None):
    """SUM-all-reduce of a one per rank: equals `world` exactly when every rank of the launch reached this point"""
    if not _collective(world):
        return 1
    import torch
    import torch.distributed as dist
    t = torch.ones(1, dtype=torch.int32, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_strings(s, world, device=None, width=160):
    """every rank's string, in rank order (all-gather of fixed-width byte rows): the per-rank device identities of an N-GPU line"""
    if not _collective(world):
        return [s]
    import torch
    import torch.distributed as dist
    raw = s.encode()[:width].ljust(width, b"\0")
    mine = torch.tensor(list(raw), dtype=torch.uint8, device=device if device is not None else "cpu")
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [bytes(t.cpu().tolist()).rstrip(b"\0").decode(errors="replace") for t in out]


def aggregate_rate(units_per_rank_per_step, steps, world, elapsed_max):
    """whole-job throughput: all units processed by all ranks / slowest rank's time"""
    return units_per_rank_per_step * steps * world / elapsed_max
