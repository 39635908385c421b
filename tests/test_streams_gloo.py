"""world_size-2 `gloo` rehearsal on CPU of the multi-GPU launcher (orb-slam2_amd/streams.py): stream
ownership is a partition, the timed window is barrier-aligned, the reported time is the MAX over
ranks and the rate is whole-job."""
import os
import sys
import time

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    st = pkg.streams
    st.init("gloo", rank=rank, world=world)
    mine = st.owned_streams(rank, world, 2)
    delay = 0.02 * (rank + 1)          # rank 1 is slower: the MAX must be reported on every rank
    done = []
    elapsed = st.timed_steps(lambda: (time.sleep(delay), done.append(1)), 5, lambda: None, world)
    det = {}
    st.timed_steps(lambda: time.sleep(delay), 2, lambda: None, world, detail=det)
    ev = (st.ranks_seen(world), st.gather_strings(f"0000:0{rank}:00.0 uuid{rank} gfx950", world), det["per_rank"], det["own"],
          st.gather_floats(float(rank) + 0.5, world))
    q.put((rank, mine, [st.stream_seed(s) for s in mine], elapsed, len(done), ev))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_rank_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29650 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    streams = [s for r in res for s in r[1]]
    assert sorted(streams) == list(range(4)) and len(set(streams)) == 4           # partition, no overlap
    seeds = [s for r in res for s in r[2]]
    assert len(set(seeds)) == 4
    assert all(r[4] == 5 for r in res)                                             # exactly K steps each
    for r in res:       # the evidence every N-rank bench line carries: ranks that met, per-rank device strings and per-rank times, in rank order
        seen, devs, per_rank, own, fl = r[5]
        assert seen == 2 and devs == ["0000:00:00.0 uuid0 gfx950", "0000:01:00.0 uuid1 gfx950"] and fl == [0.5, 1.5]
        assert len(per_rank) == 2 and per_rank[r[0]] == own and per_rank[1] > per_rank[0] >= 2 * 0.02 * 0.95
    e0, e1 = res[0][3], res[1][3]
    assert abs(e0 - e1) < 1e-9 and e0 >= 5 * 0.04 * 0.95                            # both report the slow rank's time


def test_ownership_and_rate(pkg):
    st = pkg.streams
    assert st.owned_streams(0, 1) == [0]
    assert st.owned_streams(3, 8) == [3]
    assert [s for r in range(8) for s in st.owned_streams(r, 8, 3)] == list(range(24))
    with pytest.raises(ValueError):
        st.owned_streams(8, 8)
    assert st.aggregate_rate(64, 10, 8, 2.0) == 64 * 10 * 8 / 2.0
