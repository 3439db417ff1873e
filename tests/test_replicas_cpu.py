"""not gpu: the N > 1 bookkeeping of bench.py (independent replicas, aggregate rate =
tokens of all ranks / slowest rank) on the gloo backend, world_size 2."""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from __graft_entry__ import load_package
    pkg = load_package()
    from importlib import import_module
    rep = import_module(pkg.__name__ + ".replicas")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert rep.rank_info() == (rank, rank, world)
    mine = rep.shard_prompts(5, rank, world)
    seed = rep.prompt_seed(12345, rank)

    def run():
        time.sleep(0.05 * (rank + 1))            # rank 1 is the slow replica
        return 10 * len(mine)

    elapsed, tokens = rep.timed_region(run, lambda: None, dist=dist)
    q.put((rank, mine, seed, elapsed, tokens))
    dist.destroy_process_group()


def test_two_replicas_aggregate_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, p0, s0, e0, t0), (r1, p1, s1, e1, t1) = res
    assert p0 == [0, 2, 4] and p1 == [1, 3]            # every prompt exactly once
    assert s0 != s1                                    # different prompt streams
    assert t0 == t1 == 50                              # SUM over ranks
    assert abs(e0 - e1) < 1e-9 and e0 >= 0.1           # MAX over ranks: the slow replica's time
