"""Replica (per-prompt) sharding across GPUs: the path has no exchange step, so N GPUs
are N independent decoders, one process per GPU, each with its own weight copy and
K/V caches ("replicas only").  The only cross-rank traffic is the
bookkeeping below -- a barrier and two tiny reductions for the aggregate rate --
on whatever torch.distributed backend the launcher set up (RCCL on the GPU box,
gloo in the CPU tests).  Nothing here touches the data path."""
import os
import time


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def prompt_seed(base_seed, rank):
    """each replica decodes its own synthetic prompt stream"""
    return base_seed + rank


def shard_prompts(n_prompts, rank, world):
    """round-robin assignment of independent prompts to replicas"""
    return list(range(rank, n_prompts, world))


def timed_region(run_steps, sync, dist=None, device=None):
    """barrier + sync, run, sync + barrier; returns (max elapsed over ranks, total tokens)."""
    import torch
    sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    tokens = run_steps()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        k = torch.tensor([float(tokens)], dtype=torch.float64, device=device)
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
        dist.barrier()
        return float(t.item()), int(k.item())
    return elapsed, int(tokens)
