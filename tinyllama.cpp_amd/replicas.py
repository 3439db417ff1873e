"""Replica (per-prompt) sharding across GPUs: the path has no exchange step, so N GPUs
are N independent decoders, one process per GPU, each with its own weight copy and
K/V caches ("replicas only").  The only cross-rank traffic is the
bookkeeping below -- a barrier and a tiny all-gather for the aggregate rate --
on whatever torch.distributed backend the launcher set up (RCCL on the GPU box,
gloo in the CPU tests).  Nothing here touches the data path.

`launch()` is the parent of `python bench.py --gpus N`: it starts one worker process
per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* / GTEN_HIP_DEVICE in the
environment, exactly what `python -m torch.distributed.run --nproc-per-node N` would
set), relays rank 0's output and fails if any worker fails.  The parent itself never
initialises the GPU and never exec()s."""
import os
import signal
import socket
import subprocess
import sys
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


def timed_region(run_steps, sync, dist=None, device=None, per_rank=False):
    """barrier + sync, run, sync + barrier; returns (max elapsed over ranks, total tokens)
    and, with per_rank=True, a third item: [(elapsed, tokens)] of every rank in rank order."""
    import torch
    sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    tokens = run_steps()
    sync()
    elapsed = time.perf_counter() - t0
    ranks = [(elapsed, int(tokens))]
    if dist is not None:
        mine = torch.tensor([elapsed, float(tokens)], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        dist.barrier()
        ranks = [(float(t[0].item()), int(t[1].item())) for t in every]
    out = (max(e for e, _ in ranks), sum(k for _, k in ranks))
    return out + (ranks,) if per_rank else out


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker_env(rank, world, port, base=None):
    """the environment torch.distributed.run gives rank `rank` of a one-node job"""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GTEN_HIP_DEVICE=str(rank),
               HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return env


def _stop(procs, grace=10.0):
    """end exactly the workers this launcher started (never by pattern)"""
    for p in procs:
        if p.poll() is None:
            p.send_signal(signal.SIGTERM)
    t_end = time.time() + grace
    for p in procs:
        while p.poll() is None and time.time() < t_end:
            time.sleep(0.05)
        if p.poll() is None:
            p.kill()
            p.wait()


def launch(n, worker_cmd, timeout=None, env=None, poll_s=0.05):
    """Start `n` copies of `worker_cmd` (a list), rank i with worker_env(i, n, port).
    Returns (exit code, rank 0's stdout as text).  Exit code 0 only when every worker
    returned 0; the first failing worker ends the others and its code is returned
    (124 on timeout).  Other ranks' stdout goes to this process's stderr."""
    port = free_port()
    procs = []
    try:
        for r in range(n):
            procs.append(subprocess.Popen(worker_cmd, env=worker_env(r, n, port, env),
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None, text=(r == 0)))
    except Exception:
        _stop(procs)
        raise
    t_end = None if timeout is None else time.time() + timeout
    out0, rc = "", 0
    # rank 0's pipe is drained by communicate() in a helper thread so that a full pipe never blocks it
    import threading
    box = {}
    th = threading.Thread(target=lambda: box.setdefault("out", procs[0].communicate()[0]), daemon=True)
    th.start()
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0] if bad[0] > 0 else 128 - bad[0]
            _stop(procs)
            break
        if all(c == 0 for c in codes):
            break
        if t_end is not None and time.time() > t_end:
            rc = 124
            _stop(procs)
            break
        time.sleep(poll_s)
    th.join(timeout=30)
    out0 = box.get("out") or ""
    return rc, out0
