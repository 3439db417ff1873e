"""not gpu: `python bench.py --gpus N` really starts N replicas (round-1 verdict: the flag was parsed and ignored).

bench.py runs here with `--engine stub`: a sleeping stand-in for the decoder, NO GPU and no arithmetic, so what these
tests exercise is the launcher (tinyllama.cpp_amd/replicas.py launch(): one worker per GPU, the environment
torch.distributed.run would set), the rendezvous (gloo), the barrier-bracketed timed region with MAX over ranks, the
per-rank / solo / efficiency bookkeeping and the ONE JSON line of rank 0 -- the same code the GPU run takes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(*extra, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, "--engine", "stub", "--steps", "12", "--warmup", "2", *extra],
                          capture_output=True, text=True, env=e, timeout=timeout)


def result_line(proc):
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout + proc.stderr          # ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 4])
def test_gpus_n_starts_n_replicas(n):
    p = run_bench("--gpus", str(n))
    assert p.returncode == 0, p.stderr[-2000:]
    r = result_line(p)
    assert r["n_gpus"] == n and r["steps"] == 12 and r["warmup"] == 2
    assert r["metric"].startswith("STUB") and f"{n} GPUs" in r["metric"]
    assert [x["rank"] for x in r["per_rank"]] == list(range(n))
    # value = tokens of all ranks / the slowest rank's time: the stub's higher ranks are slower, so the aggregate is
    # below the sum of the per-rank rates and at least n x the slowest rank's rate
    rates = [x["tok_s"] for x in r["per_rank"]]
    assert min(rates) * n <= r["value"] * 1.02 and r["value"] <= sum(rates) * 1.02
    assert rates[0] >= rates[-1]
    assert abs(r["value"] - 12 * n / (r["ms_per_step"] * 12e-3)) / r["value"] < 0.02
    assert 0.3 < r["efficiency"] <= 1.1 and abs(r["efficiency"] - r["value"] / (n * r["solo_rank0_tok_s"])) < 1e-3
    assert r["launcher"]["kind"].startswith("bench.py")
    assert r["config"]["parallelism"] == f"replicas x{n}"


def test_gpus_1_line_is_the_single_replica_line():
    p = run_bench("--gpus", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    r = result_line(p)
    assert r["n_gpus"] == 1 and "per_rank" not in r and "efficiency" not in r and "launcher" not in r
    assert "1 GPU;" in r["metric"]


def test_a_failing_replica_fails_the_run():
    p = run_bench("--gpus", "2", env={"GTEN_BENCH_STUB_FAIL_RANK": "1"})
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_curve_reports_smaller_replica_counts():
    p = run_bench("--gpus", "4", "--curve")
    assert p.returncode == 0, p.stderr[-2000:]
    r = result_line(p)
    assert [c["n_gpus"] for c in r["scaling_curve"]] == [1, 2, 4]
    assert r["scaling_curve"][-1]["value"] == r["value"]
    assert r["scaling_curve"][0]["efficiency_vs_1gpu_run"] == 1.0


def test_same_worker_under_torch_distributed_run():
    """the driver's own form: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N"""
    sys.path.insert(0, os.path.join(ROOT, "tinyllama.cpp_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("gten_replicas_t", os.path.join(ROOT, "tinyllama.cpp_amd", "replicas.py"))
    rep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rep)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(rep.free_port()), BENCH, "--gpus", "2",
                        "--engine", "stub", "--steps", "12", "--warmup", "2"], capture_output=True, text=True, env=e, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    r = result_line(p)
    assert r["n_gpus"] == 2 and len(r["per_rank"]) == 2 and "launcher" not in r
