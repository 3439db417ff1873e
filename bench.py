#!/usr/bin/env python3
"""bench.py -- decode throughput of TinyLlama-1.1B on MI355X through the gten_hip path.

Metric (BASELINE.json): decode tok/s, TinyLlama-1.1B q4, ctx = 2048, one
MI355X per replica; achieved HBM GB/s vs peak.  One "step" = one decoded token
= one pass of the hot path (embedding row, 22 blocks, final norm, lm_head,
greedy argmax) over a context that ENDS at n = 2048: the context is first filled
by real single-token decode steps (untimed), then W warm-up steps, then exactly
K timed steps bracketed by barrier + device synchronise.  Synthetic weights
(seeded generator, quantized with the reference converter's rules) and
teacher-forced synthetic token ids: there is no network for checkpoints.

N > 1: independent replicas, one process and one prompt stream per GPU, no
collective on the data path; `value` is the sum over ranks of
tokens decoded / the slowest rank's time.  Either launcher works:
  python bench.py --gpus N ...                          (this file starts N workers itself:
                                                         tinyllama.cpp_amd/replicas.py launch())
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
The N > 1 line also carries every rank's own rate, rank 0's rate measured alone
in the same run (the other GPUs idle) and efficiency = value / (N x that rate);
`--curve` (own launcher only) first runs the smaller power-of-two replica counts
and adds their values as `scaling_curve`.

Output: ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event
timed on the library's stream) and `cpu_baseline` (the reference's own
AVX/OpenMP build from oracle/_ref when present, else the oracle port; bounded
sample) objects.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MODES = {"q4": (4, 3), "q8": (3, 3), "f16": (1, 1)}   # (wdtype, adtype), tinyllama.cpp:258-265
N_CTX = 2048


def weight_bytes_per_elt(mode):
    return {"q4": 18 / 32, "q8": 34 / 32, "f16": 2.0}[mode]


def algorithmic_bytes(mode, n):
    """B(n) of SURVEY 8(d): every linear weight once + norm vectors + K/V rows [0,n)."""
    elts = 1_034_426_368                       # 22*(2*2048^2 + 2*256*2048 + 3*5632*2048) + 32003*2048
    w = elts * weight_bytes_per_elt(mode)
    nw = 45 * 2048 * 2
    kv_row = 2 * 256 * 22 * (2.0 if mode == "f16" else 34 / 32)
    return w + nw + kv_row * n


def linear_shapes():
    """(family, d_out, d_in, launches per token) of the W.x contractions."""
    return [("q", 2048, 2048, 22), ("k", 256, 2048, 22), ("v", 256, 2048, 22), ("o", 2048, 2048, 22),
            ("gate", 5632, 2048, 22), ("up", 5632, 2048, 22), ("down", 2048, 5632, 22), ("lm_head", 32003, 2048, 1)]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--mode", choices=sorted(MODES), default="q4")
    ap.add_argument("--path", choices=["auto", "ops", "fused"], default="auto",
                    help="ops: one kernel per gten operator; fused: the decode fast path (default when available)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--wide-streams", type=int, default=64, choices=[0, 16, 32, 48, 64, 128, 192, 256, 384, 512],
                    help="also report the aggregate rate of this many sequences on the matrix-core decode path (0: skip)")
    ap.add_argument("--streams", type=int, default=8, choices=[0, 2, 4, 8, 16, 32, 48, 64],
                    help="also measure this many sequences sharing each weight pass on the GPU (0 = skip); "
                         "reported separately, `value` stays the single-sequence rate")
    ap.add_argument("--prefill", type=int, default=512, help="also time a prompt of this many ids (0 = skip)")
    ap.add_argument("--generate", type=int, default=256, help="also time real greedy generation of this many ids ending at the context limit (0 = skip)")
    ap.add_argument("--serve", type=int, default=1024, help="also serve a queue of this many synthetic prompts through the serving slots (0 = skip)")
    ap.add_argument("--serve-slots", type=int, default=256, choices=[0, 16, 32, 48, 64, 128, 192, 256, 384, 512],
                    help="slots of the serving leg (0: --wide-streams); measured, 1024 prompts (its first 256 alone): 64 slots 27.2k (24.5k) new ids/s, "
                         "128 34.7k (28.1k), 256 37.0k (27.5k), 384 26.4k, 512 22.1k")
    ap.add_argument("--serve-slice", type=int, default=8, help="shared steps per slice of the serving leg")
    ap.add_argument("--serve-spares", type=int, default=-1,
                    help="cache sets the serving leg fills ahead of the slots that will take them (-1: the library's default, a quarter of the slots, at most 64; 0: none)")
    ap.add_argument("--serve-ramp", type=int, default=-1, help="serving: percent of the slots that hold a processed prompt before a queue's first slice (-1: the library's default, 100)")
    ap.add_argument("--ctx", type=int, default=N_CTX, help="context length the timed steps end at (metric: 2048)")
    ap.add_argument("--fill", choices=["decode", "prefill"], default="decode",
                    help="how the (untimed) context below the timed window is produced: single-token decode steps "
                         "(default) or one prompt-processing call -- the latter keeps a rocprofv3 kernel trace to "
                         "thousands instead of hundreds of thousands of launches")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (counter collection)")
    ap.add_argument("--no-lanes", action="store_true", help="skip the 256-sequence (lanes) leg")
    ap.add_argument("--lane-skip", action="store_true", help="serving: a shared run leaves out lanes whose slots are all parked (A/B; measured slower)")
    ap.add_argument("--kv-rows", action="store_true", help="wide decoders read the K / V cache rows as they lie (round 4's attention kernel) instead of "
                    "head-major shadows (A/B of csrc/gten_decode_attn_hm.h)")
    ap.add_argument("--no-wx-planes", action="store_true", help="f16 wide decoders: o and down as k_dec_mmv_f16 in two K planes instead of k_dec_wxp_f16 in eight (A/B)")
    ap.add_argument("--brief", action="store_true", help="only the metric line: no secondary legs, no CPU baseline")
    ap.add_argument("--curve", action="store_true",
                    help="with --gpus N > 1 under this file's own launcher: first run 1, 2, 4, ... < N replicas (brief) and "
                         "report their values as `scaling_curve` in the N-replica line")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="own launcher: seconds before the workers are ended")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a one-GPU box: every replica decodes on GPU 0 and the bookkeeping runs on gloo "
                         "(RCCL refuses two ranks on one device); the line says so -- not a scaling measurement")
    ap.add_argument("--engine", choices=["hip", "stub"], default="hip",
                    help="stub: NO GPU work at all -- a sleeping stand-in for the decoder, used only by the CPU tests of the "
                         "launcher / rendezvous / JSON assembly (tests/test_bench_launcher_cpu.py); its line says so")
    return ap.parse_args(argv)


def code_only(text):
    """C / C++ source without its comments and with runs of white space collapsed (string and character literals kept as they
    are): what csrc_fingerprint() hashes, so that a note added to a kernel does not orphan the counters collected on it"""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":                                       # a literal: copied up to its closing quote
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1]); i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i); i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2); i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c); i += 1
    return " ".join("".join(out).split())


def usable_cpus():
    """the cores this process may really use: the affinity mask, capped by the cgroup's CPU quota (a GPU box reports every core
    of the host in os.cpu_count() while its container is held to a share of them)"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 8
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def csrc_fingerprint():
    """sha256 (16 hex digits) over what the in-tree libgten_hip.so is built from -- the kernel sources (comments and white
    space aside: code_only) AND the compiler flags (build.HIP_FLAGS, HIP_FILE_FLAGS: round 2 changed code generation of every
    kernel by a flag change alone): what ties profiles/traffic.json (PMC counters collected by tools/collect_all.sh) to
    the kernels being benched -- the GPU box has no .git, so a hash rather than a commit id"""
    import hashlib
    from __graft_entry__ import load_package
    b = load_package().build
    h = hashlib.sha256()
    d = os.path.join(ROOT, "tinyllama.cpp_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(code_only(open(os.path.join(d, name), "r", encoding="utf-8", errors="replace").read()).encode())
    flags = [f for f in b.HIP_FLAGS if not f.startswith("-I")]          # (-I carries the checkout's absolute path)
    h.update(repr((flags, sorted(b.HIP_FILE_FLAGS.items()))).encode())
    return h.hexdigest()[:16]


def measured_traffic(mode, family):
    """HBM bytes per launch of `family` from the rocprofv3 PMC passes, or (None, why) when the committed counters
    were not collected on these kernel sources"""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, "profiles/traffic.json absent"
    try:
        t = json.load(open(tpath))
    except Exception as e:
        return None, "profiles/traffic.json unreadable: %r" % (e,)
    if t.get("csrc_sha256_16") != csrc_fingerprint():
        return None, "profiles/traffic.json was collected on other kernel sources (%s, now %s): stale, not reported" % (
            t.get("csrc_sha256_16"), csrc_fingerprint())
    v = t.get(mode, {}).get(family)
    return v, (t.get("source") if v is not None else "no counters for %s / %s in profiles/traffic.json" % (mode, family))


def measured_counters(section):
    """profiles/counters.json (tools/collect_all.sh: PMC passes of the secondary legs' kernels), or (None, why) when it
    is absent or was collected on other kernel sources"""
    cpath = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(cpath):
        return None, "profiles/counters.json absent"
    try:
        t = json.load(open(cpath))
    except Exception as e:
        return None, "profiles/counters.json unreadable: %r" % (e,)
    if t.get("csrc_sha256_16") != csrc_fingerprint():
        return None, "profiles/counters.json was collected on other kernel sources (%s, now %s): stale, not reported" % (
            t.get("csrc_sha256_16"), csrc_fingerprint())
    v = t.get(section)
    return v, (t.get("source") if v is not None else "no %s section in profiles/counters.json" % section)


MFMA_F16_PEAK_TFLOPS = 2500.0       # dense fp16 / bf16 matrix peak of MI355X (MI355X_MICROARCH.md; the 5 PF figure is 2:1 sparsity)


def prefill_roofline(args, P, flops, wx_ms):
    """the prompt GEMM (k_matmul_mfma) against the dense f16 matrix peak: achieved = linear-layer FLOPs (lm_head: last row only)
    / the event-bracketed time of the W.x launches; mfma_busy_frac = the PMC share of matrix-pipe cycles (2048-id prompt)"""
    if wx_ms <= 0:
        return None
    ach = (flops - 2.0 * 32003 * 2048 * (P - 1)) / (wx_ms * 1e-3) / 1e12
    cnt, why = measured_counters("prefill2048") if args.mode == "q4" else (None, "counters are collected for q4")
    gemm = (cnt or {}).get("k_matmul_mfma", {})
    return {"bound": "mfma", "kernel": "k_matmul_mfma", "achieved": round(ach, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / MFMA_F16_PEAK_TFLOPS, 4), "mfma_busy_frac": gemm.get("mfma_busy_frac"),
            "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) over the GEMM's dispatches of a 2048-id prompt",
            "attention_mfma_busy_frac": (cnt or {}).get("k_attn_tiled", {}).get("mfma_busy_frac"),
            "traffic": None, "counters_source": why}


def dropin_leg(host, cfg, args, toks, hip=None):
    """greedy generation to n = 2048 through oracle/_ref/libdropin.so = the reference's translation unit on this
    repository's gten API (what a user of the reference gets by swapping the gten/ directory)"""
    from oracle import orc
    lib = orc.load_dropin()
    if lib is None:
        raise RuntimeError("oracle/_ref/libdropin.so not built (needs /root/reference at build time)")
    wd, ad = MODES[args.mode]
    path = f"/tmp/gten_bench_{args.mode}_{args.seed}.gten"
    if not os.path.exists(path) or os.path.getsize(path) < 1000:
        host.write_gten(cfg, args.seed, path + ".tmp")
        os.replace(path + ".tmp", path)
    G = min(args.generate, N_CTX - 16)
    P0 = N_CTX - G
    res = {}
    for name, fused_rows in (("fused_rows", True), ("operators", False), ("fused_rows_exact", True)):
        exact = name.endswith("_exact")
        if exact and hip is None:
            continue
        lib.set_fused_rows(fused_rows)
        if exact:
            hip.set_decode_exact(True)                         # (read when the model's decoder is made: two-launch attention, whole-row statistics)
        m = lib.tinyllama(N_CTX, wd, ad)
        m.load(path)
        new = G if (fused_rows and not exact) else min(G, 32)  # the operator-by-operator loop is ~15x slower: a short sample
        m.logits(toks[:P0], 0)                                 # warm-up: first-use allocations, weight repack
        m.logits(toks[:P0 + 1], P0)                            # ... and the decoder / its graph
        t0 = time.perf_counter(); m.logits(toks[:P0], 0); t_pre = time.perf_counter() - t0
        t0 = time.perf_counter()
        ids = m.greedy(toks[:P0], P0 + new)
        dt = time.perf_counter() - t0
        res[name] = {"new_tokens": int(len(ids) - P0), "tok_s": round((len(ids) - P0) / max(dt - t_pre, 1e-9), 1),
                     "prefill_ms": round(t_pre * 1e3, 2), "wall_ms": round(dt * 1e3, 2), "ids_head": [int(x) for x in ids[P0:P0 + 8]]}
        m.close()
        if exact:
            hip.set_decode_exact(False)
    lib.set_fused_rows(True)
    a, b = res["fused_rows"]["ids_head"], res["operators"]["ids_head"]
    extra = {}
    if "fused_rows_exact" in res:
        c = res["fused_rows_exact"]["ids_head"]
        extra["first_ids_agree_exact"] = next((i for i in range(8) if c[i] != b[i]), 8)
    return {"prompt_tokens": P0, "tok_s": res["fused_rows"]["tok_s"], **res, **extra,
            "first_ids_agree": next((i for i in range(8) if a[i] != b[i]), 8),
            "note": "reference's unmodified TinyLlama::logits + host argmax per token on this repository's gten/ headers "
                    "(libdropin.so); ids generated up to n = %d; tok/s = new ids / (wall - prompt processing); first_ids_agree = how many of "
                    "the first 8 greedy ids the fused rows and the operator-by-operator run share (at n ~ 1800 they differ by f32 "
                    "summation order of the chunked softmax, so near-ties may flip; tests hold both to the reference band)" % N_CTX}


def cpu_baseline(host, cfg, mode, seed, n_steps):
    """Time the CPU path on this box's host cores on a bounded sample of the same
    workload: single-token decode steps ending at n = 2048.  The K/V history below
    the sampled rows is whatever the freshly allocated caches hold (zero pages):
    the CPU path has no data-dependent control flow, so its timing is that of a
    real context of the same length, without spending minutes on a CPU prefill."""
    from oracle import orc
    import numpy as np
    wd, ad = MODES[mode]
    toks = host.synthetic_tokens(N_CTX, seed=12345)
    ref = orc.load_ref("avx")
    path = f"/tmp/gten_bench_{mode}_{seed}.gten"
    t_build = time.time()
    if not os.path.exists(path) or os.path.getsize(path) < 1000:
        host.write_gten(cfg, seed, path + ".tmp")
        os.replace(path + ".tmp", path)
    if ref is not None:
        kind, model = "reference", ref.tinyllama(N_CTX, wd, ad)
        model.load(path)
    else:
        kind = "port"
        model = orc.load_oracle().model(orc.Config(**{k: getattr(cfg, k) for k, _ in cfg._fields_}))
        model.load_gten(path)
    t_build = time.time() - t_build
    # The reference parallelises one loop (output features of a matmul,
    # gten/ops.h:635-637) with whatever OpenMP gives it.  On a many-core host the
    # default (all hardware threads) is far from its best, so sample a few team
    # sizes and report the fastest, with the thread count actually used.
    import ctypes
    gomp = ctypes.CDLL("libgomp.so.1")
    hw = os.cpu_count() or 1
    tried = {}
    for threads in sorted(t for t in {8, 16, 32, 64, min(hw, 64)} if t <= hw):
        # team sizes up to 64 only (all 256 hardware threads of the GPU box: 19.5 s per step, round 1), and stop
        # once a larger team has become clearly slower than the best one seen
        if tried and threads > 16 and tried[max(tried)] > 1.5 * min(tried.values()):
            break
        gomp.omp_set_num_threads(threads)
        times = []
        for i in range(n_steps):
            n = N_CTX - n_steps + 1 + i
            t0 = time.perf_counter()
            model.logits(toks[:n], n - 1)
            times.append(time.perf_counter() - t0)
        tried[threads] = float(np.median(times[1:] if len(times) > 1 else times))
    model.close()
    cores, med = min(tried.items(), key=lambda kv: kv[1])
    return {"value": round(1.0 / med, 3), "unit": "tok/s", "cores": cores, "kind": kind,
            "ms_per_step_by_threads": {str(k): round(v * 1e3, 1) for k, v in tried.items()},
            "sample": f"{n_steps} single-token decode steps at n={N_CTX - n_steps + 1}..{N_CTX} (median of all but "
                      f"the first), {mode} weights, zero-filled K/V history below the sampled rows, "
                      f"{'reference -O3 -fopenmp -mavx -mf16c build (oracle/_ref)' if kind == 'reference' else 'oracle C port'}",
            "ms_per_step": round(med * 1e3, 2), "setup_s": round(t_build, 1)}


class StubDecoder:
    """NOT a decoder: no GPU, no arithmetic.  A stand-in with the decode_step interface whose step sleeps, so that the
    CPU tests can drive this file's launcher, the torch.distributed rendezvous (gloo) and the JSON assembly without a
    GPU (--engine stub; tests/test_bench_launcher_cpu.py).  The line it produces is labelled as such."""
    step_s = 0.002

    def decode_begin(self, toks):
        self.toks = toks

    def decode_step(self, n, use_graph=True):
        time.sleep(self.step_s * (1.0 + 0.25 * int(os.environ.get("RANK", "0"))))     # higher ranks are the slow replicas
        if os.environ.get("GTEN_BENCH_STUB_FAIL_RANK") == os.environ.get("RANK", "0"):
            raise SystemExit(3)

    def decode_result(self, n):
        return int(self.toks[n - 1])

    def close(self):
        pass


class StubBatch:
    """NOT a decoder (see StubDecoder): `serve` sleeps for the queue it is handed and returns prompt + filler ids, so the
    CPU tests can drive the sharding of one global prompt queue over the replicas (--engine stub)."""
    per_token_s = 2e-6

    def serve(self, prompts, max_tokens, eos=-1, slice_steps=16, max_new=0, max_new_each=None):
        import numpy as np
        each = [max_new] * len(prompts) if max_new_each is None else [int(x) for x in max_new_each]
        new = [min(e, max_tokens - len(p)) for p, e in zip(prompts, each)]
        time.sleep(self.per_token_s * (sum(len(p) for p in prompts) + 8 * sum(new)))
        got = [np.concatenate([np.asarray(p, np.int32), np.full(k, 7, np.int32)]) for p, k in zip(prompts, new)]
        if os.environ.get("GTEN_BENCH_STUB_DROP_RANK") == os.environ.get("RANK", "0") and len(got) > 1:
            got[1] = got[1][:len(prompts[1])]                 # tests: a replica that loses one request's new ids
        return got, {"prompt_tokens": float(sum(len(p) for p in prompts)), "new_tokens": float(sum(new)), "steps": float(max(new, default=0)),
                     "admissions": float(len(prompts)), "prefill_s": 0.0, "decode_s": 0.0}

    # (the batched-decode leg of N > 1: a step of S sequences sleeps like a wide step, higher ranks slower)
    step_s = 0.004

    def load_synthetic(self, seed):
        pass

    def prefill(self, seq, tokens, want=True):
        return None

    def decode_begin(self, seq, tokens):
        pass

    def decode_steps(self, n_first, count, use_graph=True):
        time.sleep(self.step_s * count * (1.0 + 0.25 * int(os.environ.get("RANK", "0"))))

    def close(self):
        pass


def batched_decode(batch, S, K, make_tokens, rep, rank, world, dist, device, sync):
    """N > 1, the batched counterpart of `value`: EVERY replica steps its own decoder of S sequences (lanes of 128 rows) K times,
    the steps ending at n = N_CTX -- the same barrier-bracketed region as the batch-1 steps (MAX elapsed over ranks, tokens
    summed), and rank 0's rate over the same steps with the other GPUs idle at a barrier as the denominator of the efficiency.
    One GPU does ~40 x the tokens per second here that it does at batch 1, which is why this, not `value`, is the number a
    deployment of the 8 GPUs would quote; `value` stays BASELINE.json's metric."""
    n0 = N_CTX - K + 1
    for q in range(S):
        toks = make_tokens(N_CTX, 1000 + q + S * rank)
        if n0 > 1:
            batch.prefill(q, toks[:n0 - 1], want=False)          # every sequence's context by ONE prompt call (batched row matrix)
        batch.decode_begin(q, toks)
    batch.decode_steps(n0, min(4, K), True)                      # untimed: captures the four-step graph
    sync()

    def run():
        batch.decode_steps(n0, K, True)
        return S * K

    solo = None
    if dist is not None:
        dist.barrier()
        if rank == 0:
            dt0, k0 = rep.timed_region(run, sync)
            solo = k0 / dt0
        dist.barrier()
    elapsed, total, per_rank = rep.timed_region(run, sync, dist=dist, device=device, per_rank=True)
    out = {"streams_per_gpu": S, "steps": K, "tok_s": round(total / elapsed, 1), "ms_per_step": round(elapsed / K * 1e3, 4),
           "per_rank": [{"rank": r, "tok_s": round(k / e, 1), "ms_per_step": round(e / K * 1e3, 4)} for r, (e, k) in enumerate(per_rank)],
           "note": "every replica steps its own %d sequences (one decoder, lanes of 128 rows) %d times, steps ending at n = %d; "
                   "tok_s = tokens of all ranks / the slowest rank's time" % (S, K, N_CTX)}
    if solo is not None:
        out["solo_rank0_tok_s"] = round(solo, 1)
        out["efficiency"] = round(total / elapsed / (world * solo), 4)
    return out


def serving_queue(n_prompts, make_tokens, seed=2024):
    """ONE global synthetic queue, the same on every rank: prompt j has 64..512 ids (ids from make_tokens(length, j)) and
    a budget of 32..224 new ids, so that requests end at different times"""
    import numpy as np
    rng = np.random.default_rng(seed)
    lens = rng.integers(64, 513, n_prompts)
    budgets = rng.integers(32, 225, n_prompts).astype(np.int32)
    return [list(make_tokens(int(n), j)) for j, n in enumerate(lens)], budgets


def sharded_serving(batch, slots, n_global, make_tokens, rep, rank, world, dist, device, sync, slice_steps=8):
    """north_star's multi-GPU split: per-prompt sharding, no collective on the data path.  Rank r serves prompts
    r, r + world, ... (replicas.shard_prompts) of ONE global queue through its own slots; the only cross-rank traffic is
    the bookkeeping -- the barrier-bracketed timed region (MAX elapsed over ranks, new ids summed) and one all-reduce of
    a served-counter per prompt, which must come back as exactly 1 everywhere."""
    import numpy as np
    import torch
    prompts, budgets = serving_queue(n_global, make_tokens)
    mine = rep.shard_prompts(n_global, rank, world)
    my_prompts, my_budgets = [prompts[j] for j in mine], budgets[mine]
    batch.serve(my_prompts[:slots], N_CTX, -1, slice_steps, max_new=4)        # warm-up: graphs, first-use allocations
    box = {}

    def run():
        box["got"], box["st"] = batch.serve(my_prompts, N_CTX, -1, slice_steps, max_new_each=my_budgets)
        return int(box["st"]["new_tokens"])

    # rank 0's rate on its own shard while the other GPUs idle at a barrier: the denominator of the serving efficiency
    solo = None
    if dist is not None:
        dist.barrier()
        if rank == 0:
            dt0, k0 = rep.timed_region(run, sync)
            solo = k0 / dt0
        dist.barrier()
    elapsed, new_total, per_rank = rep.timed_region(run, sync, dist=dist, device=device, per_rank=True)
    served = torch.zeros(n_global, dtype=torch.int32, device=device)
    for j, ids in zip(mine, box["got"]):
        if len(ids) == len(prompts[j]) + min(int(budgets[j]), N_CTX - len(prompts[j])) and list(ids[:len(prompts[j])]) == list(prompts[j]):
            served[j] += 1
    ptoks = torch.tensor([box["st"]["prompt_tokens"], box["st"]["steps"]], dtype=torch.float64, device=device)
    every = [ptoks]
    if dist is not None:
        dist.all_reduce(served)
        every = [torch.zeros_like(ptoks) for _ in range(world)]
        dist.all_gather(every, ptoks)
    served = served.cpu().numpy()
    prompt_tokens = int(sum(float(t[0].item()) for t in every))
    steps = [int(t[1].item()) for t in every]
    return {"slots_per_gpu": slots, "prompts": n_global, "prompts_per_rank": [len(rep.shard_prompts(n_global, r, world)) for r in range(world)],
            "prompt_tokens": prompt_tokens, "new_tokens": int(new_total), "wall_s": round(elapsed, 3),
            "new_tok_s": round(new_total / elapsed, 1), "all_tok_s": round((new_total + prompt_tokens) / elapsed, 1),
            "per_rank": [{"rank": r, "new_tokens": k, "wall_s": round(e, 3), "new_tok_s": round(k / e, 1), "shared_steps": steps[r],
                          "slot_utilisation": round(k / max(steps[r] * slots, 1), 3)} for r, (e, k) in enumerate(per_rank)],
            "every_prompt_served_exactly_once": bool((served == 1).all()),
            "solo_rank0_new_tok_s": round(solo, 1) if solo else None,
            "efficiency": round(new_total / elapsed / (world * solo), 4) if solo else None,
            "note": "ONE global queue (prompts of 64..512 ids, 32..224 new ids each; %d per GPU: weak scaling) sharded round-robin "
                    "over the replicas, each rank its shard through its own %d slots (continuous batching, prompts on the library's "
                    "second stream beside slices of %d shared steps); new_tok_s = new ids of all ranks / the slowest rank's wall "
                    "time; no collective on the data path (the served-counter all-reduce and the timing all-gather are bookkeeping)"
                    % (n_global // world, slots, slice_steps)}


class Leg:
    """a secondary leg of the line: an exception inside it is recorded under the leg's name instead of costing the line"""

    def __init__(self, out, name):
        self.out, self.name = out, name

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is not None and issubclass(et, Exception):
            self.out[self.name] = {"error": "%s: %s" % (et.__name__, ev)}
            return True
        return False


def remove_synth_cache():
    """the per-node synthetic-weight cache in GTEN_SYNTH_CACHE_DIR (host/tinyllama_model.h load_synthetic_cached): removed by the
    node's first local rank once its replicas have loaded -- and again on the way out of main(), so that a run that failed
    half way leaves nothing in /dev/shm (which is memory)"""
    d = os.environ.get("GTEN_SYNTH_CACHE_DIR")
    if not d:
        return
    import glob
    for f in glob.glob(os.path.join(d, "gten_synth_*")):
        try:
            os.remove(f)
        except OSError:
            pass


def load_replicas():
    """tinyllama.cpp_amd/replicas.py by path: the launcher parent must not import anything that could touch the GPU"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gten_replicas", os.path.join(ROOT, "tinyllama.cpp_amd", "replicas.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return json.loads(line)
    return None


def launcher(args, argv):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N workers, relay rank 0's line.
    This process never initialises the GPU (counting devices reads sysfs) and never exec()s."""
    rep = load_replicas()
    n = args.gpus
    cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--curve"]
    curve = []
    if args.curve:
        m = 1
        while m < n:
            sub = [a for a in cmd]
            gi = sub.index("--gpus")
            sub[gi + 1] = str(m)
            rc, out0 = rep.launch(m, sub + ["--brief"], timeout=args.launch_timeout)
            line = last_json_line(out0)
            if rc != 0 or line is None:
                print(f"bench.py: the {m}-replica run of --curve failed (exit {rc})", file=sys.stderr)
                return rc or 1
            curve.append({"n_gpus": m, "value": line["value"], "ms_per_step": line["ms_per_step"]})
            m *= 2
    t0 = time.time()
    rc, out0 = rep.launch(n, cmd, timeout=args.launch_timeout)
    line = last_json_line(out0)
    if rc != 0 or line is None:
        sys.stdout.write(out0)
        print(f"bench.py: a replica failed (exit {rc})" if rc else "bench.py: rank 0 printed no result line", file=sys.stderr)
        return rc or 1
    line["launcher"] = {"kind": "bench.py (subprocess per GPU; tinyllama.cpp_amd/replicas.py launch())", "wall_s": round(time.time() - t0, 1)}
    if curve:
        one = curve[0]["value"]
        line["scaling_curve"] = curve + [{"n_gpus": n, "value": line["value"], "ms_per_step": line["ms_per_step"]}]
        for c in line["scaling_curve"]:
            c["efficiency_vs_1gpu_run"] = round(c["value"] / (c["n_gpus"] * one), 4)
    print(json.dumps(line), flush=True)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if "--gpus" not in argv:                       # --gpus=N form: normalise for the workers' command line
            argv = [a for a in argv if not a.startswith("--gpus=")] + ["--gpus", str(args.gpus)]
        return launcher(args, argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is used", file=sys.stderr)
    if args.rehearse_one_gpu:
        local_rank = 0
        os.environ["GTEN_HIP_DEVICE"] = "0"
    os.environ.setdefault("GTEN_HIP_DEVICE", str(local_rank))
    # weight synthesis is OpenMP code on the host: share the cores between the ranks of this node
    ncpu = usable_cpus()
    os.environ.setdefault("OMP_NUM_THREADS", str(max(2, ncpu // max(world, 1))))
    os.environ.setdefault("GTEN_SYNTH_GEN_THREADS", str(ncpu))
    # ... and generate the synthetic weights ONCE per node: the first replica to get there writes them to /dev/shm, the others
    # read them (host/tinyllama_model.h load_synthetic_cached); rank 0 removes the files when it is done
    if world > 1 and os.path.isdir("/dev/shm"):
        os.environ.setdefault("GTEN_SYNTH_CACHE_DIR", "/dev/shm")
    stub = args.engine == "stub"

    import torch
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:      # one rank per GPU (torch.distributed.run or launcher() above)
        import torch.distributed as dist
        if stub or args.rehearse_one_gpu:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    try:
        return worker(args, rank, local_rank, world, dist)
    finally:
        if dist is not None and dist.is_initialized():
            dist.destroy_process_group()
        if local_rank == 0:
            remove_synth_cache()            # (a run that failed before the replicas were done with the shared weights: nothing stays behind)


def worker(args, rank, local_rank, world, dist):
    import numpy as np
    import torch
    stub = args.engine == "stub"
    rep = load_replicas()
    rep_seed = rep.prompt_seed
    K, W = args.steps, args.warmup
    global N_CTX
    N_CTX = args.ctx
    use_graph = not args.no_graph
    wd, ad = MODES[args.mode]
    if stub:
        hip = host = cfg = None
        model = StubDecoder()
        load_s = 0.0
        rng = np.random.default_rng(rep_seed(12345, rank))
        toks = rng.integers(3, 31993, N_CTX).astype(np.int32)
        fused = True
    else:
        from __graft_entry__ import load_package
        pkg = load_package()
        hip = pkg.hipabi.load(local_rank)
        if args.kv_rows:
            hip.set_kv_head_major(False)
        if args.no_wx_planes:
            hip.set_wx_planes(False)
        host = pkg.load_host()
        cfg = host.default_config(wd, ad)
        model = host.model(cfg)
        t0 = time.time()
        model.load_synthetic(args.seed)
        load_s = time.time() - t0
        toks = host.synthetic_tokens(N_CTX, seed=rep_seed(12345, rank))
        fused = hasattr(model, "decode_step") and args.path in ("auto", "fused")
        if args.path == "fused" and not fused:
            raise SystemExit("fused decode path requested but not built")

    window = 64                                  # timed steps cycle over n in (N_CTX-window, N_CTX]
    def n_of(i, total):
        if total <= N_CTX - 1:
            return N_CTX - total + 1 + i         # the last step lands on n = N_CTX
        return N_CTX - window + 1 + (i % window)

    if fused:
        model.decode_begin(toks)                 # teacher-forced ids live on the device
    else:
        model.set_fast_decode(False)

    def step(n):
        if fused:
            model.decode_step(n, use_graph)      # asynchronous graph replay; argmax stays on the device
            return None
        lg = model.logits(toks[:n], n - 1)       # logits reach the host; greedy argmax there
        return int(np.argmax(lg))

    # fill the context with real decode steps (untimed), then warm up
    total = K + W
    first = n_of(0, total)
    t_fill = time.time()
    if stub:
        pass                                     # nothing to fill
    elif args.fill == "prefill" and first > 1:
        model.logits(toks[:first - 1], 0, want=False)
    else:
        for n in range(1, first):
            if fused:
                model.decode_step(n, use_graph)
            else:
                model.logits(toks[:n], n - 1, want=False)

    def sync():
        if not stub:
            hip.sync()
            torch.cuda.synchronize()

    sync()
    t_fill = time.time() - t_fill
    # the K timed steps are consecutive positions (unless the window wraps): one asynchronous call, replayed four steps
    # per hipGraph launch (gten_hip_decoder_steps) -- the same kernels and results as K single-step calls
    consecutive = fused and not stub and use_graph and total <= N_CTX - 1 and hasattr(model, "decode_steps")
    if consecutive:
        model.decode_steps(n_of(0, total), 4, True)      # untimed: captures the four-step graph (teacher-forced ids: repeatable)
        sync()
    for i in range(W):
        step(n_of(i, total))

    def run_steps():
        if consecutive:
            model.decode_steps(n_of(W, total), K, True)
            return K
        for i in range(W, W + K):
            step(n_of(i, total))
        return K

    ddev = None if (dist is None or stub or args.rehearse_one_gpu) else "cuda"
    # N > 1: rank 0's rate with the other GPUs idle (they wait at the barrier), the denominator of `efficiency`
    solo = None
    if dist is not None:
        dist.barrier()
        if rank == 0:
            solo_dt, solo_tokens = rep.timed_region(run_steps, sync)
            solo = solo_tokens / solo_dt
        dist.barrier()
    # barrier + synchronise on both sides of exactly K steps; MAX elapsed over ranks
    elapsed, total_tokens, per_rank = rep.timed_region(run_steps, sync, dist=dist, device=ddev, per_rank=True)
    last = model.decode_result(n_of(W + K - 1, total)) if fused else step(n_of(W + K - 1, total))
    # N > 1: one global prompt queue sharded over the replicas (north_star's per-prompt split), every rank its shard
    # through its own continuous-batching slots -- the aggregate serving rate beside the batch-1 rate of `value`
    sharded = None
    if dist is not None and world > 1 and fused and args.serve > 0 and (args.serve_slots or args.wide_streams) > 1 and not args.brief:
        S = args.serve_slots or args.wide_streams
        # A failure on ONE rank (out of memory creating the slots, a serve error) must not take the already measured `value`
        # with it nor leave the other ranks in a collective: every rank reports whether it got this far, the leg runs only
        # if all did, and whatever it raises is recorded instead of propagated.
        sbatch, err = None, None
        try:
            if stub:
                sbatch = StubBatch()
                make_tokens = lambda n, j: np.random.default_rng(rep_seed(999, j)).integers(3, 31993, n).astype(np.int32)
            else:
                sbatch = host.batch(cfg, S)
                sbatch.load_synthetic(args.seed)
                make_tokens = lambda n, j: host.synthetic_tokens(n, seed=rep_seed(999, j))
        except Exception as e:                       # noqa: BLE001 -- recorded below
            err = "%s: %s" % (type(e).__name__, e)
        okf = torch.tensor([0.0 if err else 1.0], device=ddev) if ddev else torch.tensor([0.0 if err else 1.0])
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if float(okf.item()) < 0.5:
            sharded = {"error": err or "another rank could not set the leg up"}
        else:
            try:
                sharded = sharded_serving(sbatch, S, args.serve * world, make_tokens, rep, rank, world, dist, ddev, sync, args.serve_slice)
            except Exception as e:                   # noqa: BLE001
                sharded = {"error": "%s: %s" % (type(e).__name__, e)}
        if sbatch is not None:
            try:
                sbatch.close()
            except Exception:                        # noqa: BLE001
                pass
    # N > 1: the batched decode step on every replica (one decoder of 256 sequences per GPU), same bracketing as `value`
    batched = None
    if dist is not None and world > 1 and fused and (args.serve_slots or args.wide_streams) > 1 and not args.brief:
        S = 256 if not stub else 8
        bbatch, err = None, None
        try:
            if stub:
                bbatch = StubBatch()
                make_tokens_b = lambda n, j: np.random.default_rng(rep_seed(555, j)).integers(3, 31993, n).astype(np.int32)
            else:
                bbatch = host.batch(cfg, S)
                bbatch.load_synthetic(args.seed)
                make_tokens_b = lambda n, j: host.synthetic_tokens(n, seed=rep_seed(12345, j))
        except Exception as e:                       # noqa: BLE001 -- recorded below
            err = "%s: %s" % (type(e).__name__, e)
        okf = torch.tensor([0.0 if err else 1.0], device=ddev) if ddev else torch.tensor([0.0 if err else 1.0])
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        if float(okf.item()) < 0.5:
            batched = {"error": err or "another rank could not set the leg up"}
        else:
            try:
                batched = batched_decode(bbatch, S, min(K, 32), make_tokens_b, rep, rank, world, dist, ddev, sync)
            except Exception as e:                   # noqa: BLE001
                batched = {"error": "%s: %s" % (type(e).__name__, e)}
        if bbatch is not None:
            try:
                bbatch.close()
            except Exception:                        # noqa: BLE001
                pass
    # the replicas are done with each other: every rank leaves the process group here (rank 0 goes on alone with the
    # roofline / CPU-baseline legs, the others exit and free their host cores)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        if local_rank == 0:                                               # every replica of this node has read the shared weights by now
            remove_synth_cache()
        os.environ.pop("GTEN_SYNTH_CACHE_DIR", None)                      # rank 0's later models generate their own
    if rank != 0:
        model.close()
        return 0

    # ---- roofline of the dominant kernel: HIP events on the library's stream
    roofline = None
    if not stub:
        P = min(16, K)
        hip.prof_enable(True)
        for i in range(P):
            step(n_of(W + K - P + i, total))
        prof = hip.prof_read()
        hip.prof_enable(False)
        wb = weight_bytes_per_elt(args.mode)
        act_b = 2.0 if args.mode == "f16" else 34 / 32
        lin = {name: (dout * din * wb + din * act_b + dout * act_b, cnt) for name, dout, din, cnt in linear_shapes()}
        n_att = N_CTX - K // 2 if total <= N_CTX - 1 else N_CTX - window // 2          # context length in the middle of the timed window
        per_family_bytes = {
            "matmul_2d": sum(b * c for b, c in lin.values()),
            "decode_gemv_qkv": (lin["q"][0] + lin["k"][0] + lin["v"][0]) * 22,
            "decode_gemv_o": lin["o"][0] * 22,
            "decode_gemv_gateup": (lin["gate"][0] + lin["up"][0]) * 22,
            "decode_gemv_down": lin["down"][0] * 22,
            "decode_gemv_head": lin["lm_head"][0],
            # the one-launch attention reads the K and the V history of its block once: KV(n) of SURVEY 8(d) / 22 blocks
            "decode_attn_score": 2 * 256 * act_b * n_att * 22,
        }
        # the same isolated timing for every family of the step: two HIP events on the library's stream around 20
        # replays of a graph holding ONLY that family's launches (per-launch event brackets, above, carry ~2.5 us of
        # packet handling each and only rank the kernels)
        family_us, family_n = {}, {}
        if fused and use_graph:              # (--no-graph runs are counter collections: no graph is captured at all)
            for f2 in prof:
                try:
                    u2, n2 = model.time_family(hip.prof_family_index(f2), N_CTX, 20)
                    family_us[f2], family_n[f2] = round(u2, 3), n2
                except Exception:
                    pass
        # dominant kernel = the family with the largest share of the step (isolated time x launches) among those with
        # algorithmic bytes
        share = {f: family_us[f] * family_n[f] for f in family_us if f in per_family_bytes} or \
                {f: v[1] for f, v in prof.items() if f in per_family_bytes}
        fam = max(share, key=share.get)
        launches, ms = prof[fam]
        per_launch = per_family_bytes[fam] / (launches / P)
        bracket_us = ms * 1e3 / launches
        avg_us, per_replay = (family_us[fam], family_n[fam]) if fam in family_us else (bracket_us, launches // P)
        achieved = (per_launch / (avg_us * 1e-6) / 1e9) if per_launch else None
        step_ms = elapsed / K * 1e3
        roofline = {"bound": "hbm", "kernel": fam, "achieved": round(achieved, 1) if achieved else None,
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4) if achieved else None,
                    "traffic": None, "avg_launch_us": round(avg_us, 3), "launches_per_step": per_replay,
                    "algorithmic_bytes_per_launch": int(per_launch) if per_launch else None,
                    "timing": "HIP events around 20 graph replays of this kernel family alone" if fused else "HIP event pair per launch",
                    "family_avg_launch_us": family_us,
                    "per_launch_event_bracket_us": {k: round(v[1] * 1e3 / v[0], 2) for k, v in prof.items()},
                    "bracket_share_of_step": {k: round(v[1] / P / step_ms, 3) for k, v in prof.items()}}
        roofline["traffic"], roofline["traffic_source"] = measured_traffic(args.mode, fam)
        roofline["csrc_sha256_16"] = csrc_fingerprint()

    tok_s = total_tokens / elapsed
    ms_step = elapsed / K * 1e3
    n_mid = N_CTX - K // 2 if total <= N_CTX - 1 else N_CTX - window // 2
    whole = algorithmic_bytes(args.mode, n_mid) / (ms_step * 1e-3) / 1e9
    out = {
        "metric": f"decode tok/s TinyLlama-1.1B {args.mode} ctx=2048, {world} GPU{'s' if world > 1 else ''}; achieved HBM GB/s vs peak",
        "value": round(tok_s, 2), "unit": "tok/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"q4": "q4 weights x q8 activations (int8 dot, f32 accumulate)", "q8": "q8 x q8 (int8 dot, f32 accumulate)",
                  "f16": "f16 (f32 accumulate)"}[args.mode],
        "data": "synthetic",
        "config": {"workload": f"TinyLlama-1.1B {args.mode} greedy single-token decode, batch 1 per GPU, steps end at n=2048 "
                               f"(BASELINE.json configs[{ {'f16': 1, 'q8': 2, 'q4': 3}[args.mode] }])",
                   "ctx": N_CTX, "graph": use_graph, "steps_per_graph_replay": 4 if consecutive else 1, "kv_cache": "f16" if args.mode == "f16" else "q8 blocks (reference layout)",
                   "path": "fused" if fused else "ops", "argmax": "device" if fused else "host (128 KB logits D2H per step)",
                   "parallelism": f"replicas x{world}", "last_token": int(last)},
        "whole_step_hbm": {"achieved": round(whole, 1), "unit": "GB/s", "frac": round(whole / HBM_PEAK_GBPS, 4),
                           "algorithmic_bytes_per_step": int(algorithmic_bytes(args.mode, n_mid))},
        "roofline": roofline,
        "setup": {"weights_s": round(load_s, 1), "context_fill_s": round(t_fill, 1), "context_fill": args.fill},
    }
    if world > 1:
        out["per_rank"] = [{"rank": r, "tok_s": round(k / e, 2), "ms_per_step": round(e / max(k, 1) * 1e3, 4)}
                           for r, (e, k) in enumerate(per_rank)]
        out["solo_rank0_tok_s"] = round(solo, 2)
        out["efficiency"] = round(tok_s / (world * solo), 4)
        out["efficiency_note"] = ("value / (n_gpus x rank 0's rate over the same K steps while the other GPUs idle at a "
                                  "barrier, same run); value = tokens of all ranks / the slowest rank's time")
    if sharded is not None:
        out["sharded_serving"] = sharded
    if batched is not None:
        out["batched_decode"] = batched
    if world > 1:
        # what N GPUs are FOR in this design, at the top level beside `value` (which stays BASELINE.json's batch-1 metric): the
        # aggregate of the batched step and of the sharded serving queue, each with its own efficiency against rank 0 alone
        out["batched_tok_s"] = (batched or {}).get("tok_s")
        out["batched_efficiency"] = (batched or {}).get("efficiency")
        out["serving_new_tok_s"] = (sharded or {}).get("new_tok_s")
        out["serving_efficiency"] = (sharded or {}).get("efficiency")
    if args.rehearse_one_gpu and world > 1:
        out["metric"] = "REHEARSAL (all %d replicas share GPU 0): " % world + out["metric"]
    if stub:
        out["whole_step_hbm"] = None
        out["metric"] = "STUB (no GPU work): " + out["metric"]
        out["data"] = "none: --engine stub sleeps instead of decoding (launcher / rendezvous test only)"
        out["config"]["path"] = "stub"
    secondary = world == 1 and not args.brief and not stub
    # secondary (median and mean per step): the same K steps again with a synchronise after each one,
    # i.e. the latency a caller that waits for every token sees (`value` above queues the K steps back to back)
    if secondary and fused:
      with Leg(out, "per_step_synced"):
            per = []
            for i in range(K):
                n = n_of(W + i, total)
                hip.sync()
                t0 = time.perf_counter()
                model.decode_step(n, use_graph)
                hip.sync()
                per.append(time.perf_counter() - t0)
            med = float(np.median(per))
            out["per_step_synced"] = {"steps": K, "median_ms": round(med * 1e3, 4), "mean_ms": round(float(np.mean(per)) * 1e3, 4),
                                      "tok_s_from_median": round(1.0 / med, 1)}
    # secondary: the same K steps as ONE persistent launch per step (csrc/gten_decode_persist.h: granule hand-offs instead of 113
    # launch boundaries; the same bytes, tests/test_persist_gpu.py) -- measured slower than the launch chain on MI355X, reported
    # beside `value`, never in it
    persist_built = False
    if secondary and fused and args.mode == "q4" and use_graph:
        try:
            hip.set_decode_persistent(True)          # (refused by a library built without -DGTEN_WITH_PERSIST=1: the product build)
            hip.set_decode_persistent(False)
            persist_built = True
        except Exception:
            pass
    if persist_built:
      with Leg(out, "persistent_step"):
            hip.set_decode_persistent(True)
            try:
                pm = host.model(cfg)
                pm.load_synthetic(args.seed)
                pm.decode_begin(toks)
            finally:
                hip.set_decode_persistent(False)
            n0 = N_CTX - K + 1
            pm.decode_steps(1, n0 - 1, True)                     # fill the context through the persistent step itself
            pm.decode_steps(n0, K, True); hip.sync()             # warm: the four-step graph
            t0 = time.perf_counter()
            pm.decode_steps(n0, K, True)
            hip.sync()
            dtp = time.perf_counter() - t0
            nd, launches, ab, _ = hip.persist_status()
            same = pm.decode_result(N_CTX) == model.decode_result(N_CTX) if total <= N_CTX - 1 else None
            out["persistent_step"] = {"ms_per_step": round(dtp / K * 1e3, 4), "tok_s": round(K / dtp, 1), "decoders": nd, "launches": int(launches),
                                      "abort_code": ab, "same_last_id_as_chain": same,
                                      "vs_launch_chain": round((elapsed / K) / (dtp / K), 3),
                                      "note": "gten_hip_set_decode_persistent(1): one workgroup per CU, 8-byte {value, tag} granules between "
                                              "the phases of a block, weights requested ahead of the dependency chain; per-phase profile: "
                                              "profiles/r04_persistent_phases.txt, edge costs: profiles/r04_microbench_edge.txt"}
            pm.close()
    # secondary: the short-context window n in [16, 80) of the same single-sequence decode path
    if secondary and fused:
      with Leg(out, "short_ctx"):
            for n in range(1, 80):
                model.decode_step(n, use_graph)
            hip.sync()
            t0 = time.perf_counter()
            for n in range(16, 80):
                model.decode_step(n, use_graph)
            hip.sync()
            dts = time.perf_counter() - t0
            out["short_ctx"] = {"window": "n in [16, 80)", "steps": 64, "ms_per_step": round(dts / 64 * 1e3, 4), "tok_s": round(64 / dts, 1)}
    # secondary: several sequences on this GPU sharing every weight pass (not part of `value`)
    def multi_stream(S, note):
        batch = host.batch(cfg, S)
        batch.load_synthetic(args.seed)
        seqs = [host.synthetic_tokens(N_CTX, seed=rep_seed(12345, 1000 + q)) for q in range(S)]
        if args.fill == "prefill" and first > 1:
            for q in range(S):
                batch.prefill(q, seqs[q][:first - 1], want=False)      # each sequence's prompt on its own caches
        for q in range(S):
            batch.decode_begin(q, seqs[q])
        if not (args.fill == "prefill" and first > 1):
            for n in range(1, first):
                batch.decode_step(n, use_graph)
        if use_graph and total <= N_CTX - 1:
            batch.decode_steps(n_of(0, total), 4, True)  # untimed: captures the four-step graph
        for i in range(W):
            batch.decode_step(n_of(i, total), use_graph)
        hip.sync()
        t0 = time.perf_counter()
        if use_graph and total <= N_CTX - 1:
            batch.decode_steps(n_of(W, total), K, True)
        else:
            for i in range(W, W + K):
                batch.decode_step(n_of(i, total), use_graph)
        hip.sync()
        dt = time.perf_counter() - t0
        ms = dt / K * 1e3
        bytes_step = algorithmic_bytes(args.mode, n_mid) + (S - 1) * (algorithmic_bytes(args.mode, n_mid) - algorithmic_bytes(args.mode, 0))
        fams = {}
        for fam in ("decode_stage", "decode_gemv_qkv", "decode_attn_score", "decode_attn_pv", "decode_gemv_o", "decode_gemv_gateup",
                    "decode_gemv_down", "decode_gemv_head"):
            try:
                us, per = batch.time_family(hip.prof_family_index(fam), N_CTX, 20)
                fams[fam] = {"us": round(us, 2), "launches": per}
            except Exception:
                pass                                 # a family this path does not launch
        ach = bytes_step / (ms * 1e-3) / 1e9
        cnt, why = measured_counters("lanes%d" % S) if args.mode == "q4" else (None, "counters are collected for q4")
        res = {"streams": S, "tok_s": round(S * K / dt, 1), "ms_per_step": round(ms, 4),
               "speedup_vs_single": round(S * K / dt / tok_s, 2),
               "hbm": {"achieved": round(ach, 1), "unit": "GB/s",
                       "algorithmic_bytes_per_step": int(bytes_step),
                       "note": "weights once per step + one K/V history per sequence"},
               # the WHOLE step against the HBM roof (its kernels are W.x launches and attention launches in about equal parts)
               "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": (cnt or {}).get("hbm_bytes_per_step"),
                            "traffic_unit": "HBM bytes per step (PMC)", "traffic_source": why,
                            "algorithmic_bytes_per_step": int(bytes_step)},
               "gateup_kernel_us": fams["decode_gemv_gateup"]["us"], "kernel_us": fams,
               "last_tokens": [batch.decode_result(q, n_of(W + K - 1, total)) for q in range(min(S, 8))],
               "note": note}
        batch.close()
        return res

    if secondary and fused and (args.streams > 1 or args.wide_streams > 1):
        model.close()
        if args.streams > 1:
          with Leg(out, "multi_stream"):
            out["multi_stream"] = multi_stream(args.streams, "GEMV kernels: per sequence bit-identical to the single-sequence decoder "
                                               "(tests/test_multiseq_gpu.py)" if args.streams <= 8 else "W.x on the matrix cores (k_dec_mmv)")
        if args.wide_streams > 1:
          with Leg(out, "multi_stream_wide"):
            out["multi_stream_wide"] = multi_stream(args.wide_streams,
                                                    ("W.x of the step on v_mfma_f32_16x16x32_f16 (k_dec_mmv_f16, rows = sequences), per-head attention; "
                                                     if args.mode == "f16" else
                                                     "W.x of the step as int8 MFMA GEMV (k_dec_mmv, rows = sequences), GQA-grouped attention; ") +
                                                    "per sequence inside the model band around single-sequence decode, "
                                                    "graph == eager bit for bit (tests/test_multiseq_gpu.py)")
        if args.wide_streams == 64 and use_graph and total <= N_CTX - 1 and not args.no_lanes:
            # round 3: 256 sequences in ONE decoder -- lanes of 128 (f16: 64), each lane's launch chain a parallel branch of the
            # step's graph (round 2 measured the effect with two separate decoders on two streams: multi_stream_wide_x2)
            try:
                out["multi_stream_lanes"] = multi_stream(256, "one decoder, two lanes of 128 sequences: the lanes' launch "
                                                              "chains are parallel branches of one graph and fill each other's gaps, a lane's W.x workgroups "
                                                              "run eight row tiles per expanded weight fragment; per sequence the bits of a 64-sequence "
                                                              "decoder (tests/test_multiseq_gpu.py); wide attention on head-major K / V shadows (round 5: "
                                                              "csrc/gten_decode_attn_hm.h); 128 / 384 sequences measured 60.4 k / 90.3 k tok/s")
            except Exception as e:
                out["multi_stream_lanes"] = {"tok_s": None, "note": "lanes leg unavailable: %r" % (e,)}
            # ... and 512 sequences (four lanes of 128): 12.5 GB of K / V per step (f16: 24 GB), the weights' share of the bytes below 5 %
            if True:
                try:
                    out["multi_stream_lanes512"] = multi_stream(512, "one decoder, four lanes of 128 sequences: the step is 95 % K / V bytes, read once per step as "
                                                                     "contiguous runs from head-major shadows (k_dec_attn_hm)")
                except Exception as e:
                    out["multi_stream_lanes512"] = {"tok_s": None, "note": "512-sequence leg unavailable: %r" % (e,)}
        model = host.model(cfg)
        model.load_synthetic(args.seed)
    # secondary: real greedy generation of a whole batch (sampler on the device, every sequence its own prompt)
    if secondary and fused and args.generate > 0 and args.wide_streams > 1:
      with Leg(out, "batch_generation"):
            S = args.wide_streams
            G = min(args.generate, N_CTX - 16)
            P0 = N_CTX - G
            batch = host.batch(cfg, S)
            batch.load_synthetic(args.seed)
            prompts = [host.synthetic_tokens(P0, seed=rep_seed(777, q)) for q in range(S)]
            batch.prefill(0, prompts[0]); hip.sync()      # warm-up (one-time attribute / scratch set-up)
            t_pre = 1e9
            for q in (1, 2):
                t0 = time.perf_counter(); batch.prefill(q % S, prompts[q % S]); hip.sync()
                t_pre = min(t_pre, time.perf_counter() - t0)
            t0 = time.perf_counter()
            ids = batch.generate(prompts, N_CTX)
            dt = time.perf_counter() - t0
            new = sum(len(x) - P0 for x in ids)
            out["batch_generation"] = {"streams": S, "prompt_tokens_each": P0, "new_tokens": int(new), "wall_s": round(dt, 3),
                                       "prefill_s_estimate": round(S * t_pre, 3),
                                       "tok_s": round(new / max(dt - S * t_pre, 1e-9), 1),
                                       "note": "greedy ids generated up to n = %d for every sequence; tok/s = new ids / (wall time - "
                                               "n_seq x one prompt's processing time)" % N_CTX}
            batch.close()
    # secondary: continuous batching -- a queue of synthetic prompts (64..512 ids each, 32..224 new ids per prompt)
    # through the serving slots sharing every weight pass; a slot that ends takes the next prompt while the others go on
    if secondary and fused and args.serve > 0 and (args.serve_slots or args.wide_streams) > 1:
      with Leg(out, "serving"):
            S = args.serve_slots or args.wide_streams
            hip.set_lane_skip(args.lane_skip)
            batch_spares = args.serve_spares
            batch = host.batch(cfg, S)
            batch.load_synthetic(args.seed)
            prompts, budgets = serving_queue(args.serve, lambda n, j: host.synthetic_tokens(n, seed=rep_seed(999, j)))
            batch.set_serve_spares(batch_spares)
            if args.serve_ramp >= 0:
                batch.set_serve_ramp(args.serve_ramp)
            batch.serve(prompts[:S], N_CTX, -1, args.serve_slice, max_new=4)           # warm-up: graphs, first-use allocations
            t0 = time.perf_counter()
            got, st = batch.serve(prompts, N_CTX, -1, args.serve_slice, max_new_each=budgets)
            dt = time.perf_counter() - t0
            out["serving"] = {"slots": S, "prompts": int(args.serve), "prompt_tokens": int(st["prompt_tokens"]), "new_tokens": int(st["new_tokens"]),
                              "wall_s": round(dt, 3), "new_tok_s": round(st["new_tokens"] / dt, 1),
                              "all_tok_s": round((st["new_tokens"] + st["prompt_tokens"]) / dt, 1),
                              "shared_steps": int(st["steps"]), "prefill_s": round(st["prefill_s"], 3), "decode_s": round(st["decode_s"], 3),
                              "slot_utilisation": round(st["new_tokens"] / max(st["steps"] * S, 1), 3),
                              # (round 4: a lane whose slots are all parked sits a run out -- the share of COMPUTED slot-steps that made an id)
                              "lane_steps": int(st.get("lane_steps", 0)), "lane_rows": int(st.get("lane_rows", 0)),
                              "computed_slot_utilisation": round(st["new_tokens"] / max(st.get("lane_steps", 0) * st.get("lane_rows", 0), 1), 3),
                              "slice_steps": args.serve_slice,
                              "note": "sustained rates over the whole queue (prompts of 64..512 ids, 32..224 new ids each); prompt processing runs on "
                                      "the library's second stream BESIDE the slices of shared steps (prefill_s = host time spent in it, "
                                      "overlapped); a slot whose run ends inside a slice repeats its last step until the slice is over; "
                                      "slot_utilisation = new ids / (shared steps x slots)"}
            # the queue against the two roofs it is made of: its shared steps' weight + K/V bytes against HBM, its prompt ids'
            # linear-layer FLOPs against the matrix peak -- both over the SAME wall time, so the two fractions add up to what the
            # chip was asked for (prompts and steps overlap on two streams)
            try:
                w_bytes = algorithmic_bytes(args.mode, 0)
                kv_per_pos = algorithmic_bytes(args.mode, 1) - w_bytes
                # mean context of a live slot ~ prompt (64..512) + half its budget (32..224): ~350 positions
                mean_ctx = (st["prompt_tokens"] + 0.5 * st["new_tokens"]) / max(args.serve, 1)
                step_bytes = st["steps"] * w_bytes + st["new_tokens"] * kv_per_pos * mean_ctx
                ach_gbps = step_bytes / dt / 1e9
                ach_tf = 2.0 * 1_034_426_368 * st["prompt_tokens"] / dt / 1e12
                out["serving"]["roofline"] = {
                    "bound": "hbm", "achieved": round(ach_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach_gbps / HBM_PEAK_GBPS, 4),
                    "traffic": None,
                    "what": "shared steps x weight bytes + new ids x K/V bytes at the queue's mean context (%.0f positions), over the whole wall time" % mean_ctx,
                    "prompt_side": {"bound": "mfma", "achieved": round(ach_tf, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": round(ach_tf / MFMA_F16_PEAK_TFLOPS, 4), "what": "prompt ids x linear-layer FLOPs over the same wall time"}}
            except Exception as e:                                       # a reporting extra: never costs the line
                out["serving"]["roofline"] = "not computed: %s" % type(e).__name__
            try:
                # the same queue once more, untimed: the ids must not depend on which prompt overlapped which slice
                got2, _ = batch.serve(prompts, N_CTX, -1, args.serve_slice, max_new_each=budgets)
                out["serving"]["same_ids_second_run"] = bool(len(got2) == len(got) and all(np.array_equal(a, b) for a, b in zip(got, got2)))
            except Exception as e:                                       # a reporting extra: never costs the line
                out["serving"]["same_ids_second_run"] = "not checked: %s" % type(e).__name__
            try:
                # the first 256 prompts alone (round 2's queue): two rounds of 128 slots, so the tail weighs more
                if args.serve > 256:
                    t0 = time.perf_counter()
                    _, st2 = batch.serve(prompts[:256], N_CTX, -1, args.serve_slice, max_new_each=budgets[:256])
                    dt2 = time.perf_counter() - t0
                    out["serving"]["first_256_prompts"] = {"new_tok_s": round(st2["new_tokens"] / dt2, 1), "wall_s": round(dt2, 3),
                                                           "shared_steps": int(st2["steps"]),
                                                           "slot_utilisation": round(st2["new_tokens"] / max(st2["steps"] * S, 1), 3)}
            except Exception as e:
                out["serving"]["first_256_prompts"] = "not measured: %s" % type(e).__name__
            batch.close()
    # secondary: prompt processing on the matrix cores (not part of `value`)
    if secondary and args.prefill > 0:
      with Leg(out, "prefill"):
            P = min(args.prefill, N_CTX)
            model.set_fast_decode(False)
            model.logits(toks[:P], 0, want=False)
            hip.sync()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                model.logits(toks[:P], 0, want=False)
            hip.sync()
            dt = (time.perf_counter() - t0) / reps
            flops = 2.0 * 1_034_426_368 * P
            # where the time goes: one more pass with an event pair around every launch (adds ~2 us per launch)
            hip.prof_enable(True)
            model.logits(toks[:P], 0, want=False)
            fams = {k: round(v[1], 3) for k, v in hip.prof_read().items()}
            hip.prof_enable(False)
            wx_ms = fams.get("matmul_2d_mfma", 0.0)
            out["prefill"] = {"prompt_tokens": P, "ms": round(dt * 1e3, 3), "tok_s": round(P / dt, 1),
                              "linear_tflops": round(flops / dt / 1e12, 2),
                              "family_ms": fams,
                              "wx_tflops": round((flops - 2.0 * 32003 * 2048 * (P - 1)) / (wx_ms * 1e-3) / 1e12, 1) if wx_ms > 0 else None,
                              "roofline": prefill_roofline(args, P, flops, wx_ms),
                              "note": "W.x on v_mfma_f32_16x16x32_f16 (gten_mfma.hip, fast form: block deltas folded into the f16 operands, K accumulated in the matrix core), "
                                      "attention on gten_attn_tiled.hip (int8 / f16 MFMA scores), element-wise ops on the "
                                      "block-pair kernel; linear_tflops = linear-layer FLOPs / whole prefill time; wx_tflops = the same FLOPs (lm_head: last row only) / "
                                      "the event-bracketed time of the W.x launches alone (activation expansion included)"}
    # secondary: real greedy generation (every token is the argmax of the previous step), the reference-style loop
    # (logits to the host, host argmax, one call per token) against the sampler on the device
    if secondary and fused and args.generate > 0:
      with Leg(out, "greedy_generation"):
            G = min(args.generate, N_CTX - 16)
            P0 = N_CTX - G
            prompt = toks[:P0]
            model.set_fast_decode(True)                  # (the prefill leg above switches the fused single-row path off)
            t0 = time.perf_counter(); model.logits(prompt, 0, want=True); t_pre = time.perf_counter() - t0
            res = {}
            for name, fn in (("host_loop", model.greedy), ("device_sampler", model.generate)):
                t0 = time.perf_counter()
                ids = fn(prompt, N_CTX)
                dt = time.perf_counter() - t0
                res[name] = {"new_tokens": int(len(ids) - P0), "tok_s": round((len(ids) - P0) / max(dt - t_pre, 1e-9), 1), "last": int(ids[-1])}
            out["greedy_generation"] = {"prompt_tokens": P0, "prefill_ms": round(t_pre * 1e3, 2), **res,
                                        "same_ids": res["host_loop"]["last"] == res["device_sampler"]["last"],
                                        "note": "ids generated up to n = %d; tok/s = new ids / (wall time - prompt processing time)" % N_CTX}
    # secondary: the reference's OWN caller on this path -- its unmodified tinyllama.cpp (TinyLlama::logits per token,
    # logits read on the host, host argmax: tinyllama.cpp:395-440) compiled against this repository's gten headers
    # (oracle/Makefile `dropin`, built where /root/reference exists; the .so travels).  gten/modules.h records the
    # single-row module calls and runs them as one fused decoder step.
    if secondary and fused and args.generate > 0 and args.mode == "q4":
        try:
            out["dropin"] = dropin_leg(host, cfg, args, toks, hip)
        except Exception as e:
            out["dropin"] = {"tok_s": None, "note": "drop-in leg unavailable: %r" % (e,)}
    # the CPU path beside it: once per run, on rank 0, after the other ranks have exited
    if not args.no_cpu_baseline and not args.brief and not stub:
        try:
            out["cpu_baseline"] = cpu_baseline(host, cfg, args.mode, args.seed, args.cpu_steps)
        except Exception as e:                      # the baseline is a reported extra, never the product path
            out["cpu_baseline"] = {"value": None, "unit": "tok/s", "cores": 0, "kind": "unavailable", "sample": repr(e)}
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
