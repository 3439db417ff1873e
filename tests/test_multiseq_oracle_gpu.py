"""-m gpu: multi-sequence decode against the ORACLE and the reference's golden fixtures --
not against this repository's own single-sequence decoder (tests/test_multiseq_gpu.py does that, bit for bit up to
8 sequences).  The wide path (16-64 sequences: W.x on the matrix cores, grouped attention with fused p.V terms)
deliberately differs from the single-sequence kernels in f32 summation order, so it is held directly to the bands
of SURVEY 8(c): f16 max |dlogit| <= 0.03 with identical greedy ids, q8/q4 rms <= 0.10 / max <= 0.5 (x logit std /
0.91), top-1 equal wherever the reference's top-1/top-2 gap is clear."""
import os

import numpy as np
import pytest

from gpu_common import hip, record_margin  # noqa: F401
from __graft_entry__ import load_package
from helpers import MODES, Q4, Q8, tiny_config
from test_golden_gpu import band
from test_model_gpu import check_logits, host_cfg

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,wd,ad", MODES())
@pytest.mark.parametrize("n_seq", [8, 16, 32, 64, 128, 256])
def test_batch_slots_against_the_oracle(hip, oracle, name, wd, ad, n_seq):
    """every watched slot of an n_seq batch decodes its own token stream from n = 1 across the 256-position attention
    chunk boundary; its logits are compared with the oracle model run on the same stream (own K/V history each)"""
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(wd, ad, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2)
    cfg = host_cfg(ocfg)
    N = 262
    checks = (1, 2, 33, 100, 255, 256, 257, N)
    # (lanes of 64 -- f16 -- or of 128 rows -- q8 / q4: both sides of every seam)
    watch = (0, 1, n_seq // 2, n_seq - 1) if n_seq <= 64 else tuple(sorted({0, 63, 64, 127, 128, n_seq - 1} & set(range(n_seq))))
    streams = [host.synthetic_tokens(N, seed=500 + q, n_vocab=cfg.n_vocab) for q in range(n_seq)]
    batch = host.batch(cfg, n_seq)
    weights = [host.synth_weight(cfg, 2468, i) for i in range(len(cfg.weight_shapes()))]
    for i, w in enumerate(weights):
        batch.set_weight(i, w)
    for q in range(n_seq):
        batch.decode_begin(q, streams[q])
    got = {}
    for n in range(1, N + 1):
        batch.decode_step(n, True)
        if n in checks:
            got[n] = {q: (batch.decode_result(q, n), batch.logits(q).copy()) for q in watch}
    batch.close()
    worst = [0.0, 0.0]
    for q in watch:
        om = oracle.model(ocfg)
        for i, w in enumerate(weights):
            om.set_weight(i, w)
        for n in range(1, N + 1):
            want = om.logits(streams[q][:n], n - 1)
            if n in checks:
                gid, glog = got[n][q]
                assert np.isfinite(glog).all()
                std = float(want.std())
                rms, mx = check_logits(name, glog, want, std)
                worst = [max(worst[0], rms), max(worst[1], mx)]
                top2 = np.sort(want)[-2:]
                assert gid == int(np.argmax(glog))                     # the device argmax is the argmax of these logits
                if name == "f16" and top2[1] - top2[0] > 0.03 * max(std / 0.91, 1.0):
                    assert gid == int(np.argmax(want)), (name, n_seq, q, n)
        om.close()
    print(f"{name} S={n_seq}: worst rms {worst[0]:.4g} max {worst[1]:.4g} against the oracle")


def test_wide_batch_with_three_heads_per_group_takes_the_per_head_attention(hip, oracle):
    """3 query heads per kv head is not a group size the grouped attention kernels serve: a wide batch (16 sequences, W.x on the
    matrix cores) then runs the per-head one-launch attention kernel with the sequence in the grid -- held to the oracle here,
    slot by slot, across the attention chunk boundary (round 4: this combination had no test)"""
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(Q4, Q8, n_embd=768, n_ffn=512, n_heads=12, n_kv_heads=4, max_ctx=320, n_layers=2)
    cfg = host_cfg(ocfg)
    n_seq, N = 16, 262
    checks = (1, 2, 40, 255, 256, 257, N)
    watch = (0, 7, 15)
    streams = [host.synthetic_tokens(N, seed=900 + q, n_vocab=cfg.n_vocab) for q in range(n_seq)]
    weights = [host.synth_weight(cfg, 8642, i) for i in range(len(cfg.weight_shapes()))]
    batch = host.batch(cfg, n_seq)
    for i, w in enumerate(weights):
        batch.set_weight(i, w)
    for q in range(n_seq):
        batch.decode_begin(q, streams[q])
    got = {}
    for n in range(1, N + 1):
        batch.decode_step(n, n % 2 == 0)
        if n in checks:
            got[n] = {q: (batch.decode_result(q, n), batch.logits(q).copy()) for q in watch}
    batch.close()
    for q in watch:
        om = oracle.model(ocfg)
        for i, w in enumerate(weights):
            om.set_weight(i, w)
        for n in range(1, N + 1):
            want = om.logits(streams[q][:n], n - 1)
            if n in checks:
                gid, glog = got[n][q]
                check_logits("q4", glog, want, float(want.std()))
                assert gid == int(np.argmax(glog))
        om.close()


@pytest.fixture(scope="module")
def full_golden():
    path = os.path.join(G, "full_model_golden.npz")
    if not os.path.exists(path):
        pytest.skip("full_model_golden.npz not generated")
    return np.load(path)


@pytest.mark.parametrize("name,wd,ad,S", [m + (64,) for m in MODES()] + [m + (S,) for m in MODES() if m[0] == "q4" for S in (128, 256, 384, 512)])
def test_full_size_64_sequences_against_reference_golden(hip, full_golden, name, wd, ad, S):
    """TinyLlama-1.1B, 64 sequences sharing the weight passes: every slot carries the reference's golden token stream
    (15-id prompt through the slot's operator path, then 23 teacher-forced steps of the whole batch on the matrix-core
    decode kernels) and EVERY slot is held to the reference's logits at every step"""
    g = full_golden
    pkg = load_package()
    host = pkg.load_host()
    cfg = host.default_config(wd, ad)
    cfg.max_ctx = 256
    batch = host.batch(cfg, S)                  # (128: one lane of 128 rows; 256: two of them in one decoder)
    batch.load_synthetic(int(g["seed"][0]))
    toks = g[f"{name}.avx.tokens"]
    probe = g["probe_ids"]
    steps = g[f"{name}.avx.top_ids"].shape[0]

    def hold(step, lg, what):
        ids = g[f"{name}.avx.top_ids"][step]
        ref_vals = np.concatenate([g[f"{name}.avx.top_logits"][step], g[f"{name}.avx.probes"][step]])
        got_vals = np.concatenate([lg[ids], lg[probe]])
        std = float(g[f"{name}.avx.stats"][step][1])
        rms, mx = band(name, got_vals - ref_vals, std)
        assert abs(float(lg.mean()) - float(g[f"{name}.avx.stats"][step][0])) < 0.02, what
        assert abs(float(lg.std()) - std) < 0.02 * max(std, 1.0), what
        gap = float(g[f"{name}.avx.top_logits"][step][0] - g[f"{name}.avx.top_logits"][step][1])
        same = int(np.argmax(lg)) == int(ids[0])
        if name == "f16":
            assert same or gap < 0.03, (what, gap)
        elif gap > 0.5 * max(std / 0.91, 1.0):
            assert same, (what, gap)
        return rms, mx

    worst = [0.0, 0.0]
    # (up to 256 sequences EVERY slot is held; three and four lanes of 128 rows: both sides of every lane seam and the ends)
    held = range(S) if S <= 256 else sorted({0, 1, 127, 128, 255, 256, 383, 384, S - 1} & set(range(S)))
    for q in range(S):
        lg = batch.prefill(q, toks[:15], want=(q in held))
        if q in held:
            hold(0, lg, (name, "prefill", q))
        batch.decode_begin(q, toks)
    for step in range(1, steps):
        n = 15 + step
        batch.decode_step(n, True)
        for q in held:
            rms, mx = hold(step, batch.logits(q), (name, "slot", q, "step", step))
            worst = [max(worst[0], rms), max(worst[1], mx)]
            assert batch.decode_result(q, n) == int(np.argmax(batch.logits(q)))
    batch.close()
    print(f"{name} S={S} full size: worst rms {worst[0]:.4f} max {worst[1]:.4f} over {len(held)} slots x {steps - 1} steps")


@pytest.fixture(scope="module")
def extra_golden():
    path = os.path.join(G, "full_extra_golden.npz")
    if not os.path.exists(path):
        pytest.skip("full_extra_golden.npz not generated")
    return np.load(path)


@pytest.mark.parametrize("name,wd,ad,S", [m + (S,) for m in MODES() if m[0] == "q4" for S in (64, 128)] + [m + (64,) for m in MODES() if m[0] != "q4"])
def test_wide_path_long_context_probe(hip, full_golden, extra_golden, name, wd, ad, S):
    """the 64-sequence path -- and, q4, a lane of 128 rows -- at the BASELINE.json metric point: every slot steps the reference's
    long-context token stream from n = 1 to n = 2048; slots 0 / 31 / S - 1 are held to the reference's probe at n = 257, 1024,
    2047, 2048.  q4 / q8: the yardstick of tests/test_golden_gpu.py::test_long_context_probe_q4 (1.35 x the reference's own
    AVX-vs-scalar spread at that length, max <= 0.5) plus the reference's top-1 wherever its gap is clear;
    f16: max |dlogit| inside the 0.03 band and the reference's top-1 unless its gap is below 0.03 (tests/test_prefill_gpu.py)."""
    g = full_golden if name == "q4" else extra_golden
    if f"long.{name}.ns" not in g:
        pytest.skip("long-context probe not in the fixture")
    pkg = load_package()
    host = pkg.load_host()
    cfg = host.default_config(wd, ad)
    batch = host.batch(cfg, S)
    batch.load_synthetic(int(g["seed"][0]))
    toks = host.synthetic_tokens(2048, seed=int(g["token_seed"][0]))
    for q in range(S):
        batch.decode_begin(q, toks)
    ns = [int(n) for n in g[f"long.{name}.ns"]]
    probe = g["probe_ids"]
    for n in range(1, 2049):
        batch.decode_step(n, True)
        if n not in ns:
            continue
        own = g[f"long.{name}.n{n}.probes"] - g[f"long.{name}.n{n}.probes.scalar"]
        own_rms = float(np.sqrt((own * own).mean()))
        ids = g[f"long.{name}.n{n}.top_ids"]
        ref_vals = np.concatenate([g[f"long.{name}.n{n}.top_logits"], g[f"long.{name}.n{n}.probes"]])
        std = float(g[f"long.{name}.n{n}.stats"][1])
        gap = float(g[f"long.{name}.n{n}.top_logits"][0] - g[f"long.{name}.n{n}.top_logits"][1])
        for q in (0, 31, S - 1):
            lg = batch.logits(q)
            d = np.concatenate([lg[ids], lg[probe]]) - ref_vals
            rms, mx = float(np.sqrt((d * d).mean())), float(np.abs(d).max())
            print(f"{name} S={S} slot {q} n={n}: rms {rms:.4f} (reference's own spread {own_rms:.4f}) max {mx:.4f} "
                  f"top1 {int(np.argmax(lg))} ref avx {int(ids[0])} scalar {int(g[f'long.{name}.n{n}.top_ids.scalar'][0])} (gap {gap:.3f})")
            if q == 0:
                record_margin(f"{name} wide S={S} n={n}", rms, own_rms, mx)
            if name == "f16":
                assert mx <= 0.03 * max(std / 0.91, 1.0), (q, n, mx)
                assert int(np.argmax(lg)) == int(ids[0]) or gap < 0.03, (q, n, gap)
            else:
                assert rms <= 1.35 * own_rms and mx <= 0.5, (q, n, rms, own_rms, mx)
                if gap > 0.5 * max(std / 0.91, 1.0):
                    assert int(np.argmax(lg)) == int(ids[0]), (q, n, gap)
    batch.close()
