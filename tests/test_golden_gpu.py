"""-m gpu: the HIP path against the committed golden fixtures produced by the REAL reference
(tests/golden/make_golden.py).  Operators through the C-ABI; models through the C++ driver.

Tolerances: integer/byte work is exact; a different f32 summation order can flip a rounding
at a tie, so operator outputs are compared with helpers.compare_rows (values within one
quantization step, >= 97 % of bytes identical) and logits with the bands of SURVEY 8(c):
f16 max|d| <= 0.03 and identical greedy ids; q8/q4 rms <= 0.10, max <= 0.5, top-1 equal where
the reference's own top-1/top-2 gap exceeds 0.5 (the reference's AVX and scalar builds differ
from each other by rms 0.06 / max 0.27 on these weights)."""
import os

import numpy as np
import pytest

from gpu_common import hip, record_margin  # noqa: F401
from __graft_entry__ import load_package
from helpers import F16, F32, MODES, Q4, Q8, compare_rows, row_bytes, tiny_config

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ops_against_reference_golden(hip):
    g = np.load(os.path.join(G, "ops_golden.npz"))
    for name, wd, ad in MODES():
        x, w = g[f"{name}.matmul.x"], g[f"{name}.matmul.w"]
        xd, wdv = hip.upload(x), hip.upload_weight(w, wd, 96, 256)
        for sp in (0, 2):
            for od, odn in ((ad, "a"), (F32, "f32")):
                want = g[f"{name}.matmul.out.avx.sp{sp}.{odn}"]
                out = hip.upload(np.zeros_like(want))
                hip.matmul_2d(xd, ad, wdv, wd, out, od, 3, 256, 96, sp)
                compare_rows(out.download(shape=want.shape)[sp:], want[sp:], od, 96, f"matmul {name}", atol=8e-6)
        tab, toks = g[f"{name}.embed.table"], g[f"{name}.embed.tokens"]
        want = g[f"{name}.embed.out"]
        out = hip.upload(np.zeros_like(want))
        hip.token_embed(hip.upload_weight(tab, wd, 50, 256), wd, 50, hip.upload(toks), out, ad, 5, 256, 1)
        assert np.array_equal(out.download(shape=want.shape), want)
    for ad, an in ((F16, "f16"), (Q8, "q8")):
        x, y, wn = g[f"{an}.row.x"], g[f"{an}.row.y"], g[f"{an}.row.w"]
        n, d = 4, 256
        xd, yd, wd_ = hip.upload(x), hip.upload(y), hip.upload(wn)
        for sp in (0, 3):
            out = hip.upload(np.zeros_like(x)); hip.rms_norm(xd, ad, wd_, out, n, d, sp)
            compare_rows(out.download(shape=x.shape), g[f"{an}.rms_norm.sp{sp}"], ad, d, "rms_norm")
            buf = hip.upload(x); hip.rotary_emb(buf, ad, n, d, 64, sp)
            compare_rows(buf.download(shape=x.shape), g[f"{an}.rope.sp{sp}"], ad, d, "rope", min_exact=0.999)
            out = hip.upload(np.zeros_like(x)); hip.silu(xd, out, ad, n, d, sp)
            compare_rows(out.download(shape=x.shape), g[f"{an}.silu.sp{sp}"], ad, d, "silu", min_exact=0.995)
            out = hip.upload(np.zeros_like(x)); hip.mul(xd, yd, out, ad, n, d, sp)
            assert np.array_equal(out.download(shape=x.shape), g[f"{an}.mul.sp{sp}"])
            out = hip.upload(np.zeros_like(x)); hip.add(xd, yd, out, ad, n, d, sp)
            assert np.array_equal(out.download(shape=x.shape), g[f"{an}.add.sp{sp}"])
        xr = np.zeros((2048, row_bytes(ad, 128)), np.uint8)
        xr[2044:] = g[f"{an}.rope_far.x"]
        buf = hip.upload(xr); hip.rotary_emb(buf, ad, 2048, 128, 64, 2044)
        compare_rows(buf.download(shape=xr.shape)[2044:], g[f"{an}.rope_far.out"], ad, 128, "rope@2047", min_exact=0.999)
        for n_att, sp in ((5, 0), (33, 0), (40, 0), (40, 39)):
            key = f"{an}.attn.n{n_att}.sp{sp}"
            want = g[key + ".out.avx"]
            out = hip.upload(np.zeros_like(want))
            hip.qkv_attn(hip.upload(g[key + ".q"]), hip.upload(g[key + ".k"]), hip.upload(g[key + ".v"]), out, ad, n_att, 8, 2, 64, sp)
            compare_rows(out.download(shape=want.shape)[sp:], want[sp:], ad, 512, key, min_exact=0.90, steps=2.0, atol=2e-4)


def band(name, d, std):
    s = max(std / 0.91, 1.0)
    rms, mx = float(np.sqrt((d * d).mean())), float(np.abs(d).max())
    if name == "f16":
        assert mx <= 0.03 * s, (name, mx)
    else:
        assert rms <= 0.10 * s and mx <= 0.5 * s, (name, rms, mx)
    return rms, mx


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_tiny_model_against_reference_golden(hip, name, wd, ad):
    g = np.load(os.path.join(G, "tiny_model_golden.npz"))
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(wd, ad, n_heads=4, n_kv_heads=2)
    cfg = pkg.HostConfig(**{k: getattr(ocfg, k) for k, _ in ocfg._fields_})
    m = host.model(cfg)
    for i in range(m.n_weights()):
        m.set_weight(i, host.synth_weight(cfg, int(g["seed"][0]), i))
    toks, want = g[f"{name}.tokens"], g[f"{name}.logits.avx"]
    for step in range(want.shape[0]):
        n = 9 + step
        got = m.logits(toks[:n], 0 if step == 0 else n - 1)
        band(name, got - want[step], float(want[step].std()))
        if name == "f16":
            assert int(np.argmax(got)) == int(np.argmax(want[step]))
    m.close()


@pytest.fixture(scope="module")
def full_golden():
    path = os.path.join(G, "full_model_golden.npz")
    if not os.path.exists(path):
        pytest.skip("full_model_golden.npz not generated")
    return np.load(path)


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_full_size_model_against_reference_golden(hip, full_golden, name, wd, ad):
    """TinyLlama-1.1B on the seeded synthetic weights: 15-id prefill (operator path) then 23 greedy
    decode steps (fused path), teacher-forced with the reference's ids, against the reference's own
    TinyLlama class."""
    g = full_golden
    pkg = load_package()
    host = pkg.load_host()
    cfg = host.default_config(wd, ad)
    cfg.max_ctx = 256
    m = host.model(cfg)
    m.load_synthetic(int(g["seed"][0]))
    toks = g[f"{name}.avx.tokens"]
    probe = g["probe_ids"]
    worst = [0.0, 0.0]
    agree = 0
    steps = g[f"{name}.avx.top_ids"].shape[0]
    for step in range(steps):
        n = 15 + step
        lg = m.logits(toks[:n], 0 if step == 0 else n - 1)
        ids = g[f"{name}.avx.top_ids"][step]
        ref_vals = np.concatenate([g[f"{name}.avx.top_logits"][step], g[f"{name}.avx.probes"][step]])
        got_vals = np.concatenate([lg[ids], lg[probe]])
        std = float(g[f"{name}.avx.stats"][step][1])
        rms, mx = band(name, got_vals - ref_vals, std)
        worst = [max(worst[0], rms), max(worst[1], mx)]
        assert abs(float(lg.mean()) - float(g[f"{name}.avx.stats"][step][0])) < 0.02
        assert abs(float(lg.std()) - std) < 0.02 * max(std, 1.0)
        gap = float(g[f"{name}.avx.top_logits"][step][0] - g[f"{name}.avx.top_logits"][step][1])
        same = int(np.argmax(lg)) == int(ids[0])
        agree += int(same)
        if name == "f16":
            assert same or gap < 0.03, (step, gap)        # token-for-token at fp16
        elif gap > 0.5 * max(std / 0.91, 1.0):
            assert same, (name, step, gap)
    print(f"{name}: worst rms {worst[0]:.4f} max {worst[1]:.4f}; greedy agreement {agree}/{steps}")
    m.close()


def test_long_context_probe_q4(hip, full_golden):
    """decode at n = 257, 1024, 2047, 2048 reached by stepping from n = 1 with teacher-forced ids:
    crosses every attention chunk boundary; the last step is the BASELINE.json metric point"""
    g = full_golden
    if "long.q4.ns" not in g:
        pytest.skip("long-context probe not in the fixture")
    pkg = load_package()
    host = pkg.load_host()
    cfg = host.default_config(Q4, Q8)
    m = host.model(cfg)
    m.load_synthetic(int(g["seed"][0]))
    toks = host.synthetic_tokens(2048, seed=int(g["token_seed"][0]))
    m.decode_begin(toks)
    ns = [int(n) for n in g["long.q4.ns"]]
    probe = g["probe_ids"]
    prev = 1
    for n in ns:
        for k in range(prev, n):
            m.decode_step(k, True)
        lg = m.logits(toks[:n], n - 1)             # fused path, logits to the host
        prev = n + 1
        ids = g[f"long.q4.n{n}.top_ids"]
        ref_vals = np.concatenate([g[f"long.q4.n{n}.top_logits"], g[f"long.q4.n{n}.probes"]])
        got_vals = np.concatenate([lg[ids], lg[probe]])
        std = float(g[f"long.q4.n{n}.stats"][1])
        # yardstick at THIS context length: the reference's own AVX build vs its own scalar build on
        # the same ids (rms 0.074 at n=257 growing to 0.100 at n=2047; at n=2048 their top-1 differ).
        # The K/V history of every implementation carries its own rounding noise, so the spread grows
        # with n; the GPU must stay within 1.35x of the reference's self-disagreement, max <= 0.5.
        own = g[f"long.q4.n{n}.probes"] - g[f"long.q4.n{n}.probes.scalar"]
        own_rms = float(np.sqrt((own * own).mean()))
        d = got_vals - ref_vals
        rms, mx = float(np.sqrt((d * d).mean())), float(np.abs(d).max())
        print(f"n={n}: rms {rms:.4f} (reference's own spread {own_rms:.4f}) max {mx:.4f} (std {std:.3f}) "
              f"top1 {int(np.argmax(lg))} ref avx {int(ids[0])} scalar {int(g[f'long.q4.n{n}.top_ids.scalar'][0])}")
        record_margin(f"q4 fused step n={n}", rms, own_rms, mx)
        assert rms <= 1.35 * own_rms and mx <= 0.5, (n, rms, own_rms, mx)
        # ... and the reference's top-1 wherever its own top-1 / top-2 gap is clear (the fixture's gaps at these
        # lengths are 0.02-0.19 on synthetic weights, so this seldom binds -- at n = 257 the fused step's top-1 differs from
        # both reference builds inside a gap of 0.02 -- but the rule is the rule)
        gap = float(g[f"long.q4.n{n}.top_logits"][0] - g[f"long.q4.n{n}.top_logits"][1])
        print(f"      reference gap {gap:.3f}; margin used: rms {rms / (1.35 * own_rms):.2f} of the bar, max {mx / 0.5:.2f}")
        if gap > 0.5 * max(std / 0.91, 1.0):
            assert int(np.argmax(lg)) == int(ids[0]), (n, gap)
    m.close()
