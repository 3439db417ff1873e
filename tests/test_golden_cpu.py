"""not gpu: the oracle against the committed golden fixtures (tests/golden/*.npz), which
were produced by the REAL reference (tests/golden/make_golden.py, build container only).
This is what pins the oracle where oracle/_ref cannot be built.  Everything is bit-exact:
the oracle in "avx order" vs the reference's AVX build, in "scalar order" vs its scalar build."""
import os

import numpy as np
import pytest

from helpers import F16, F32, MODES, Q4, Q8, row_bytes, tiny_config

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def test_converter_pins(oracle):
    """weight quantizers == tinyllama_to_gten.py:24-148 (round half to even, zero blocks, nibble packing)"""
    g = load("converter_pins.npz")
    w = g["w"]
    assert np.array_equal(oracle.quantize_weight(w, Q8), g["q8"])
    assert np.array_equal(oracle.quantize_weight(w, Q4), g["q4"])
    assert np.array_equal(oracle.quantize_weight(w, F16), g["f16"])
    # the ties really are ties: 0.5 -> 0, 1.5 -> 2, 2.5 -> 2 under half-to-even
    q8 = g["q8"].reshape(8, 8, 34)[3, 0, 2:].view(np.int8)
    assert list(q8[:6]) == [127, 0, 2, 2, 0, -2]
    q4 = g["q4"].reshape(8, 8, 18)[4, 0, 2:]
    assert [int(b >> 4) - 7 for b in q4[:4]] == [7, 0, 2, 2]


@pytest.mark.parametrize("kind", ["avx", "scalar"])
def test_ops_golden(oracle, kind):
    g = load("ops_golden.npz")
    oracle.set_simd(kind == "avx")
    try:
        for name, wd, ad in MODES():
            x, w = g[f"{name}.matmul.x"], g[f"{name}.matmul.w"]
            for sp in (0, 2):
                for od, odn in ((ad, "a"), (F32, "f32")):
                    o = np.zeros((3, row_bytes(od, 96)), np.uint8)
                    oracle.matmul_2d(x, ad, w, wd, o, od, 3, 256, 96, sp)
                    assert np.array_equal(o, g[f"{name}.matmul.out.{kind}.sp{sp}.{odn}"]), (name, sp, odn)
            o = np.zeros((5, row_bytes(ad, 256)), np.uint8)
            oracle.token_embed(g[f"{name}.embed.table"], wd, g[f"{name}.embed.tokens"], o, ad, 256, 1)
            assert np.array_equal(o, g[f"{name}.embed.out"])
        for ad, an in ((F16, "f16"), (Q8, "q8")):
            x, y, wn = g[f"{an}.row.x"], g[f"{an}.row.y"], g[f"{an}.row.w"]
            n, d = 4, 256
            for sp in (0, 3):
                o = np.zeros_like(x); oracle.rms_norm(x, ad, wn, o, n, d, sp); assert np.array_equal(o, g[f"{an}.rms_norm.sp{sp}"])
                a = x.copy(); oracle.rotary_emb(a, ad, n, d, 64, sp); assert np.array_equal(a, g[f"{an}.rope.sp{sp}"])
                o = np.zeros_like(x); oracle.silu(x, o, ad, n, d, sp); assert np.array_equal(o, g[f"{an}.silu.sp{sp}"])
                o = np.zeros_like(x); oracle.mul(x, y, o, ad, n, d, sp); assert np.array_equal(o, g[f"{an}.mul.sp{sp}"])
                o = np.zeros_like(x); oracle.add(x, y, o, ad, n, d, sp); assert np.array_equal(o, g[f"{an}.add.sp{sp}"])
            xr = np.zeros((2048, row_bytes(ad, 128)), np.uint8)
            xr[2044:] = g[f"{an}.rope_far.x"]
            oracle.rotary_emb(xr, ad, 2048, 128, 64, 2044)
            assert np.array_equal(xr[2044:], g[f"{an}.rope_far.out"])
            for n_att, sp in ((5, 0), (33, 0), (40, 0), (40, 39)):
                key = f"{an}.attn.n{n_att}.sp{sp}"
                o = np.zeros((n_att, row_bytes(ad, 512)), np.uint8)
                oracle.qkv_attn(g[key + ".q"], g[key + ".k"], g[key + ".v"], o, ad, n_att, 8, 2, 64, sp)
                assert np.array_equal(o, g[key + f".out.{kind}"]), key
    finally:
        oracle.set_simd(True)


def _host():
    from __graft_entry__ import load_package
    pkg = load_package()
    pkg.build.build_all()
    return pkg, pkg.load_host()


@pytest.mark.parametrize("kind", ["avx", "scalar"])
@pytest.mark.parametrize("name,wd,ad", MODES())
def test_tiny_model_golden(oracle, kind, name, wd, ad):
    g = load("tiny_model_golden.npz")
    pkg, host = _host()
    ocfg = tiny_config(wd, ad, n_heads=4, n_kv_heads=2)
    cfg = pkg.HostConfig(**{k: getattr(ocfg, k) for k, _ in ocfg._fields_})
    oracle.set_simd(kind == "avx")
    try:
        m = oracle.model(ocfg)
        for i in range(m.n_weights()):
            m.set_weight(i, host.synth_weight(cfg, int(g["seed"][0]), i))
        toks = g[f"{name}.tokens"]
        want = g[f"{name}.logits.{kind}"]
        for step in range(want.shape[0]):
            n = 9 + step
            got = m.logits(toks[:n], 0 if step == 0 else n - 1)
            assert np.array_equal(got, want[step]), (name, kind, step)
        m.close()
    finally:
        oracle.set_simd(True)


def test_full_model_golden_q4_first_steps(oracle):
    """TinyLlama-1.1B q4 on the seeded synthetic weights: the oracle reproduces the reference's own
    TinyLlama class bit for bit (prefill of 15 ids + 3 decode steps; ~30 s of CPU)."""
    path = os.path.join(G, "full_model_golden.npz")
    if not os.path.exists(path):
        pytest.skip("full_model_golden.npz not generated yet")
    g = np.load(path)
    pkg, host = _host()
    cfg = host.default_config(Q4, Q8)
    from oracle import orc
    ocfg = orc.Config(**{k: getattr(cfg, k) for k, _ in cfg._fields_})
    ocfg.max_ctx = 64
    m = oracle.model(ocfg)
    for i in range(m.n_weights()):
        m.set_weight(i, host.synth_weight(cfg, int(g["seed"][0]), i))
    toks = g["q4.avx.tokens"]
    for step in range(4):
        n = 15 + step
        lg = m.logits(toks[:n], 0 if step == 0 else n - 1)
        ids = g["q4.avx.top_ids"][step]
        assert np.array_equal(lg[ids], g["q4.avx.top_logits"][step]), step
        assert int(np.argmax(lg)) == int(ids[0]) == int(toks[n])
        assert np.array_equal(lg[g["probe_ids"]], g["q4.avx.probes"][step])
    m.close()


def test_round_half_away_by_one_addition():
    """The HIP quantizer rounds x * scale half away from zero (gten/quants.h:62: roundf) as
    trunc(x + copysign(0x1.fffffep-2, x)) (csrc/gten_dev.h: round_half_away_i).  Exhaustive over every
    binary32 value from 2^-30 up to 2^22 (a Q8 quant never leaves +-127.0001); below 2^-30 both give 0."""
    c = np.float32(float.fromhex("0x1.fffffep-2"))
    assert c < np.float32(0.5) and np.nextafter(c, np.float32(1)) == np.float32(0.5)
    for e in range(-30, 22):
        base = np.float32(2.0 ** e).view(np.uint32)
        for lo in range(0, 1 << 23, 1 << 22):
            x = (base + np.arange(lo, lo + (1 << 22), dtype=np.uint32)).astype(np.uint32).view(np.float32)
            got = np.trunc(x + c)                                   # one f32 addition (RN), then truncation
            want = np.floor(x.astype(np.float64) + 0.5)             # half away from zero, exact in binary64
            assert np.array_equal(got.astype(np.float64), want), e
            if lo == 0:
                assert np.array_equal(np.trunc(-x + np.copysign(c, -x)).astype(np.float64), -want)
    tiny = np.float32(2.0 ** -31)
    assert np.trunc(tiny + c) == 0


def test_div127_markstein():
    """The HIP Q8 quantizer divides the block absmax by 127 with a 3-instruction Markstein sequence
    (csrc/gten_dev.h: div127) instead of the IEEE division expansion.  It must be the correctly rounded
    quotient the reference computes (gten/quants.h:52-58: absmax / 127.0f): exhaustive over every binary32
    significand of one binade -- binary scaling carries the result to all other binades."""
    y = np.float32(1.0) / np.float32(127.0)
    assert float(y).hex() == "0x1.0204080000000p-7"            # the constant in div127
    a = np.arange(2 ** 23, 2 ** 24, dtype=np.int64).astype(np.float32)
    ref = a / np.float32(127.0)                                   # IEEE division, correctly rounded
    # binary64 carries each step exactly (24 x 24-bit products, sums far inside 53 bits); one rounding per cast
    q0 = (a.astype(np.float64) * np.float64(y)).astype(np.float32)
    r64 = a.astype(np.float64) - 127.0 * q0.astype(np.float64)    # fma(-127, q0, a)
    r = r64.astype(np.float32)
    assert np.array_equal(r.astype(np.float64), r64)              # the remainder is exact
    q1 = (r.astype(np.float64) * np.float64(y) + q0.astype(np.float64)).astype(np.float32)
    assert np.array_equal(q1, ref)
