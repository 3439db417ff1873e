"""Pin the oracle (oracle/gten_oracle.c) to the REAL reference (oracle/_ref).

Runs only where oracle/_ref has been built (this container: it needs
/root/reference; a prebuilt copy also travels to the GPU box).  Every check is
BIT-EXACT: the oracle in "avx order" must equal the reference's AVX build
(README.md:25 flags) and in "scalar order" its scalar build (README.md:17).
"""
import numpy as np
import pytest

from helpers import F16, F32, MODES, Q4, Q8, act_rows, random_weights, rng, row_bytes, tiny_config, weight_rows


def test_fp16_conversions_exhaustive_half_and_sampled_float(ref_pair):
    orc_, ref = ref_pair
    for h in range(0, 65536, 1):
        a, b = orc_.fp16_to_fp32(h), ref.fp16_to_fp32(h)
        assert (a == b) or (a != a and b != b), h
    r = rng(1)
    bits = np.concatenate([
        r.integers(0, 2**32, size=200000, dtype=np.uint64).astype(np.uint32),
        # every half value exactly, its neighbours and the exact midpoints between halves
        (np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32).view(np.uint32)),
    ])
    halves = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16).astype(np.float64)
    mids = ((halves[:-1] + halves[1:]) / 2).astype(np.float32)
    bits = np.concatenate([bits, mids.view(np.uint32), mids.view(np.uint32) + 1, mids.view(np.uint32) - 1,
                           np.array([0x477fefff, 0x477ff000, 0x477ff001, 0x7f800000, 0x7f800001, 0xffc00000,
                                     0x33000000, 0x33000001, 0x32ffffff, 0x00000001, 0x80000000], dtype=np.uint32)])
    for u in bits:
        f = np.array([u], dtype=np.uint32).view(np.float32)[0]
        assert orc_.fp32_to_fp16(f) == ref.fp32_to_fp16(f), hex(int(u))


@pytest.mark.parametrize("n", [32, 64, 40, 33, 5, 2048])
def test_q8_activation_quantizer(ref_pair, n):
    orc_, ref = ref_pair
    r = rng(n)
    x = r.standard_normal((4, n)).astype(np.float32)
    x[1, : min(n, 32)] = 0.0                       # an all-zero block -> delta 0, scale 0
    x[2, 0] = 127.0; x[2, 1] = 0.5; x[2, 2] = -0.5; x[2, 3] = 1.5  # exact .5 ties with delta = 1
    a, b = orc_.quantize_rows(x, Q8), ref.quantize_rows(x, Q8)
    assert np.array_equal(a, b)
    assert np.array_equal(orc_.dequantize_rows(a, Q8, n), ref.dequantize_rows(a, Q8, n))


def test_q4_dequantize(ref_pair):
    orc_, ref = ref_pair
    r = rng(3)
    blocks = r.integers(0, 256, size=(3, row_bytes(Q4, 256)), dtype=np.uint8)
    # scales: keep them finite halves
    for row in blocks:
        row.reshape(-1, 18)[:, 1] &= 0x3b
    a, b = orc_.dequantize_rows(blocks, Q4, 256), ref.dequantize_rows(blocks, Q4, 256)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("name,wd,ad", MODES())
@pytest.mark.parametrize("n,d_in,d_out,sp", [(3, 256, 96, 0), (3, 256, 96, 2), (1, 2048, 64, 0), (2, 5632, 32, 1)])
def test_matmul_2d(ref_pair, name, wd, ad, n, d_in, d_out, sp):
    orc_, ref = ref_pair
    r = rng(hash((name, n, d_in)) % 2**31)
    x, _ = act_rows(orc_, r, n, d_in, ad)
    w, _ = weight_rows(orc_, r, d_out, d_in, wd)
    for od in (ad, F32):
        o1 = np.zeros((n, row_bytes(od, d_out)), np.uint8)
        o2 = o1.copy()
        orc_.matmul_2d(x, ad, w, wd, o1, od, n, d_in, d_out, sp)
        ref.matmul_2d(x, ad, w, wd, o2, od, n, d_in, d_out, sp)
        assert np.array_equal(o1, o2), (name, od)
        assert o1[sp:].any()


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_token_embed(ref_pair, name, wd, ad):
    orc_, ref = ref_pair
    r = rng(5)
    w, _ = weight_rows(orc_, r, 50, 256, wd)
    toks = np.array([3, 49, 0, 7, 7], np.int32)
    o1 = np.zeros((5, row_bytes(ad, 256)), np.uint8)
    o2 = o1.copy()
    orc_.token_embed(w, wd, toks, o1, ad, 256, 1)
    ref.token_embed(w, wd, toks, o2, ad, 256, 1)
    assert np.array_equal(o1, o2)
    assert not o1[0].any() and o1[1:].any()


@pytest.mark.parametrize("ad", [F16, Q8])
def test_rms_norm_rope_silu_mul_add(ref_pair, ad):
    orc_, ref = ref_pair
    r = rng(11)
    n, d = 4, 256
    x, _ = act_rows(orc_, r, n, d, ad)
    y, _ = act_rows(orc_, r, n, d, ad)
    w = (1 + 0.05 * r.standard_normal(d)).astype(np.float16)
    for sp in (0, 3):
        o1 = np.zeros_like(x); o2 = np.zeros_like(x)
        orc_.rms_norm(x, ad, w, o1, n, d, sp); ref.rms_norm(x, ad, w, o2, n, d, sp)
        assert np.array_equal(o1, o2), "rms_norm"
        a1 = x.copy(); a2 = x.copy()
        orc_.rotary_emb(a1, ad, n, d, 64, sp); ref.rotary_emb(a2, ad, n, d, 64, sp)
        assert np.array_equal(a1, a2), "rope"
        o1[:] = 0; o2[:] = 0
        orc_.silu(x, o1, ad, n, d, sp); ref.silu(x, o2, ad, n, d, sp)
        assert np.array_equal(o1, o2), "silu"
        a1 = x.copy(); a2 = x.copy()
        orc_.silu(a1, a1, ad, n, d, sp); ref.silu(a2, a2, ad, n, d, sp)
        assert np.array_equal(a1, a2) and np.array_equal(a1[sp:], o1[sp:]), "silu_inplace"
        o1[:] = 0; o2[:] = 0
        orc_.mul(x, y, o1, ad, n, d, sp); ref.mul(x, y, o2, ad, n, d, sp)
        assert np.array_equal(o1, o2), "mul"
        a1 = x.copy(); a2 = x.copy()
        orc_.mul(a1, y, a1, ad, n, d, sp); ref.mul(a2, y, a2, ad, n, d, sp)
        assert np.array_equal(a1, a2), "mul_inplace"
        o1[:] = 0; o2[:] = 0
        orc_.add(x, y, o1, ad, n, d, sp); ref.add(x, y, o2, ad, n, d, sp)
        assert np.array_equal(o1, o2), "add"


def test_rope_far_positions(ref_pair):
    """angles up to 2047 rad: libm range reduction must agree (gten/ops.h:743-746)."""
    orc_, ref = ref_pair
    r = rng(13)
    n, d = 2048, 128
    x, _ = act_rows(orc_, r, n, d, F16)
    a1 = x.copy(); a2 = x.copy()
    orc_.rotary_emb(a1, F16, n, d, 64, 2040); ref.rotary_emb(a2, F16, n, d, 64, 2040)
    assert np.array_equal(a1, a2)
    assert not np.array_equal(a1[2047], x[2047])


@pytest.mark.parametrize("ad", [F16, Q8])
@pytest.mark.parametrize("n,sp", [(5, 0), (33, 0), (40, 0), (40, 39), (70, 64)])
def test_qkv_attn(ref_pair, ad, n, sp):
    orc_, ref = ref_pair
    r = rng(n * 7 + sp)
    H, G, dh = 8, 2, 64
    q, _ = act_rows(orc_, r, n, H * dh, ad)
    k, _ = act_rows(orc_, r, n, G * dh, ad)
    v, _ = act_rows(orc_, r, n, G * dh, ad)
    o1 = np.zeros((n, row_bytes(ad, H * dh)), np.uint8)
    o2 = o1.copy()
    orc_.qkv_attn(q, k, v, o1, ad, n, H, G, dh, sp)
    ref.qkv_attn(q, k, v, o2, ad, n, H, G, dh, sp)
    assert np.array_equal(o1, o2)
    assert o1[sp:].any()


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_tiny_model_logits(ref_pair, name, wd, ad):
    """prefill of 9 tokens then 6 single-token decode steps, full logits bit-exact."""
    orc_, ref = ref_pair
    cfg = tiny_config(wd, ad)
    ws = random_weights(orc_, cfg, seed=77)
    m1, m2 = orc_.model(cfg), ref.model(cfg)
    assert m1.n_weights() == m2.n_weights() == len(ws)
    for i, w in enumerate(ws):
        assert m1.weight_bytes(i) == m2.weight_bytes(i) == w.size
        m1.set_weight(i, w); m2.set_weight(i, w)
    toks = list(rng(5).integers(3, cfg.n_vocab, size=9))
    for step in range(7):
        sp = 0 if step == 0 else len(toks) - 1
        l1, l2 = m1.logits(toks, sp), m2.logits(toks, sp)
        assert np.array_equal(l1, l2), (name, step, np.abs(l1 - l2).max())
        assert np.isfinite(l1).all() and l1.std() > 0
        toks.append(int(np.argmax(l1)))
    m1.close(); m2.close()
