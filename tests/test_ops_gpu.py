"""-m gpu: every operator of include/gten_hip.h against the oracle, through the C-ABI.

Same seeded inputs on both sides, at sizes the oracle finishes in seconds.
The comparison rule and its tolerance are in helpers.compare_rows.
"""
import numpy as np
import pytest

from gpu_common import hip  # noqa: F401
from helpers import F16, F32, MODES, Q4, Q8, act_rows, compare_rows, rng, row_bytes, weight_rows

pytestmark = pytest.mark.gpu


def up_weight(hip, blocks, dtype, rows, cols):
    return hip.upload_weight(blocks, dtype, rows, cols)


@pytest.mark.parametrize("wd", [Q8, Q4, F16])
def test_pack_weight_roundtrip(hip, oracle, wd):
    """the load-time repack holds exactly the bytes of the .gten block stream"""
    r = rng(1)
    rows, cols = 7, 256
    blocks, _ = weight_rows(oracle, r, rows, cols, wd)
    dev = up_weight(hip, blocks, wd, rows, cols)
    packed = dev.download()
    nb = cols // 32
    if wd == F16:
        assert np.array_equal(packed, blocks.reshape(-1))
        return
    qbytes = 16 if wd == Q4 else 32
    src = blocks.reshape(rows, nb, 2 + qbytes)
    ds = packed[rows * nb * qbytes:].view(np.uint16).reshape(rows, nb)
    assert np.array_equal(ds, src[:, :, :2].copy().view(np.uint16).reshape(rows, nb))
    if wd == Q4:
        assert np.array_equal(packed[: rows * nb * 16].reshape(rows, nb, 16), src[:, :, 2:])
    else:
        planes = packed[: rows * nb * 32].reshape(rows, 2, nb, 16)
        assert np.array_equal(planes[:, 0], src[:, :, 2:18])
        assert np.array_equal(planes[:, 1], src[:, :, 18:34])


@pytest.mark.parametrize("name,wd,ad", MODES())
@pytest.mark.parametrize("n,d_in,d_out,sp", [(3, 256, 96, 0), (3, 256, 96, 2), (1, 2048, 2048, 0),
                                            (2, 5632, 64, 1), (1, 2048, 5632, 0)])
def test_matmul_2d(hip, oracle, name, wd, ad, n, d_in, d_out, sp):
    r = rng(n * 1000 + d_in + d_out)
    x, _ = act_rows(oracle, r, n, d_in, ad)
    w, _ = weight_rows(oracle, r, d_out, d_in, wd)
    xd, wdv = hip.upload(x), up_weight(hip, w, wd, d_out, d_in)
    for od in (ad, F32):
        want = np.full((n, row_bytes(od, d_out)), 0xAB, np.uint8)
        oracle.matmul_2d(x, ad, w, wd, want, od, n, d_in, d_out, sp)
        od_dev = hip.upload(np.full_like(want, 0xAB))
        hip.matmul_2d(xd, ad, wdv, wd, od_dev, od, n, d_in, d_out, sp)
        got = od_dev.download(shape=want.shape)
        assert np.array_equal(got[:sp], want[:sp]), "rows below start_pos must be untouched"
        compare_rows(got[sp:], want[sp:], od, d_out, f"matmul {name}->{od}", atol=8e-6)


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_lm_head_ragged_width(hip, oracle, name, wd, ad):
    """d_out = 32003 is not a multiple of anything (EmbeddingLinear, gten/modules.cpp:70-81)"""
    r = rng(9)
    d_in, d_out = 256, 1003
    x, _ = act_rows(oracle, r, 1, d_in, ad)
    w, _ = weight_rows(oracle, r, d_out, d_in, wd)
    want = np.zeros((1, d_out * 4), np.uint8)
    oracle.matmul_2d(x, ad, w, wd, want, F32, 1, d_in, d_out, 0)
    out = hip.alloc(d_out * 4 + 64)
    out.zero(0x7F)
    hip.matmul_2d(hip.upload(x), ad, up_weight(hip, w, wd, d_out, d_in), wd, out, F32, 1, d_in, d_out, 0)
    got = out.download()
    compare_rows(got[: d_out * 4].reshape(1, -1), want, F32, d_out, "lm_head")
    assert (got[d_out * 4:] == 0x7F).all(), "wrote past the ragged end"


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_token_embed(hip, oracle, name, wd, ad):
    r = rng(5)
    V, d = 50, 256
    w, _ = weight_rows(oracle, r, V, d, wd)
    toks = np.array([3, 49, 0, 7, 7], np.int32)
    want = np.zeros((5, row_bytes(ad, d)), np.uint8)
    oracle.token_embed(w, wd, toks, want, ad, d, 1)
    out = hip.upload(np.zeros_like(want))
    hip.token_embed(up_weight(hip, w, wd, V, d), wd, V, hip.upload(toks), out, ad, 5, d, 1)
    got = out.download(shape=want.shape)
    assert np.array_equal(got, want), "token_embed is a copy / exact requantisation: must be bit-exact"


@pytest.mark.parametrize("ad", [F16, Q8])
@pytest.mark.parametrize("d", [256, 2048, 5632])
def test_rowwise_ops(hip, oracle, ad, d):
    r = rng(d)
    n = 4
    x, _ = act_rows(oracle, r, n, d, ad)
    y, _ = act_rows(oracle, r, n, d, ad)
    w = (1 + 0.05 * r.standard_normal(d)).astype(np.float16)
    xd, yd, wdv = hip.upload(x), hip.upload(y), hip.upload(w)
    for sp in (0, 3):
        want = np.zeros_like(x)
        out = hip.upload(np.zeros_like(x))
        oracle.rms_norm(x, ad, w, want, n, d, sp)
        hip.rms_norm(xd, ad, wdv, out, n, d, sp)
        compare_rows(out.download(shape=x.shape), want, ad, d, "rms_norm")

        want = x.copy()
        buf = hip.upload(x)
        oracle.rotary_emb(want, ad, n, d, 64, sp)
        hip.rotary_emb(buf, ad, n, d, 64, sp)
        compare_rows(buf.download(shape=x.shape), want, ad, d, "rope", min_exact=0.999)

        want = np.zeros_like(x)
        out.zero()
        oracle.silu(x, want, ad, n, d, sp)
        hip.silu(xd, out, ad, n, d, sp)
        compare_rows(out.download(shape=x.shape), want, ad, d, "silu", min_exact=0.995)

        want = x.copy()
        buf = hip.upload(x)
        oracle.silu(want, want, ad, n, d, sp)
        hip.silu(buf, buf, ad, n, d, sp)
        compare_rows(buf.download(shape=x.shape), want, ad, d, "silu_inplace", min_exact=0.995)

        want = np.zeros_like(x)
        out.zero()
        oracle.mul(x, y, want, ad, n, d, sp)
        hip.mul(xd, yd, out, ad, n, d, sp)
        assert np.array_equal(out.download(shape=x.shape), want), "mul must be bit-exact"

        want = x.copy()
        buf = hip.upload(x)
        oracle.mul(want, y, want, ad, n, d, sp)
        hip.mul(buf, yd, buf, ad, n, d, sp)
        assert np.array_equal(buf.download(shape=x.shape), want), "mul_inplace must be bit-exact"

        want = np.zeros_like(x)
        out.zero()
        oracle.add(x, y, want, ad, n, d, sp)
        hip.add(xd, yd, out, ad, n, d, sp)
        assert np.array_equal(out.download(shape=x.shape), want), "add must be bit-exact"


@pytest.mark.parametrize("d", [2048, 5632])
def test_rowwise_ops_many_rows_block_pair_kernel(hip, oracle, d):
    """prompt-sized calls of silu / mul / add on Q8 rows take the block-pair kernel (k_elementwise_q8x2: 17 dwords per
    pair of blocks, everything in registers): the same bytes as the oracle for mul and add, silu inside its tolerance,
    out of place and in place, from a start row"""
    r = rng(7 * d)
    n = 24
    x, _ = act_rows(oracle, r, n, d, Q8)
    y, _ = act_rows(oracle, r, n, d, Q8)
    xd, yd = hip.upload(x), hip.upload(y)
    for sp in (0, 5):
        want = np.zeros_like(x)
        out = hip.upload(np.zeros_like(x))
        oracle.mul(x, y, want, Q8, n, d, sp)
        hip.mul(xd, yd, out, Q8, n, d, sp)
        assert np.array_equal(out.download(shape=x.shape), want), "mul must be bit-exact"
        want = np.zeros_like(x)
        out.zero()
        oracle.add(x, y, want, Q8, n, d, sp)
        hip.add(xd, yd, out, Q8, n, d, sp)
        assert np.array_equal(out.download(shape=x.shape), want), "add must be bit-exact"
        want = x.copy()
        buf = hip.upload(x)
        oracle.add(want, y, want, Q8, n, d, sp)
        hip.add(buf, yd, buf, Q8, n, d, sp)
        assert np.array_equal(buf.download(shape=x.shape), want), "add_inplace must be bit-exact"
        want = x.copy()
        buf = hip.upload(x)
        oracle.silu(want, want, Q8, n, d, sp)
        hip.silu(buf, buf, Q8, n, d, sp)
        compare_rows(buf.download(shape=x.shape), want, Q8, d, "silu_inplace", min_exact=0.995)


@pytest.mark.parametrize("d", [2048, 5632])
def test_rowwise_ops_many_rows_f16_vector_kernel(hip, oracle, d):
    """prompt-sized calls of silu / mul / add on f16 rows take k_elementwise_f16x8 (8 elements per thread): mul and add are
    one rounding of an exact f32 result -- the oracle's bytes; silu inside its tolerance; in place and from a start row"""
    r = rng(9 * d)
    n = 24
    x, _ = act_rows(oracle, r, n, d, F16)
    y, _ = act_rows(oracle, r, n, d, F16)
    xd, yd = hip.upload(x), hip.upload(y)
    for sp in (0, 5):
        for name, ofn, hfn in (("mul", oracle.mul, hip.mul), ("add", oracle.add, hip.add)):
            want = np.zeros_like(x)
            out = hip.upload(np.zeros_like(x))
            ofn(x, y, want, F16, n, d, sp)
            hfn(xd, yd, out, F16, n, d, sp)
            assert np.array_equal(out.download(shape=x.shape), want), name + " must be bit-exact"
        want = x.copy()
        buf = hip.upload(x)
        oracle.mul(want, y, want, F16, n, d, sp)
        hip.mul(buf, yd, buf, F16, n, d, sp)
        assert np.array_equal(buf.download(shape=x.shape), want), "mul_inplace must be bit-exact"
        want = x.copy()
        buf = hip.upload(x)
        oracle.silu(want, want, F16, n, d, sp)
        hip.silu(buf, buf, F16, n, d, sp)
        compare_rows(buf.download(shape=x.shape), want, F16, d, "silu_inplace f16", min_exact=0.995)


def test_rope_last_position(hip, oracle):
    """position 2047: the angle table comes from host libm like the reference's"""
    r = rng(13)
    n, d = 2048, 128
    x, _ = act_rows(oracle, r, n, d, F16)
    want = x.copy()
    oracle.rotary_emb(want, F16, n, d, 64, 2040)
    buf = hip.upload(x)
    hip.rotary_emb(buf, F16, n, d, 64, 2040)
    compare_rows(buf.download(shape=x.shape), want, F16, d, "rope@2047", min_exact=0.999)


@pytest.mark.parametrize("ad", [F16, Q8])
@pytest.mark.parametrize("n,sp,H,G,dh", [(5, 0, 8, 2, 64), (33, 0, 8, 2, 64), (40, 39, 8, 2, 64), (70, 64, 8, 2, 32),
                                         (300, 299, 32, 4, 64), (1, 0, 4, 4, 64)])
def test_qkv_attn(hip, oracle, ad, n, sp, H, G, dh):
    r = rng(n * 7 + sp)
    q, _ = act_rows(oracle, r, n, H * dh, ad)
    k, _ = act_rows(oracle, r, n, G * dh, ad)
    v, _ = act_rows(oracle, r, n, G * dh, ad)
    want = np.zeros((n, row_bytes(ad, H * dh)), np.uint8)
    oracle.qkv_attn(q, k, v, want, ad, n, H, G, dh, sp)
    out = hip.upload(np.zeros_like(want))
    hip.qkv_attn(hip.upload(q), hip.upload(k), hip.upload(v), out, ad, n, H, G, dh, sp)
    got = out.download(shape=want.shape)
    assert not got[:sp].any()
    # two chained roundings (probabilities, then the output row): allow 2 steps.  f16 prompt-sized calls (>= 16 new rows,
    # 64-wide heads) take the tiled kernel, whose f16 MFMA scores differ from the scalar loop by f32 summation order
    # (~1e-7): enough to flip the f16 rounding of a probability that sits on a tie -- for a DOMINANT probability (p ~ 1,
    # ulp 4.9e-4) that moves the whole output row by 4.9e-4 |v|, the spread the reference's own AVX and scalar builds show
    tiled_f16 = ad == F16 and dh == 64 and n - sp >= 16
    compare_rows(got[sp:], want[sp:], ad, H * dh, "qkv_attn", min_exact=0.90, steps=2.0, atol=2e-3 if tiled_f16 else 2e-4)


def test_errors_are_reported_not_swallowed(hip):
    from __graft_entry__ import load_package
    pkg = load_package()
    a = hip.alloc(1024)
    with pytest.raises(pkg.GtenHipError):
        hip.matmul_2d(a, F16, a, Q4, a, F16, 1, 256, 32)       # dtype pair the reference does not dispatch
    with pytest.raises(pkg.GtenHipError):
        hip.add(a, a, a, Q8, 2, 256, start_pos=2)             # no rows to compute


@pytest.mark.parametrize("name,wd,ad", MODES())
@pytest.mark.parametrize("n,d_in,d_out,sp", [(40, 256, 96, 0), (100, 2048, 256, 3), (64, 5632, 128, 0), (17, 256, 64, 1)])
def test_matmul_2d_prefill_on_matrix_cores(hip, oracle, name, wd, ad, n, d_in, d_out, sp):
    """>= 16 new rows take the MFMA kernel (gten_mfma.hip), in two forms for quantized weights.
    EXACT (gten_hip_set_prefill_exact(1)): one MFMA is one exact integer block dot and the blocks are accumulated in
    the reference's scalar-build order, so the outputs equal the oracle in scalar order BIT FOR BIT.
    FAST (default): the block deltas are folded into the f16 operands and the sums accumulate inside the matrix core --
    every operand element carries one fp16 rounding (relative 2^-11): the f32 outputs agree with the oracle to 2e-3 of
    the row's rms, Q8 / f16 outputs within one quantization step.  f16 weights accumulate inside the MFMA (tolerance)."""
    r = rng(n * 131 + d_in + d_out)
    x, _ = act_rows(oracle, r, n, d_in, ad)
    w, _ = weight_rows(oracle, r, d_out, d_in, wd)
    xd, wdv = hip.upload(x), up_weight(hip, w, wd, d_out, d_in)
    for exact_form in ((True, False) if wd != F16 else (True,)):
        hip.set_prefill_exact(exact_form)
        try:
            for od in (ad, F32):
                want = np.full((n, row_bytes(od, d_out)), 0xAB, np.uint8)
                oracle.matmul_2d(x, ad, w, wd, want, od, n, d_in, d_out, sp)
                od_dev = hip.upload(np.full_like(want, 0xAB))
                hip.matmul_2d(xd, ad, wdv, wd, od_dev, od, n, d_in, d_out, sp)
                got = od_dev.download(shape=want.shape)
                assert np.array_equal(got[:sp], want[:sp]), "rows below start_pos must be untouched"
                if exact_form:
                    compare_rows(got[sp:], want[sp:], od, d_out, f"mfma matmul {name}->{od}", atol=8e-6)
                elif od == F32:
                    g, t = got[sp:].view(np.float32), want[sp:].view(np.float32)
                    rms = np.sqrt((t * t).mean(axis=1, keepdims=True))
                    assert (np.abs(g - t) <= 2e-3 * rms + 1e-7).all(), (name, float((np.abs(g - t) / rms).max()))
                else:
                    compare_rows(got[sp:], want[sp:], od, d_out, f"fast mfma matmul {name}->{od}", min_exact=0.80, steps=2.0, atol=2e-4)
                if wd != F16 and exact_form:
                    oracle.set_simd(False)
                    try:
                        exact = np.full_like(want, 0xAB)
                        oracle.matmul_2d(x, ad, w, wd, exact, od, n, d_in, d_out, sp)
                    finally:
                        oracle.set_simd(True)
                    assert np.array_equal(got[sp:], exact[sp:]), f"{name}->{od}: the exact MFMA form must reproduce the scalar-order oracle"
        finally:
            hip.set_prefill_exact(False)


@pytest.mark.parametrize("n,sp,H,G", [(300, 0, 8, 2), (64, 0, 4, 4), (20, 0, 4, 2), (333, 40, 8, 2), (1100, 1000, 4, 1),
                                      (257, 0, 4, 2), (48, 31, 4, 2)])
def test_qkv_attn_tiled_prefill_equals_row_kernel(hip, oracle, monkeypatch, n, sp, H, G):
    """>= 16 new Q8 rows with 64-wide heads take gten_attn_tiled.hip (32 rows of a head per workgroup, int8
    MFMA scores, three passes over the K tiles).  EXACT form (gten_hip_set_prefill_exact(1)): BIT-IDENTICAL to the
    row-at-a-time kernel -- which the fused decoder is tested against -- across ragged tiles, chunked prefill
    (start_pos > 0) and 256-tile edges.  FAST form (default): p.V on the matrix cores with f16 operands -- inside the
    oracle's band like the exact form, and within fp16 rounding noise of it."""
    dh = 64
    r = rng(n * 11 + sp + H)
    q, _ = act_rows(oracle, r, n, H * dh, Q8)
    k, _ = act_rows(oracle, r, n, G * dh, Q8)
    v, _ = act_rows(oracle, r, n, G * dh, Q8)
    qd, kd, vd = hip.upload(q), hip.upload(k), hip.upload(v)
    outs = []
    hip.set_prefill_exact(True)
    try:
        for pieces in (True, False):
            # the row kernel k_attn (calls of fewer than 16 new rows: the same rows in 15-row pieces), then the tiled kernel
            out = hip.upload(np.full((n, row_bytes(Q8, H * dh)), 0xCD, np.uint8))
            if pieces:
                for r0 in range(sp, n, 15):
                    hip.qkv_attn(qd, kd, vd, out, Q8, min(r0 + 15, n), H, G, dh, r0)
            else:
                hip.qkv_attn(qd, kd, vd, out, Q8, n, H, G, dh, sp)
            outs.append(out.download(shape=(n, row_bytes(Q8, H * dh))))
    finally:
        hip.set_prefill_exact(False)
    out = hip.upload(np.full((n, row_bytes(Q8, H * dh)), 0xCD, np.uint8))
    hip.qkv_attn(qd, kd, vd, out, Q8, n, H, G, dh, sp)
    fast = out.download(shape=(n, row_bytes(Q8, H * dh)))
    assert (outs[1][:sp] == 0xCD).all() and (fast[:sp] == 0xCD).all(), "rows before start_pos must not be written"
    assert np.array_equal(outs[0], outs[1])
    want = np.zeros((n, row_bytes(Q8, H * dh)), np.uint8)
    oracle.qkv_attn(q, k, v, want, Q8, n, H, G, dh, sp)
    compare_rows(outs[1][sp:], want[sp:], Q8, H * dh, "qkv_attn_tiled", min_exact=0.90, steps=2.0, atol=2e-4)
    compare_rows(fast[sp:], want[sp:], Q8, H * dh, "qkv_attn_tiled fast", min_exact=0.85, steps=2.0, atol=2e-4)
    compare_rows(fast[sp:], outs[1][sp:], Q8, H * dh, "qkv_attn_tiled fast vs exact", min_exact=0.85, steps=2.0, atol=2e-4)


def test_q8_scale_arithmetic_is_ieee(hip):
    """The quantizer's delta = absmax / 127 and scale = 1 / delta run as 3-instruction Markstein sequences
    (csrc/gten_dev.h); on the device they equal the IEEE division expansion for every binary32 significand of six
    binades (50 million operands each)."""
    bad_div, bad_recip = hip.selftest_q8scale()
    assert bad_div == 0
    assert bad_recip == 0


def test_rms_norm_wave_kernel_is_bit_identical_to_the_row_kernel(hip, oracle):
    """2048-wide Q8 rows, four or more at a time, take the one-wave-per-row kernel: same sum tree, same bytes as the
    workgroup-per-row kernel that single rows take"""
    r = rng(77)
    n, d = 37, 2048
    x, _ = act_rows(oracle, r, n, d, Q8, scale=3.0)
    x[5] = x[6]                                           # a repeated row
    w = (1 + 0.1 * r.standard_normal(d)).astype(np.float16)
    xd, wdv = hip.upload(x), hip.upload(w)
    many = hip.upload(np.zeros_like(x))
    one = hip.upload(np.zeros_like(x))
    hip.rms_norm(xd, Q8, wdv, many, n, d, 0)
    for row in range(n):
        hip.rms_norm(xd, Q8, wdv, one, row + 1, d, row)
    assert np.array_equal(many.download(shape=x.shape), one.download(shape=x.shape))
    want = np.zeros_like(x)
    oracle.rms_norm(x, Q8, w, want, n, d, 0)
    compare_rows(many.download(shape=x.shape), want, Q8, d, "rms_norm wave kernel")


@pytest.mark.parametrize("d", [2048, 256])
def test_rope_wave_kernel_is_bit_identical_to_the_row_kernel(hip, oracle, d):
    """Q8 rows with 64-wide heads, four or more at a time, take the one-wave-per-row kernel (lane = block, the partner half
    of a head from the neighbouring lane): the bytes of the workgroup-per-row kernel that single rows take"""
    r = rng(91 + d)
    n = 41
    x, _ = act_rows(oracle, r, n, d, Q8, scale=2.0)
    many = hip.upload(x)
    one = hip.upload(x)
    hip.rotary_emb(many, Q8, n, d, 64, 3)
    for row in range(3, n):
        hip.rotary_emb(one, Q8, row + 1, d, 64, row)
    got = many.download(shape=x.shape)
    assert np.array_equal(got, one.download(shape=x.shape))
    assert np.array_equal(got[:3], x[:3]), "rows before start_pos must not be touched"
    want = x.copy()
    oracle.rotary_emb(want, Q8, n, d, 64, 3)
    compare_rows(got, want, Q8, d, "rope wave kernel", min_exact=0.999)
