"""Shared test helpers: seeded inputs in the reference's storage formats."""
import numpy as np

from oracle import orc
from oracle.orc import F16, F32, Q4, Q8, row_bytes  # noqa: F401


def rng(seed):
    return np.random.default_rng(seed)


def act_rows(lib, r, n, d, dtype, scale=1.0):
    """Random activation rows in storage layout (uint8 [n][row_bytes]) + their f32 values."""
    x = (r.standard_normal((n, d)) * scale).astype(np.float32)
    b = lib.quantize_rows(x, dtype)
    return b, lib.dequantize_rows(b, dtype, d)


def weight_rows(oracle, r, rows, cols, dtype, scale=0.02):
    w = (r.standard_normal((rows, cols)) * scale).astype(np.float32)
    return oracle.quantize_weight(w, dtype), w


def tiny_config(wdtype, adtype, **kw):
    cfg = dict(n_vocab=512, max_ctx=128, n_embd=256, n_ffn=512, n_layers=2, n_heads=8, n_kv_heads=2,
               wdtype=wdtype, adtype=adtype)
    cfg.update(kw)
    return orc.Config(**cfg)


def weight_shapes(cfg):
    """(rows, cols, dtype) of every tensor in .gten order (tinyllama.cpp:345-391)."""
    E, F, V = cfg.n_embd, cfg.n_ffn, cfg.n_vocab
    KV = (E // cfg.n_heads) * cfg.n_kv_heads
    W = cfg.wdtype
    out = [(V, E, W)]
    for _ in range(cfg.n_layers):
        out += [(E, E, W), (KV, E, W), (KV, E, W), (E, E, W), (F, E, W), (F, E, W), (E, F, W),
                (1, E, F16), (1, E, F16)]
    out += [(1, E, F16), (V, E, W)]
    return out


def random_weights(oracle, cfg, seed):
    """List of uint8 arrays (storage layout) for every tensor of a model with config cfg."""
    r = rng(seed)
    ws = []
    for rows, cols, dt in weight_shapes(cfg):
        if rows == 1 and dt == F16:
            w = (1.0 + 0.05 * r.standard_normal((1, cols))).astype(np.float32)
        else:
            w = (0.02 * r.standard_normal((rows, cols))).astype(np.float32)
        ws.append(oracle.quantize_weight(w, dt))
    return ws


def MODES():
    """(name, wdtype, adtype) as chosen by tinyllama.cpp:258-265."""
    return [("f16", F16, F16), ("q8", Q8, Q8), ("q4", Q4, Q8)]


# ---------------------------------------------------------------- comparators

def q8_fields(b, d):
    """uint8 [n][row_bytes(Q8,d)] -> (deltas f32 [n][nb], quants int8 [n][nb][32]); d % 32 == 0."""
    n = b.shape[0]
    blk = b.reshape(n, d // 32, 34)
    deltas = blk[:, :, :2].copy().view(np.float16).astype(np.float32).reshape(n, d // 32)
    q = blk[:, :, 2:].copy().view(np.int8)
    return deltas, q


def compare_rows(got, want, dtype, d, what="", min_exact=0.97, steps=1.0, atol=0.0):
    """Storage rows from the HIP path vs the oracle.

    Integer/byte work is exact in both; what can differ is the f32 summation
    order (and libm's last ulp), which moves a value by ~1e-7 relative and can
    flip a rounding at a tie.  So: the dequantized values must agree within
    `steps` quantization steps (Q8: delta of the block; f16: one ulp), and at
    least `min_exact` of the stored bytes must be identical.  `atol` covers f32
    accumulation error of long dot products whose result cancels to near zero
    (there the error is absolute, ~K*2^-24*|partial sums|, not relative); for attention
    it also covers a flipped rounding of one PROBABILITY (1 ulp of p ~ 1.5e-5 at p ~ 1/32)
    times |v|, which is absolute in the output however small the output is.
    """
    got = np.ascontiguousarray(got).view(np.uint8)
    want = np.ascontiguousarray(want).view(np.uint8)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if dtype == Q8:
        dg, qg = q8_fields(got, d)
        dw, qw = q8_fields(want, d)
        vg = qg.astype(np.float32) * dg[:, :, None]
        vw = qw.astype(np.float32) * dw[:, :, None]
        tol = steps * np.maximum(dg, dw)[:, :, None] * 1.001 + 1e-30
        err = np.abs(vg - vw)
        assert (err <= tol).all(), (what, float(err.max()), float(tol.min()))
        rel_d = np.abs(dg - dw) <= np.maximum(np.abs(dw), 1e-30) * 2.0 ** -9
        assert rel_d.all(), (what, "block deltas differ by more than one fp16 ulp")
    elif dtype == F16:
        vg = got.view(np.float16).astype(np.float32)
        vw = want.view(np.float16).astype(np.float32)
        ulp = np.maximum(np.abs(vw), 2.0 ** -14) * 2.0 ** -10
        assert (np.abs(vg - vw) <= np.maximum(steps * ulp * 1.001, atol)).all(), (what, float(np.abs(vg - vw).max()))
    else:
        vg = got.view(np.float32)
        vw = want.view(np.float32)
        np.testing.assert_allclose(vg, vw, rtol=2e-5, atol=max(atol, 2e-6), err_msg=what)
        return 1.0
    exact = float((got == want).mean())
    assert exact >= min_exact, (what, f"only {exact:.4f} of the bytes are identical")
    return exact
