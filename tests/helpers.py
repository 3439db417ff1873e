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
