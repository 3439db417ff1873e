"""-m gpu: gten_hip_block_rows (one AttentionBlock over many new rows as one composed call) against the module-by-module
operator sequence of gten/modules.cpp:177-254 on the same buffers -- EVERY buffer byte for byte -- and, through the host
model, against the reference-pinned prompt fixtures (tests/test_prefill_gpu.py runs with the composed call by default).
"""
import numpy as np
import pytest

from gpu_common import hip  # noqa: F401
from helpers import F16, Q4, Q8, act_rows, rng, row_bytes, weight_rows

pytestmark = pytest.mark.gpu

ACTS = ("attn_norm_out", "q", "k", "v", "attn_out", "o", "h", "ffn_norm_out", "gate", "up", "down", "out")


def make_block(hip, oracle, r, wd, E, H, KVH, F, max_ctx):
    dh = E // H
    KV = dh * KVH
    w = {}
    for name, rows, cols in (("wq", E, E), ("wk", KV, E), ("wv", KV, E), ("wo", E, E), ("wgate", F, E), ("wup", F, E), ("wdown", E, F)):
        blocks, _ = weight_rows(oracle, r, rows, cols, wd, scale=0.05)
        w[name] = hip.upload_weight(blocks, wd, rows, cols)
    for name in ("attn_norm_w", "ffn_norm_w"):
        w[name] = hip.upload((1.0 + 0.05 * r.standard_normal(E)).astype(np.float16))
    widths = dict(attn_norm_out=E, q=E, k=KV, v=KV, attn_out=E, o=E, h=E, ffn_norm_out=E, gate=F, up=F, down=E, out=E)
    return w, widths


def alloc_acts(hip, widths, max_ctx, fill, ad=Q8):
    bufs = {}
    for k, d in widths.items():
        bufs[k] = hip.alloc(max_ctx * row_bytes(ad, d))
        bufs[k].zero(fill)
    return bufs


def modules_sequence(hip, w, a, inp, wd, n, s, E, H, KVH, F, Q8=Q8):
    """gten/modules.cpp:198-253, operator by operator (Q8 here = the activation dtype of the run)"""
    dh = E // H
    KV = dh * KVH
    hip.rms_norm(inp, Q8, w["attn_norm_w"], a["attn_norm_out"], n, E, s)
    hip.matmul_2d(a["attn_norm_out"], Q8, w["wq"], wd, a["q"], Q8, n, E, E, s)
    hip.matmul_2d(a["attn_norm_out"], Q8, w["wk"], wd, a["k"], Q8, n, E, KV, s)
    hip.rotary_emb(a["q"], Q8, n, E, dh, s)
    hip.rotary_emb(a["k"], Q8, n, KV, dh, s)
    hip.matmul_2d(a["attn_norm_out"], Q8, w["wv"], wd, a["v"], Q8, n, E, KV, s)
    hip.qkv_attn(a["q"], a["k"], a["v"], a["attn_out"], Q8, n, H, KVH, dh, s)
    hip.matmul_2d(a["attn_out"], Q8, w["wo"], wd, a["o"], Q8, n, E, E, s)
    hip.add(inp, a["o"], a["h"], Q8, n, E, s)
    hip.rms_norm(a["h"], Q8, w["ffn_norm_w"], a["ffn_norm_out"], n, E, s)
    hip.matmul_2d(a["ffn_norm_out"], Q8, w["wgate"], wd, a["gate"], Q8, n, E, F, s)
    hip.matmul_2d(a["ffn_norm_out"], Q8, w["wup"], wd, a["up"], Q8, n, E, F, s)
    hip.silu(a["gate"], a["gate"], Q8, n, F, s)
    hip.mul(a["gate"], a["up"], a["gate"], Q8, n, F, s)
    hip.matmul_2d(a["gate"], Q8, w["wdown"], wd, a["down"], Q8, n, F, E, s)
    hip.add(a["h"], a["down"], a["out"], Q8, n, E, s)


CASES = [
    # E, heads, kv heads, F, max_ctx, [(n, start_pos), ...] -- consecutive calls on the same caches
    (256, 4, 2, 512, 160, [(16, 0)]),
    (256, 4, 2, 512, 160, [(37, 0), (100, 37), (160, 100)]),
    (256, 4, 1, 640, 96, [(96, 0)]),
    (2048, 32, 4, 5632, 96, [(96, 0)]),          # the model's block
]


@pytest.mark.parametrize("wd", [Q8, Q4, F16])
@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_block_rows_equals_module_sequence(hip, oracle, wd, exact, case):
    E, H, KVH, F, max_ctx, calls = CASES[case]
    if E == 2048 and exact and wd == Q8:
        pytest.skip("one full-size exact case is enough")
    ad = F16 if wd == F16 else Q8
    r = rng(100 + case)
    w, widths = make_block(hip, oracle, r, wd, E, H, KVH, F, max_ctx)
    a_ref = alloc_acts(hip, widths, max_ctx, 0, ad)
    a_got = alloc_acts(hip, widths, max_ctx, 0, ad)
    xb, _ = act_rows(oracle, r, max_ctx, E, ad, scale=1.0)
    inp = hip.upload(xb)
    hip.set_prefill_exact(exact)
    try:
        for n, s in calls:
            hip.set_block_rows(False)
            modules_sequence(hip, w, a_ref, inp, wd, n, s, E, H, KVH, F, Q8=ad)
            hip.set_block_rows(True)
            ints = dict(adtype=ad, wdtype=wd, n_embd=E, n_heads=H, n_kv_heads=KVH, n_ffn=F)
            bufs = dict(w)
            bufs.update(a_got)
            bufs["inp"] = inp
            assert hip.block_rows(n, s, ints, bufs), "the composed call refused a configuration it is meant for"
            hip.sync()
            for k in ACTS:
                nb = n * row_bytes(ad, widths[k])
                want = a_ref[k].download(nbytes=nb)
                got = a_got[k].download(nbytes=nb)
                assert np.array_equal(got, want), (k, n, s, int((got != want).sum()), "bytes differ")
    finally:
        hip.set_prefill_exact(False)
        hip.set_block_rows(True)


def test_block_rows_declines_what_it_does_not_compute(hip, oracle):
    r = rng(7)
    E, H, KVH, F, max_ctx = 256, 4, 2, 512, 64
    w, widths = make_block(hip, oracle, r, Q8, E, H, KVH, F, max_ctx)
    a = alloc_acts(hip, widths, max_ctx, 0)
    xb, _ = act_rows(oracle, r, max_ctx, E, Q8)
    inp = hip.upload(xb)
    bufs = dict(w)
    bufs.update(a)
    bufs["inp"] = inp
    ints = dict(adtype=Q8, wdtype=Q8, n_embd=E, n_heads=H, n_kv_heads=KVH, n_ffn=F)
    assert not hip.block_rows(8, 0, ints, bufs)                      # fewer than 16 new rows: the row kernels' business
    assert not hip.block_rows(64, 0, dict(ints, adtype=F16, wdtype=Q8), bufs)  # a pair the reference does not dispatch
    assert not hip.block_rows(64, 0, dict(ints, n_heads=8), bufs)    # d_head 32
    hip.set_block_rows(False)
    try:
        assert not hip.block_rows(64, 0, ints, bufs)
    finally:
        hip.set_block_rows(True)
    assert hip.block_rows(64, 0, ints, bufs)
    hip.sync()
