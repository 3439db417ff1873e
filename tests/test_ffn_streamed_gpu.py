"""gate | up and the lm_head of a full 128-row q4 lane as the streamed kernel (csrc/gten_decode_ffn.h, round 5) against
k_dec_mmvh<Q4, 8, 4, ..>: the same f16 operands, the same eight K slices accumulated in the matrix core from zero and added in the
same order -- so not a band but THE SAME BITS, for every sequence of one lane (128) and of two (256).  q8 weights: gate | up + the
silu . mul chain as ONE streamed launch (k_dec_ffn_q8) against the k_dec_mmvh<Q8, 8, 2, false> + k_dec_silumul_rows pair (two K planes
of eight wave slices): the same bits again.  64 sequences (four row tiles): the same kernels with waves 4 .. 7 expanding weights only,
against k_dec_mmvh<.., 4, ..>.  f16 weights and activations (lanes of 64 rows): k_dec_ffn_f16 against k_dec_mmv_f16 + k_dec_silumul_rows_f16."""
import numpy as np
import pytest

from gpu_common import hip  # noqa: F401
from __graft_entry__ import load_package
from helpers import F16, Q4, Q8, tiny_config
from test_model_gpu import host_cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wd", [Q4, Q8, F16])
@pytest.mark.parametrize("n_seq", [64, 128, 256])
def test_streamed_gate_up_equals_the_slab_kernel_bit_for_bit(hip, n_seq, wd):
    pkg = load_package()
    host = pkg.load_host()
    # the kernel is selected at K = 2048 (TinyLlama's width) -- for the lm_head from 16 384 columns up (a ragged last tile here);
    # a narrow FFN and two blocks keep the model small
    cfg = host_cfg(tiny_config(wd, F16 if wd == F16 else Q8, n_embd=2048, n_heads=32, n_kv_heads=4, n_ffn=768, n_layers=2, n_vocab=16403, max_ctx=64))
    weights = [host.synth_weight(cfg, 31, i) for i in range(len(cfg.weight_shapes()))]
    streams = [host.synthetic_tokens(40, seed=7000 + q, n_vocab=cfg.n_vocab) for q in range(n_seq)]
    out = []
    for on in (True, False):
        hip.set_ffn_streamed(on)
        try:
            b = host.batch(cfg, n_seq)
            for i, w in enumerate(weights):
                b.set_weight(i, w)
            for q in range(n_seq):
                b.decode_begin(q, streams[q])
            snap = []
            for n in range(1, 21):
                b.decode_step(n, n % 2 == 0)                 # graph replays and eager launches alike
                if n in (1, 2, 7, 20):
                    snap.append([b.logits(q).copy() for q in range(n_seq)])
            b.decode_step_ragged(np.full(n_seq, 21, np.int32), True)       # (the ragged entry point's graph)
            snap.append([b.logits(q).copy() for q in range(n_seq)])
            out.append(snap)
            b.close()
        finally:
            hip.set_ffn_streamed(True)
    for sa, sb in zip(out[0], out[1]):
        for q, (a, b_) in enumerate(zip(sa, sb)):
            assert np.isfinite(a).all()
            assert np.array_equal(a, b_), (n_seq, q, float(np.abs(a - b_).max()))
    assert not np.array_equal(out[0][-1][0], out[0][-1][1])          # independent sequences


@pytest.mark.parametrize("wd", [Q4, Q8, F16])
def test_a_lane_of_128_rows_equals_64_sequence_decoders_at_full_width(hip, wd):
    """at TinyLlama's width every configuration runs lanes of 128 rows (f16 since round 5: k_dec_mmv_f16<8>, k_dec_ffn_f16 on eight waves):
    per sequence the logits and ids of a 64-sequence decoder holding the same sequences -- the streamed kernels, the eight-tile and the
    four-tile launches all form a row's sums in the same order"""
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(wd, F16 if wd == F16 else Q8, n_embd=2048, n_heads=32, n_kv_heads=4, n_ffn=768, n_layers=2, n_vocab=16403, max_ctx=320))
    weights = [host.synth_weight(cfg, 57, i) for i in range(len(cfg.weight_shapes()))]
    N = 262
    streams = [host.synthetic_tokens(N + 2, seed=4200 + q, n_vocab=cfg.n_vocab) for q in range(128)]
    big = host.batch(cfg, 128)
    for i, w in enumerate(weights):
        big.set_weight(i, w)
    for q in range(128):
        big.decode_begin(q, streams[q])
    big.decode_steps(1, N, True)
    got = [(big.decode_result(q, N), big.logits(q).copy()) for q in range(128)]
    big.close()
    for half in range(2):
        small = host.batch(cfg, 64)
        for i, w in enumerate(weights):
            small.set_weight(i, w)
        for q in range(64):
            small.decode_begin(q, streams[64 * half + q])
        small.decode_steps(1, N, True)
        for q in range(64):
            ids, lg = got[64 * half + q]
            assert small.decode_result(q, N) == ids and np.array_equal(small.logits(q), lg), (half, q)
        small.close()
