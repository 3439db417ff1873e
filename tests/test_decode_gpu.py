"""-m gpu: the fused single-token decode path (gten_decode.hip) against
(a) the operator-by-operator HIP path -- same rounding points, so identical
    logits while the context fits one attention chunk, and
(b) the oracle, within the stated tolerance (see test_model_gpu.py)."""
import numpy as np
import pytest

from gpu_common import hip  # noqa: F401
from __graft_entry__ import load_package
from helpers import MODES, tiny_config
from test_model_gpu import check_logits, host_cfg

pytestmark = pytest.mark.gpu


def build_models(host, oracle, wd, ad, seed, n_models=2, **kw):
    ocfg = tiny_config(wd, ad, n_heads=4, n_kv_heads=2, **kw)
    cfg = host_cfg(ocfg)
    gms = [host.model(cfg) for _ in range(n_models)]
    om = oracle.model(ocfg)
    for i in range(gms[0].n_weights()):
        w = host.synth_weight(cfg, seed, i)
        for g in gms:
            g.set_weight(i, w)
        om.set_weight(i, w)
    return cfg, gms, om


@pytest.mark.oracle_parity
@pytest.mark.parametrize("name,wd,ad", MODES())
def test_fused_decode_against_oracle(hip, oracle, name, wd, ad):
    """the fused path (and the operator path beside it) inside the band around the ORACLE, step by step; asserted on its
    own, before and independently of the bit-identity property below"""
    pkg = load_package()
    host = pkg.load_host()
    cfg, (fast, slow), om = build_models(host, oracle, wd, ad, seed=2468)
    slow.set_fast_decode(False)
    toks = list(host.synthetic_tokens(7, seed=5, n_vocab=cfg.n_vocab))
    for step in range(12):
        sp = 0 if step == 0 else len(toks) - 1
        a = fast.logits(toks, sp)          # step 0: operator prefill, later steps: fused path
        b = slow.logits(toks, sp)
        want = om.logits(toks, sp)
        check_logits(name, a, want, float(want.std()))
        check_logits(name, b, want, float(want.std()))
        toks.append(int(np.argmax(want)))
    fast.close(); slow.close(); om.close()


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_fused_decode_equals_operator_path(hip, name, wd, ad):
    """property (HIP against HIP): same rounding points, so identical logits while the context fits one attention chunk"""
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(wd, ad, n_heads=4, n_kv_heads=2))
    fast, slow = host.model(cfg), host.model(cfg)
    for i in range(fast.n_weights()):
        w = host.synth_weight(cfg, 2468, i)
        fast.set_weight(i, w); slow.set_weight(i, w)
    slow.set_fast_decode(False)
    toks = list(host.synthetic_tokens(7, seed=5, n_vocab=cfg.n_vocab))
    for step in range(12):
        sp = 0 if step == 0 else len(toks) - 1
        a = fast.logits(toks, sp)
        b = slow.logits(toks, sp)
        assert np.array_equal(a, b), (name, step, float(np.abs(a - b).max()))
        toks.append(int(np.argmax(a)))
    fast.close(); slow.close()


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_fused_decode_long_context_and_graph_replay(hip, oracle, name, wd, ad):
    """cross the 256-position attention chunk boundary (ragged last chunk, partial
    probability block) with the throughput API: device-resident ids, graph replay"""
    pkg = load_package()
    host = pkg.load_host()
    cfg, (g_graph, g_eager), om = build_models(host, oracle, wd, ad, seed=1357, max_ctx=320, n_layers=1)
    N = 290
    toks = host.synthetic_tokens(N, seed=9, n_vocab=cfg.n_vocab)
    g_graph.decode_begin(toks)
    g_eager.decode_begin(toks)
    res_g, res_e = [], []
    for n in range(1, N + 1):
        g_graph.decode_step(n, True)
        g_eager.decode_step(n, False)
    for n in (1, 2, 33, 255, 256, 257, 288, N):
        res_g.append(g_graph.decode_result(n))
        res_e.append(g_eager.decode_result(n))
    assert res_g == res_e, "graph replay must equal eager launches"
    # oracle: same teacher-forced ids, step by step (1 layer, small: seconds)
    want_tok = {}
    for n in range(1, N + 1):
        lg = om.logits(toks[:n], n - 1)
        if n in (1, 2, 33, 255, 256, 257, 288, N):
            want_tok[n] = (int(np.argmax(lg)), lg)
    # logits of the LAST step are still in the model's logits buffer
    got_last = g_graph.logits(toks[:N], N - 1)      # recomputes row N-1 through the fused path: same bytes
    check_logits(name, got_last, want_tok[N][1], float(want_tok[N][1].std()))
    agree = sum(int(res_g[i] == want_tok[n][0]) for i, n in enumerate((1, 2, 33, 255, 256, 257, 288, N)))
    if name == "f16":
        assert agree == 8, (res_g, [want_tok[n][0] for n in want_tok])
    else:
        assert agree >= 6
    g_graph.close(); g_eager.close(); om.close()


def test_kv_cache_rows_identical_between_paths(hip, oracle):
    """the fused path appends the same K/V bytes the operators would"""
    from helpers import Q4, Q8
    pkg = load_package()
    host = pkg.load_host()
    cfg, (fast, slow), om = build_models(host, oracle, Q4, Q8, seed=99, n_layers=2)
    slow.set_fast_decode(False)
    toks = list(host.synthetic_tokens(5, seed=2, n_vocab=cfg.n_vocab))
    for step in range(6):
        sp = 0 if step == 0 else len(toks) - 1
        a = fast.logits(toks, sp)
        b = slow.logits(toks, sp)
        toks.append(int(np.argmax(a)))
        assert np.array_equal(a, b)
    # a further step on each consumes the caches written so far: equality of its
    # logits implies equality of every cached row that matters
    a = fast.logits(toks, len(toks) - 1)
    b = slow.logits(toks, len(toks) - 1)
    assert np.array_equal(a, b)
    fast.close(); slow.close(); om.close()


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_fused_decode_d_head_32(hip, oracle, name, wd, ad):
    """d_head = 32 takes the generic attention kernels of the fused path (one Q8 block per head,
    2-byte aligned head slices): still bit-identical to the operator path, still inside the band"""
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(wd, ad, n_heads=8, n_kv_heads=2)          # n_embd 256 / 8 heads
    cfg = host_cfg(ocfg)
    fast, slow, om = host.model(cfg), host.model(cfg), oracle.model(ocfg)
    slow.set_fast_decode(False)
    for i in range(fast.n_weights()):
        w = host.synth_weight(cfg, 321, i)
        fast.set_weight(i, w); slow.set_weight(i, w); om.set_weight(i, w)
    toks = list(host.synthetic_tokens(5, seed=4, n_vocab=cfg.n_vocab))
    for step in range(8):
        sp = 0 if step == 0 else len(toks) - 1
        a, b, want = fast.logits(toks, sp), slow.logits(toks, sp), om.logits(toks, sp)
        assert np.array_equal(a, b), (name, step)
        check_logits(name, a, want, float(want.std()))
        toks.append(int(np.argmax(want)))
    fast.close(); slow.close(); om.close()


def test_decode_at_the_last_position(hip, oracle):
    """n = max_ctx: the step that fills the cache to its last row (RoPE table's last entry, ragged
    last attention chunk when max_ctx is not a multiple of 256)"""
    from helpers import Q4, Q8
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(Q4, Q8, n_heads=4, n_kv_heads=2, max_ctx=40, n_layers=1)
    cfg = host_cfg(ocfg)
    gm, om = host.model(cfg), oracle.model(ocfg)
    for i in range(gm.n_weights()):
        w = host.synth_weight(cfg, 5, i)
        gm.set_weight(i, w); om.set_weight(i, w)
    toks = host.synthetic_tokens(40, seed=8, n_vocab=cfg.n_vocab)
    for n in range(1, 41):
        got = gm.logits(toks[:n], n - 1)
        want = om.logits(toks[:n], n - 1)
    check_logits("q4", got, want, float(want.std()))
    gm.close(); om.close()


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_one_pass_attention_against_the_two_launch_form(hip, oracle, name, wd, ad):
    """default: scores, chunk-local softmax and p.V in ONE launch per block (k_dec_attn_one64), the chunks joined with
    their weights in the o projection's prologue.  gten_hip_set_decode_exact(1): two launches, probabilities rounded
    against the statistics of the whole row (the reference's rounding point, gten/ops.h:972-997).  While the context
    fits one chunk (n <= 256) both are the same bytes; beyond, the logits stay inside the model band (a probability
    block is rounded against its chunk's scale: fp16 rounding of the block delta) and BOTH sit inside the band
    around the oracle."""
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(wd, ad, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2)
    cfg = host_cfg(ocfg)
    N = 300
    toks = host.synthetic_tokens(N, seed=11, n_vocab=cfg.n_vocab)
    weights = [host.synth_weight(cfg, 77, i) for i in range(len(cfg.weight_shapes()))]
    watch = (1, 2, 100, 255, 256, 257, 290, N)
    outs = []
    for two in (False, True):
        hip.set_decode_exact(two)                   # (read when the decoder is created)
        gm = host.model(cfg)
        for i, w in enumerate(weights):
            gm.set_weight(i, w)
        gm.decode_begin(toks)                       # the decoder (and its launch choice) is made here
        got = {}
        for n in range(1, N + 1):
            gm.decode_step(n, n % 2 == 0)           # alternate graph replay and eager launches
            if n in watch:
                got[n] = (gm.decode_result(n), gm.logits(toks[:n], n - 1).copy())
        outs.append(got)
        gm.close()
    hip.set_decode_exact(False)
    om = oracle.model(ocfg)
    for i, w in enumerate(weights):
        om.set_weight(i, w)
    for n in range(1, N + 1):
        want = om.logits(toks[:n], n - 1)
        if n not in watch:
            continue
        (ida, la), (idb, lb) = outs[0][n], outs[1][n]
        if n <= 256:
            assert ida == idb and np.array_equal(la, lb), (name, n)
        else:
            check_logits(name, la, lb, float(lb.std()))
        check_logits(name, la, want, float(want.std()))
        check_logits(name, lb, want, float(want.std()))
    om.close()


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_steps_call_equals_single_steps(hip, name, wd, ad):
    """gten_hip_decoder_steps (count consecutive steps, four per hipGraph replay, the position advanced on the device) leaves
    the same argmax ids, logits and K/V rows as the same steps one call at a time; counts that are not multiples of four,
    a start in the middle, eager mode, and the multi-sequence decoder"""
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(wd, ad, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    weights = [host.synth_weight(cfg, 404, i) for i in range(len(cfg.weight_shapes()))]
    toks = host.synthetic_tokens(300, seed=21, n_vocab=cfg.n_vocab)
    a, b = host.model(cfg), host.model(cfg)
    for i, w in enumerate(weights):
        a.set_weight(i, w); b.set_weight(i, w)
    a.decode_begin(toks); b.decode_begin(toks)
    for n in range(1, 301):
        a.decode_step(n, True)
    b.decode_steps(1, 7, True)           # 4 + 3
    b.decode_steps(8, 1, True)
    b.decode_steps(9, 250, True)
    b.decode_steps(259, 10, False)       # eager
    b.decode_steps(269, 32, True)
    for n in (1, 7, 8, 9, 12, 255, 256, 257, 258, 268, 269, 300):
        assert a.decode_result(n) == b.decode_result(n), (name, n)
    assert np.array_equal(a.logits(toks[:300], 299), b.logits(toks[:300], 299))
    a.close(); b.close()
    ba, bb = host.batch(cfg, 4), host.batch(cfg, 4)
    for i, w in enumerate(weights):
        ba.set_weight(i, w); bb.set_weight(i, w)
    streams = [host.synthetic_tokens(60, seed=60 + q, n_vocab=cfg.n_vocab) for q in range(4)]
    for q in range(4):
        ba.decode_begin(q, streams[q]); bb.decode_begin(q, streams[q])
    for n in range(1, 61):
        ba.decode_step(n, True)
    bb.decode_steps(1, 60, True)
    for q in range(4):
        assert ba.decode_result(q, 60) == bb.decode_result(q, 60) and np.array_equal(ba.logits(q), bb.logits(q))
    ba.close(); bb.close()
