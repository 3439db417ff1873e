"""-m gpu: multi-sequence decode (several sequences share every weight pass) against the
single-sequence decoder: per sequence the logits must be BIT-IDENTICAL (same arithmetic in the same
order; only the weights are streamed once instead of once per sequence)."""
import numpy as np
import pytest

from gpu_common import hip  # noqa: F401
from __graft_entry__ import load_package
from helpers import MODES, tiny_config
from test_model_gpu import host_cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,wd,ad", MODES())
@pytest.mark.parametrize("n_seq", [2, 4, 8])
def test_each_sequence_matches_its_own_single_sequence_decode(hip, name, wd, ad, n_seq):
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(wd, ad, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    batch = host.batch(cfg, n_seq)
    singles = [host.model(cfg) for _ in range(n_seq)]
    for i in range(len(cfg.weight_shapes())):
        w = host.synth_weight(cfg, 555, i)
        batch.set_weight(i, w)
        for m in singles:
            m.set_weight(i, w)
    P, N = 7, 270                                   # crosses the 256-position attention chunk boundary
    streams = [host.synthetic_tokens(N, seed=100 + q, n_vocab=cfg.n_vocab) for q in range(n_seq)]
    for q in range(n_seq):
        a = batch.prefill(q, streams[q][:P])        # operator path, sequence q's own caches
        b = singles[q].logits(streams[q][:P], 0)
        assert np.array_equal(a, b), (name, q, "prefill")
        batch.decode_begin(q, streams[q])
        singles[q].decode_begin(streams[q])
    checks = (P + 1, P + 2, 40, 255, 256, 257, N)
    for n in range(P + 1, N + 1):
        batch.decode_step(n, True)
        for m in singles:
            m.decode_step(n, True)
        if n in checks:
            for q in range(n_seq):
                assert batch.decode_result(q, n) == singles[q].decode_result(n), (name, n_seq, q, n)
                got = batch.logits(q)
                want = singles[q].logits(streams[q][:n], n - 1)      # recomputes row n-1 (same bytes) and copies logits
                assert np.array_equal(got, want), (name, n_seq, q, n, float(np.abs(got - want).max()))
    # sequences really are independent: different streams gave different logits
    assert not np.array_equal(batch.logits(0), batch.logits(1))
    batch.close()
    for m in singles:
        m.close()


def test_multiseq_graph_replay_equals_eager(hip):
    from helpers import Q4, Q8
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(Q4, Q8, n_heads=4, n_kv_heads=2, max_ctx=64, n_layers=2))
    res = []
    for use_graph in (True, False):
        b = host.batch(cfg, 4)
        for i in range(len(cfg.weight_shapes())):
            b.set_weight(i, host.synth_weight(cfg, 9, i))
        for q in range(4):
            b.decode_begin(q, host.synthetic_tokens(40, seed=q + 1, n_vocab=cfg.n_vocab))
        for n in range(1, 41):
            b.decode_step(n, use_graph)
        res.append([[b.decode_result(q, n) for n in (1, 2, 17, 40)] for q in range(4)])
        b.close()
    assert res[0] == res[1]


@pytest.mark.parametrize("name,wd,ad", MODES())
@pytest.mark.parametrize("n_seq", [16, 32])
def test_wide_batch_on_matrix_cores_tracks_single_sequence_decode(hip, name, wd, ad, n_seq):
    """n_seq >= 16: every W.x of the step is a skinny matrix product on the matrix cores (rows = sequences).  The
    linears follow the MFMA kernel's block order instead of the GEMV wave tree, so per sequence the logits are
    compared with the single-sequence decoder inside the model band (test_model_gpu.check_logits), the greedy ids
    must agree wherever the top-2 margin is clear, and graph replay must equal eager launches bit for bit."""
    from test_model_gpu import check_logits
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(wd, ad, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    N = 262                                          # crosses the 256-position attention chunk boundary
    streams = [host.synthetic_tokens(N, seed=300 + q, n_vocab=cfg.n_vocab) for q in range(n_seq)]
    runs = []
    for use_graph in (True, False):
        batch = host.batch(cfg, n_seq)
        for i in range(len(cfg.weight_shapes())):
            batch.set_weight(i, host.synth_weight(cfg, 777, i))
        for q in range(n_seq):
            batch.decode_begin(q, streams[q])
        snap = {}
        for n in range(1, N + 1):
            batch.decode_step(n, use_graph)
            if n in (1, 2, 33, 256, 257, N):
                snap[n] = [(batch.decode_result(q, n), batch.logits(q).copy()) for q in (0, 1, n_seq // 2, n_seq - 1)]
        runs.append(snap)
        batch.close()
    for n in runs[0]:
        for (ra, la), (rb, lb) in zip(runs[0][n], runs[1][n]):
            assert ra == rb and np.array_equal(la, lb), (name, n_seq, n, "graph != eager")
    # single-sequence decoders for a few of the sequences
    for k, q in enumerate((0, 1, n_seq // 2, n_seq - 1)):
        m = host.model(cfg)
        for i in range(len(cfg.weight_shapes())):
            m.set_weight(i, host.synth_weight(cfg, 777, i))
        m.decode_begin(streams[q])
        for n in range(1, N + 1):
            m.decode_step(n, True)
            if n in runs[0]:
                want = m.logits(streams[q][:n], n - 1)
                got_id, got = runs[0][n][k]
                check_logits(name, got, want, float(want.std()))
                top2 = np.sort(want)[-2:]
                if top2[1] - top2[0] > 0.05 * float(want.std()):
                    assert got_id == int(np.argmax(want)), (name, n_seq, q, n)
        m.close()
    assert not np.array_equal(runs[0][N][0][1], runs[0][N][1][1])       # independent sequences


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_batch_generation_on_device_equals_single_sequence_generation(hip, name, wd, ad):
    """four sequences with prompts of different lengths generate together (sampler on the device, each sequence at its own
    position, finished sequences parked): every sequence's ids are those of generating it alone -- including one that
    stops early at eos and one that fills the context"""
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(wd, ad, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    prompts = [list(host.synthetic_tokens(n, seed=70 + n, n_vocab=cfg.n_vocab)) for n in (5, 40, 9, 17)]
    total = 300
    m = host.model(cfg)
    for i in range(len(cfg.weight_shapes())):
        m.set_weight(i, host.synth_weight(cfg, 99, i))
    alone = [m.generate(p, total) for p in prompts]
    eos = int(alone[2][60])                                     # sequence 2 will stop here (or earlier); the others may too
    want = [m.generate(p, total, eos) for p in prompts]
    m.close()
    b = host.batch(cfg, 4)
    for i in range(len(cfg.weight_shapes())):
        b.set_weight(i, host.synth_weight(cfg, 99, i))
    got = b.generate(prompts, total, eos)
    b.close()
    assert any(len(w) < total for w in want) and any(len(w) > 100 for w in want)
    for q in range(4):
        assert got[q].tolist() == want[q].tolist(), (name, q, len(got[q]), len(want[q]))


@pytest.mark.parametrize("heads,kv", [(8, 1), (4, 2), (4, 4)])
def test_wide_exact_forms_track_the_fast_forms(hip, heads, kv):
    """gten_hip_set_decode_exact(1): the wide step with row-global rounding points of the probabilities (the grouped VALU
    pair k_dec_attn_score_g / _pv_g with exact p.V terms) and exact integer block sums in the W.x (k_dec_mmv), for 8, 2 and 1
    query heads per kv head, across the attention chunk boundary.  Both forms inside the model band of each other, equal
    greedy ids wherever the margin is clear; each form is repeatable bit for bit."""
    from helpers import Q4, Q8
    from test_model_gpu import check_logits
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(Q4, Q8, n_embd=64 * heads, n_ffn=512, n_heads=heads, n_kv_heads=kv, max_ctx=320, n_layers=2))
    n_seq, N = 16, 270
    streams = [host.synthetic_tokens(N, seed=40 + q, n_vocab=cfg.n_vocab) for q in range(n_seq)]
    runs = []
    try:
        for exact in (True, True, False):
            hip.set_decode_exact(exact)                 # (read when the decoder is created)
            batch = host.batch(cfg, n_seq)
            for i in range(len(cfg.weight_shapes())):
                batch.set_weight(i, host.synth_weight(cfg, 12, i))
            for q in range(n_seq):
                batch.decode_begin(q, streams[q])
            snaps = []
            for n in range(1, N + 1):
                batch.decode_step(n, n % 3 != 0)
                if n in (1, 5, 255, 256, 257, N):
                    snaps.append([(batch.decode_result(q, n), batch.logits(q).copy()) for q in range(n_seq)])
            runs.append(snaps)
            batch.close()
    finally:
        hip.set_decode_exact(False)
    for sa, sb, sc in zip(*runs):
        for (ra, la), (rb, lb), (rc, lc) in zip(sa, sb, sc):
            assert ra == rb and np.array_equal(la, lb)              # exact form: repeatable
            check_logits("q4", lc, la, float(la.std()))
            top2 = np.sort(la)[-2:]
            if top2[1] - top2[0] > 0.05 * float(la.std()):
                assert ra == rc


@pytest.mark.parametrize("n_seq", [4, 16])
def test_sequences_at_different_positions(hip, n_seq):
    """continuous batching: sequence q joins `3 q` steps late, so at any step the sequences sit at different context
    lengths (some on either side of the 256-position attention chunk boundary).  S = 4 (GEMV kernels): every
    sequence bit-identical to its single-sequence decode; S = 16 (matrix cores): inside the model band, greedy ids
    equal where the top-2 margin is clear."""
    from helpers import Q4, Q8
    from test_model_gpu import check_logits
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(Q4, Q8, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    N = 262
    delay = [(3 * q) % 40 for q in range(n_seq)]
    streams = [host.synthetic_tokens(N, seed=700 + q, n_vocab=cfg.n_vocab) for q in range(n_seq)]
    batch = host.batch(cfg, n_seq)
    for i in range(len(cfg.weight_shapes())):
        batch.set_weight(i, host.synth_weight(cfg, 4242, i))
    for q in range(n_seq):
        batch.decode_begin(q, streams[q])
    watch = (0, 1, n_seq // 2, n_seq - 1)
    got = {q: {} for q in watch}
    for t in range(1, N + max(delay) + 1):
        ns = [min(N, max(1, t - delay[q])) for q in range(n_seq)]      # waiting / finished slots repeat a valid step
        batch.decode_step_ragged(ns, use_graph=(t % 2 == 0))
        for q in watch:
            n = ns[q]
            if n in (1, 2, 40, 255, 256, 257, N) and n not in got[q]:
                got[q][n] = (batch.decode_result(q, n), batch.logits(q).copy())
    batch.close()
    for q in watch:
        m = host.model(cfg)
        for i in range(len(cfg.weight_shapes())):
            m.set_weight(i, host.synth_weight(cfg, 4242, i))
        m.decode_begin(streams[q])
        for n in range(1, N + 1):
            m.decode_step(n, True)
            if n in got[q]:
                want = m.logits(streams[q][:n], n - 1)
                gid, glog = got[q][n]
                if n_seq <= 8:
                    assert gid == m.decode_result(n) and np.array_equal(glog, want), (n_seq, q, n)
                else:
                    check_logits("q4", glog, want, float(want.std()))
                    top2 = np.sort(want)[-2:]
                    if top2[1] - top2[0] > 0.05 * float(want.std()):
                        assert gid == int(np.argmax(want)), (n_seq, q, n)
        assert len(got[q]) == 7
        m.close()


def test_multiseq_argument_errors_are_reported(hip):
    """bad batch sizes / positions fail loudly through the C-ABI instead of launching"""
    from helpers import Q4, Q8
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(Q4, Q8, n_heads=4, n_kv_heads=2, max_ctx=64, n_layers=1))
    for bad in (1, 3, 12, 24, 80):
        with pytest.raises(pkg.GtenHipError):
            host.batch(cfg, bad)
    b = host.batch(cfg, 16)
    for i in range(len(cfg.weight_shapes())):
        b.set_weight(i, host.synth_weight(cfg, 1, i))
    for q in range(16):
        b.decode_begin(q, host.synthetic_tokens(8, seed=q, n_vocab=cfg.n_vocab))
    with pytest.raises(pkg.GtenHipError):
        b.decode_step_ragged([1] * 15 + [65])            # past max_ctx
    with pytest.raises(pkg.GtenHipError):
        b.decode_step_ragged([0] + [1] * 15)             # position 0 does not exist
    with pytest.raises(pkg.GtenHipError):
        b.decode_step(65)
    b.decode_step_ragged([1 + (q % 8) for q in range(16)])     # valid: runs
    assert 0 <= b.decode_result(3, 1 + 3) < cfg.n_vocab
    b.close()


@pytest.mark.parametrize("name,wd,ad", MODES())
@pytest.mark.parametrize("n_seq", [128, 192, 256, 512])
def test_lanes_equal_separate_64_sequence_decoders(hip, name, wd, ad, n_seq):
    """more than 64 sequences run as LANES -- parallel branches of one graph, each on its sequences' rows of every buffer; lanes
    of 128 rows (eight row tiles per W.x workgroup) for q8 / q4 at 128, 256, 384, 512 sequences, lanes of 64 for f16 at this width (at
    n_embd = 2048 f16 runs lanes of 128 too: tests/test_ffn_streamed_gpu.py) and for 192: per sequence the logits and ids of a 64-sequence decoder holding the same sequences, bit for bit, eager and
    replayed, from n = 1 across the attention chunk boundary, and with the sequences at different positions"""
    if n_seq == 512 and name == "f16":
        pytest.skip("f16 runs lanes of 64: at most 256 sequences")
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(wd, ad, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    weights = [host.synth_weight(cfg, 97, i) for i in range(len(cfg.weight_shapes()))]
    N = 266
    streams = [host.synthetic_tokens(N + 2, seed=900 + q, n_vocab=cfg.n_vocab) for q in range(n_seq)]
    big = host.batch(cfg, n_seq)
    for i, w in enumerate(weights):
        big.set_weight(i, w)
    for q in range(n_seq):
        big.decode_begin(q, streams[q])
    for n in range(1, 9):
        big.decode_step(n, n % 2 == 0)                      # eager and single-step replays
    big.decode_steps(9, N - 8, True)                        # four steps per replay
    got_ids = [big.decode_result(q, N) for q in range(n_seq)]
    got = [big.logits(q).copy() for q in range(n_seq)]
    # ragged: every sequence one more step at its own position (half of them stay where they are and redo row N - 1)
    ns = [N + (q % 2) for q in range(n_seq)]
    big.decode_step_ragged(ns, True)
    got_r = [big.logits(q).copy() for q in range(n_seq)]
    big.close()
    for lane in range(n_seq // 64):
        small = host.batch(cfg, 64)
        for i, w in enumerate(weights):
            small.set_weight(i, w)
        for q in range(64):
            small.decode_begin(q, streams[64 * lane + q])
        small.decode_steps(1, N, True)
        for q in range(64):
            assert small.decode_result(q, N) == got_ids[64 * lane + q], (name, lane, q)
            assert np.array_equal(small.logits(q), got[64 * lane + q]), (name, lane, q)
        small.decode_step_ragged(ns[64 * lane:64 * lane + 64], True)
        for q in range(64):
            assert np.array_equal(small.logits(q), got_r[64 * lane + q]), (name, "ragged", lane, q)
        small.close()
