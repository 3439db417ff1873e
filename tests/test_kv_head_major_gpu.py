"""Head-major K / V shadows of the wide decoders (csrc/gten_decode_attn_hm.h, round 5): the new attention kernel against round 4's
kernel on the cache rows, and -- the property the layout change hangs on -- a shadow can never be stale: whatever writes into
a sequence's cache rows through the C-ABI (a prompt on the operator path, a batched prompt's row copy, ANOTHER decoder's
appends), the shared decoder's next step sees it.

The yardstick of every test here is the same sequence of calls on a decoder that keeps NO shadows (gten_hip_set_kv_head_major(0):
k_dec_attn_mm_g reads the rows as they lie), so a stale shadow shows up as logits of another context, not as a tolerance question.
The parity of the new kernel with the ORACLE and the reference's goldens is tests/test_multiseq_oracle_gpu.py (default: shadows on)."""
import numpy as np
import pytest

from gpu_common import hip  # noqa: F401
from __graft_entry__ import load_package
from helpers import MODES, Q4, Q8, tiny_config
from test_model_gpu import host_cfg

pytestmark = pytest.mark.gpu

QMODES = MODES()          # f16 too: its wide decoders keep f16 shadows (k_dec_attn_hm_f16) in place of the two-launch pair on the rows


def make_batch(hip, host, cfg, n_seq, head_major, seed=777):
    hip.set_kv_head_major(head_major)
    try:
        b = host.batch(cfg, n_seq)
        for i in range(len(cfg.weight_shapes())):
            b.set_weight(i, host.synth_weight(cfg, seed, i))
        b.decode_begin(0, host.synthetic_tokens(4, seed=1, n_vocab=cfg.n_vocab))     # (creates the shared decoder under the switch)
    finally:
        hip.set_kv_head_major(True)
    assert b.kv_info()[0] == head_major
    return b


def close(a, b, what, rel=4e-2):
    """two kernels that differ in the order of f32 additions only: a sum next to a Q8 rounding tie may flip one quant, which moves
    a handful of logits by a few percent of the logit spread -- not the O(1) of a wrong context"""
    d = a.astype(np.float64) - b.astype(np.float64)
    s = float(b.std())
    rms, mx = float(np.sqrt((d * d).mean())), float(np.abs(d).max())
    assert rms <= rel * s and mx <= 10 * rel * s, (what, rms / s, mx / s)
    return rms / s


@pytest.mark.oracle_parity
@pytest.mark.parametrize("name,wd,ad", QMODES)
@pytest.mark.parametrize("heads,kv,n_seq", [(8, 1, 16), (8, 2, 32), (4, 2, 16), (8, 4, 48), (4, 4, 64)])
def test_head_major_kernel_tracks_the_row_layout_kernel(hip, oracle, name, wd, ad, heads, kv, n_seq):
    """every group size (8 / 4 / 2 / 1 query heads per kv head), 16-64 sequences on both sides of the 256-position chunk
    boundaries (so: chunks with and without the new position, partial tiles, a chunk that holds ONLY the new position): logits
    beside the row-layout kernel's, and inside the oracle's band"""
    from test_model_gpu import check_logits
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(wd, ad, n_embd=64 * heads, n_heads=heads, n_kv_heads=kv, max_ctx=576, n_layers=2)
    cfg = host_cfg(ocfg)
    N = 530
    streams = [host.synthetic_tokens(N, seed=900 + q, n_vocab=cfg.n_vocab) for q in range(n_seq)]
    probes = (1, 2, 16, 17, 33, 255, 256, 257, 258, 300, 512, 513, 514, N)
    held = (0, 1, n_seq // 2, n_seq - 1)
    weights = [host.synth_weight(cfg, 777, i) for i in range(len(cfg.weight_shapes()))]
    snaps = []
    for hm in (True, False):
        b = make_batch(hip, host, cfg, n_seq, hm)
        for q in range(n_seq):
            b.decode_begin(q, streams[q])
        snap = {}
        for n in range(1, N + 1):
            b.decode_step(n, True)
            if n in probes:
                snap[n] = [(b.decode_result(q, n), b.logits(q).copy()) for q in held]
        snaps.append(snap)
        assert b.kv_info()[1] == (n_seq if hm else 0)        # one import per sequence at the start, none while it continues
        b.close()
    worst = 0.0
    for n in probes:
        for (ia, la), (ib, lb), q in zip(snaps[0][n], snaps[1][n], held):
            assert np.isfinite(la).all()
            worst = max(worst, close(la, lb, (name, heads, kv, n_seq, q, n)))
            top2 = np.sort(lb)[-2:]
            if top2[1] - top2[0] > 0.05 * float(lb.std()):
                assert ia == ib, (name, q, n)
    # the oracle on one sequence (operator by operator on the CPU): the new kernel inside the model band
    om = oracle.model(ocfg)
    for i, w in enumerate(weights):
        om.set_weight(i, w)
    for n in range(1, 301):
        want = om.logits(streams[held[1]][:n], n - 1)
        if n in probes:
            check_logits(name, snaps[0][n][1][1], want, float(want.std()))
    om.close()
    print(f"{name} heads {heads}/{kv} S={n_seq}: head-major vs row kernel worst rms {worst:.2e} of the logit spread")


@pytest.mark.parametrize("name,wd,ad", QMODES)
def test_rows_rewritten_between_steps_reach_the_shadow(hip, name, wd, ad):
    """a sequence's cache rows rewritten BETWEEN two consecutive steps of the shared decoder -- no slot call, no position change,
    nothing that tells the decoder -- by (a) a short prompt on the operator path, (b) a batched prompt (row copy out of the
    shared prompt matrix), (c) the sequence's own single-sequence decoder appending rows: the next shared step must use them"""
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(wd, ad, n_embd=512, n_heads=8, n_kv_heads=2, max_ctx=320, n_layers=2))
    S, P = 16, 40
    A = [host.synthetic_tokens(P + 24, seed=40 + q, n_vocab=cfg.n_vocab) for q in range(S)]
    B = [host.synthetic_tokens(P + 24, seed=70 + q, n_vocab=cfg.n_vocab) for q in range(S)]

    def play(hm):
        b = make_batch(hip, host, cfg, S, hm)
        out = []
        for q in range(S):
            b.prefill(q, A[q][:P], want=False)               # (>= 16 ids: segments of the shared row matrix + row copies)
            b.decode_begin(q, A[q])
        for n in range(P + 1, P + 5):
            b.decode_step(n, True)
        out.append(b.logits(3).copy())
        imports0 = b.kv_info()[1]
        # (a) sequence 3: rows [0, 15) from another prompt on the OPERATOR path (fewer than 16 ids) -- the steps go on at P + 5
        b.prefill(3, B[3][:15], want=False)
        # (b) sequence 5: a whole other prompt of P ids through the batched prompt path
        b.prefill(5, B[5][:P], want=False)
        # (c) sequence 7: its own decoder re-decodes rows 8 .. P + 3 with OTHER ids (appends K / V rows of another history)
        b.seq_steps(7, np.concatenate([A[7][:8], B[7][8:]]), 9, P + 4 - 8)
        b.decode_step(P + 5, True)
        out += [b.logits(q).copy() for q in (3, 5, 7, 8)]
        imports1 = b.kv_info()[1]
        for n in range(P + 6, P + 9):
            b.decode_step(n, True)
        out += [b.logits(q).copy() for q in (3, 5, 7, 8)]
        assert b.kv_info()[1] == imports1                    # nothing written since: nothing imported again
        b.close()
        return out, imports1 - imports0

    got, imported = play(True)
    want, none = play(False)
    assert none == 0 and imported == 3, (imported, none)     # exactly the three rewritten sequences
    for i, (g, w) in enumerate(zip(got, want)):
        close(g, w, (name, "snapshot", i))
    # ... and the rewrites did change the logits (the test would pass vacuously otherwise): sequences 3, 5, 7 moved, 8 did not
    ctrl = make_batch(hip, host, cfg, S, True)
    for q in range(S):
        ctrl.prefill(q, A[q][:P], want=False)
        ctrl.decode_begin(q, A[q])
    for n in range(P + 1, P + 6):
        ctrl.decode_step(n, True)
    for k, q in enumerate((3, 5, 7)):
        d = got[1 + k] - ctrl.logits(q)
        moved = float(np.sqrt((d * d).mean())) / float(got[1 + k].std())
        print(f"{name}: rewriting sequence {q}'s rows moved its logits by {moved:.3f} of their spread")
        assert moved > 0.3, q                              # (measured 0.6 - 1.3; two kernels on the SAME rows differ by 0.014 - 0.018)
    assert np.array_equal(got[4], ctrl.logits(8))
    ctrl.close()


def test_restarted_and_rebound_slots_never_read_an_old_shadow(hip):
    """the slot calls of continuous batching: a slot restarted on the same rows after they were refilled, on ANOTHER set of rows
    (slot_bind: prompts processed ahead), parked and started again -- generate()/serve() through 16 slots with and without
    shadows give the same ids for a queue that makes every slot end and restart several times"""
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(Q4, Q8, n_embd=512, n_heads=8, n_kv_heads=2, max_ctx=320, n_layers=2))
    r = np.random.default_rng(5)
    prompts = [host.synthetic_tokens(int(r.integers(16, 120)), seed=500 + j, n_vocab=cfg.n_vocab) for j in range(70)]
    each = [int(r.integers(4, 40)) for _ in prompts]
    runs = {}
    for hm in (True, False):
        for spares, slice_steps in ((0, 5), (8, 5), (8, 3)):
            b = make_batch(hip, host, cfg, 16, hm)
            b.set_serve_spares(spares)
            ids, st = b.serve(prompts, 320, eos=-1, slice_steps=slice_steps, max_new_each=each)
            runs[(hm, spares, slice_steps)] = ids
            imports = b.kv_info()[1]
            assert (imports >= len(prompts)) if hm else imports == 0, (hm, spares, imports)
            b.close()
    # one kernel, three schedules (which slot, which set of rows, which slice a prompt gets): the same ids, id for id -- a shadow
    # left over from a slot's previous sequence or from the rows it was bound to before would show here
    for hm in (True, False):
        base = runs[(hm, 0, 5)]
        for key, ids in runs.items():
            if key[0] == hm:
                for j, (a, b_) in enumerate(zip(ids, base)):
                    assert np.array_equal(a, b_), (key, j)
    # the two kernels against each other: greedy ids of near-flat logits part ways at a near-tie now and then and stay apart from
    # there on, so: same lengths, and most prompts id for id
    same = sum(int(np.array_equal(a, b_)) for a, b_ in zip(runs[(True, 0, 5)], runs[(False, 0, 5)]))
    print(f"head-major vs row kernel through 16 slots: {same} of {len(prompts)} prompts id for id")
    assert same >= (3 * len(prompts)) // 4, same
    for a, b_ in zip(runs[(True, 0, 5)], runs[(False, 0, 5)]):
        assert len(a) == len(b_)
