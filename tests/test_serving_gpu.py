"""-m gpu: continuous batching (round-1 verdict "slot admission/eviction").  A queue of prompts is
served through the slots of a multi-sequence decoder: a slot whose sequence ended takes the next prompt while the
others go on (TinyLlamaBatch::serve; gten_hip_decoder_slot_start / _slot_park / _run / _slot_ids).  Up to 8 slots every
sequence's ids must be EXACTLY those of generating it alone on the single-sequence decoder."""
import numpy as np
import pytest

from gpu_common import hip  # noqa: F401
from __graft_entry__ import load_package
from helpers import MODES, Q4, Q8, tiny_config
from test_model_gpu import host_cfg

pytestmark = pytest.mark.gpu


def make_prompts(host, cfg, lengths, seed):
    return [list(host.synthetic_tokens(n, seed=seed + 7 * i + n, n_vocab=cfg.n_vocab)) for i, n in enumerate(lengths)]


@pytest.mark.parametrize("name,wd,ad", MODES())
@pytest.mark.parametrize("n_seq", [2, 8])
def test_queue_through_slots_equals_single_sequence_generation(hip, name, wd, ad, n_seq):
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(wd, ad, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    weights = [host.synth_weight(cfg, 31337, i) for i in range(len(cfg.weight_shapes()))]
    # prompts of 1 .. 260 ids: single-row and matrix-core prompt processing, both sides of the attention chunk boundary,
    # one that fills the context, one with no room at all
    lengths = [5, 40, 1, 17, 260, 9, 33, 100, 2, 64, 300, 319, 320, 12, 250, 7, 21]
    prompts = make_prompts(host, cfg, lengths, 1000)
    total = 320
    m = host.model(cfg)
    for i, w in enumerate(weights):
        m.set_weight(i, w)
    alone = [m.generate(p, total) if len(p) < total else np.array(p, np.int32) for p in prompts]
    eos = int(alone[0][30])                              # a token that comes up: several sequences stop early at it
    want = [m.generate(p, total, eos) if len(p) < total else np.array(p, np.int32) for p in prompts]
    want_new = [m.generate(p, min(total, len(p) + 25), eos) if len(p) < total else np.array(p, np.int32) for p in prompts]
    m.close()
    assert any(len(w) < total for w in want) and any(len(w) == total for w in want)
    b = host.batch(cfg, n_seq)
    for i, w in enumerate(weights):
        b.set_weight(i, w)
    for slice_steps in (16, 5):
        got, st = b.serve(prompts, total, eos, slice_steps)
        assert len(got) == len(prompts)
        for j in range(len(prompts)):
            assert got[j].tolist() == want[j].tolist(), (name, n_seq, slice_steps, j, len(got[j]), len(want[j]))
        assert st["prompt_tokens"] == sum(lengths)
        assert st["new_tokens"] == sum(len(w) - len(p) for w, p in zip(want, prompts))
        assert st["admissions"] >= len(prompts) - 2       # (the prompts with no room are returned as they came)
    # a bound on the new ids per prompt
    got, st = b.serve(prompts, total, eos, 8, max_new=25)
    for j in range(len(prompts)):
        assert got[j].tolist() == want_new[j].tolist(), (name, n_seq, "max_new", j)
    # ... and per prompt (requests that end at different times: slots are refilled while the others decode)
    each = [3 + (7 * j) % 40 for j in range(len(prompts))]
    want_each = []
    m = host.model(cfg)
    for i, w in enumerate(weights):
        m.set_weight(i, w)
    for p, e in zip(prompts, each):
        want_each.append(m.generate(p, min(total, len(p) + e), eos) if len(p) < total else np.array(p, np.int32))
    m.close()
    got, st = b.serve(prompts, total, eos, 4, max_new_each=each)
    for j in range(len(prompts)):
        assert got[j].tolist() == want_each[j].tolist(), (name, n_seq, "max_new_each", j)
    # the decoder still serves the other entry points afterwards (slots view is reset)
    toks = host.synthetic_tokens(20, seed=5, n_vocab=cfg.n_vocab)
    for q in range(n_seq):
        b.decode_begin(q, toks)
    for n in range(1, 21):
        b.decode_step(n, True)
    assert 0 <= b.decode_result(0, 20) < cfg.n_vocab
    b.close()


def test_wide_batch_serves_a_queue(hip):
    """16 slots (W.x on the matrix cores): the queue is served completely; with as many prompts as slots (no admission
    after the start) the ids are those of gten_host_batch_generate on the same batch; longer queues keep every
    invariant (prompt kept, first new id = argmax of the prompt's logits, lengths within bounds)"""
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(Q4, Q8, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    weights = [host.synth_weight(cfg, 2718, i) for i in range(len(cfg.weight_shapes()))]
    b = host.batch(cfg, 16)
    for i, w in enumerate(weights):
        b.set_weight(i, w)
    lengths = [5 + (11 * i) % 90 for i in range(16)]
    prompts = make_prompts(host, cfg, lengths, 50)
    ref = b.generate(prompts, 200)
    got, st = b.serve(prompts, 200, -1, 16)
    for j in range(16):
        assert got[j].tolist() == ref[j].tolist(), j
    lengths = [3 + (17 * i) % 150 for i in range(50)]
    prompts = make_prompts(host, cfg, lengths, 90)
    got, st = b.serve(prompts, 220, -1, 16, max_new=40)
    assert st["admissions"] == 50 and st["new_tokens"] == sum(len(g) - len(p) for g, p in zip(got, prompts))
    for j, (g, p) in enumerate(zip(got, prompts)):
        assert g[: len(p)].tolist() == p and len(g) == min(220, len(p) + 40), j
        lg = b.prefill(0, p)
        assert int(g[len(p)]) == int(np.argmax(lg)), j
        assert ((g >= 0) & (g < cfg.n_vocab)).all()
    b.close()


def test_queue_through_128_slots_equals_16_slots(hip):
    """two lanes of 64 slots in one decoder: a queue longer than the slots gives, prompt by prompt, the ids the 16-slot
    decoder gives (a sequence's rows do not depend on its neighbours or its slot)"""
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(Q4, Q8, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    weights = [host.synth_weight(cfg, 2718, i) for i in range(len(cfg.weight_shapes()))]
    lengths = [3 + (17 * i) % 150 for i in range(300)]
    prompts = make_prompts(host, cfg, lengths, 91)
    budgets = np.array([5 + (7 * i) % 60 for i in range(300)], np.int32)
    outs = []
    for slots in (128, 16):
        b = host.batch(cfg, slots)
        for i, w in enumerate(weights):
            b.set_weight(i, w)
        got, st = b.serve(prompts, 300, -1, 16, max_new_each=budgets)
        assert st["admissions"] == 300
        outs.append(got)
        b.close()
    for j in range(300):
        assert outs[0][j].tolist() == outs[1][j].tolist(), j


def test_lanes_without_a_live_slot_sit_the_run_out(hip):
    """round 4: a shared run leaves out every lane whose slots are all parked, and admission fills the lanes that already run
    first.  The ids must not depend on it: a queue through 256 slots (two lanes of 128 rows) -- short enough that the second
    lane is empty at the start and again in the tail -- with and without lane skipping, and through 16 slots; the skipping run
    must have computed fewer lane-steps than steps x lanes.  In the tail of a queue serve() also MOVES the last sequences into as
    few lanes as they fit (parked, their cache sets bound to free slots of the lanes that stay, started again): covered by the
    same comparison, the 16-slot run never moves anything."""
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(Q4, Q8, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    weights = [host.synth_weight(cfg, 1414, i) for i in range(len(cfg.weight_shapes()))]
    lengths = [16 + (23 * i) % 140 for i in range(330)]
    prompts = make_prompts(host, cfg, lengths, 17)
    budgets = np.array([4 + (11 * i) % 70 for i in range(330)], np.int32)
    outs, stats = [], []
    try:
        for slots, skip in ((256, True), (256, False), (16, True)):
            hip.set_lane_skip(skip)
            b = host.batch(cfg, slots)
            for i, w in enumerate(weights):
                b.set_weight(i, w)
            got, st = b.serve(prompts, 300, -1, 8, max_new_each=budgets)
            assert st["admissions"] == 330
            outs.append(got); stats.append(st)
            b.close()
    finally:
        hip.set_lane_skip(False)
    for j in range(330):
        assert outs[0][j].tolist() == outs[1][j].tolist() == outs[2][j].tolist(), j
    # (without the switch only the TAIL leaves lanes out -- the queue is empty, the last sequences have been moved into as few
    #  lanes as they fit; with it every run does, also while the slots fill)
    assert stats[0]["lane_rows"] == 128 and stats[0]["lane_steps"] <= stats[1]["lane_steps"] < 2 * stats[1]["steps"]
    assert stats[0]["lane_steps"] < 2 * stats[0]["steps"], stats[0]
    print("lane-steps with / without skipping:", stats[0]["lane_steps"], stats[1]["lane_steps"], "steps", stats[0]["steps"], stats[1]["steps"])


def test_slot_api_errors_are_reported(hip):
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(Q4, Q8, n_heads=4, n_kv_heads=2, max_ctx=64, n_layers=1))
    b = host.batch(cfg, 4)
    for i in range(len(cfg.weight_shapes())):
        b.set_weight(i, host.synth_weight(cfg, 1, i))
    with pytest.raises(pkg.GtenHipError):
        b.serve([[1, 2, 3]], 0)                          # nothing to generate into
    with pytest.raises(pkg.GtenHipError):
        b.serve([list(range(1, 70))], 80)                # prompt longer than the context
    got, st = b.serve([[1, 2, 3]], 64, -1, 16)
    assert len(got[0]) == 64
    b.close()


@pytest.mark.parametrize("name,wd,ad", [m for m in MODES() if m[0] == "q4"])
def test_every_admission_schedule_gives_the_same_ids(hip, name, wd, ad):
    """How many prompts get processed beside a slice depends on timing; the ids must not.  Fixed schedules (k prompts per
    slice) x slice lengths walk through many (membership, steps-per-slice) patterns, each compared with generating alone."""
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(wd, ad, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    weights = [host.synth_weight(cfg, 31337, i) for i in range(len(cfg.weight_shapes()))]
    lengths = [5, 40, 1, 17, 260, 9, 33, 100, 2, 64, 300, 319, 12, 250, 7, 21, 255, 256, 257, 130]
    prompts = make_prompts(host, cfg, lengths, 1000)
    total = 320
    m = host.model(cfg)
    for i, w in enumerate(weights):
        m.set_weight(i, w)
    alone = [m.generate(p, total) for p in prompts]
    eos = int(alone[0][30])
    want = [m.generate(p, total, eos) for p in prompts]
    m.close()
    b = host.batch(cfg, 8)
    for i, w in enumerate(weights):
        b.set_weight(i, w)
    try:
        for k in (1, 2, 3, 8):
            for slice_steps in (16, 7, 4, 3):
                b.set_serve_schedule(k)
                got, _ = b.serve(prompts, total, eos, slice_steps)
                for j in range(len(prompts)):
                    assert got[j].tolist() == want[j].tolist(), (k, slice_steps, j, len(prompts[j]))
    finally:
        b.set_serve_schedule(0)
    b.close()


def test_prompts_processed_ahead_onto_spare_cache_sets_give_the_same_ids(hip):
    """serve() fills SPARE cache sets with the prompts to come while every slot is busy and binds a ready set to the next slot
    that ends (gten_hip_decoder_slot_bind) -- which slot, which set and which slice a prompt joins all change with the number
    of spares and the schedule; the ids must not: 0 / 3 / 16 / default spares x two schedules x two slice lengths, 16 slots,
    against the same queue through ONE sequence's generate; afterwards every slot is back on its own sequence's caches
    (a fixed batch generates the same ids as before the queue)."""
    from helpers import Q4, Q8
    pkg = load_package()
    host = pkg.load_host()
    cfg = host_cfg(tiny_config(Q4, Q8, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2))
    weights = [host.synth_weight(cfg, 4711, i) for i in range(len(cfg.weight_shapes()))]
    lengths = [16 + (37 * i) % 200 for i in range(60)] + [5, 300, 17, 256]
    prompts = make_prompts(host, cfg, lengths, 2000)
    budgets = np.array([3 + (11 * i) % 40 for i in range(len(prompts))], np.int32)
    total = 320
    b = host.batch(cfg, 16)
    for i, w in enumerate(weights):
        b.set_weight(i, w)
    cap = max(len(p) for p in prompts[:16]) + 6
    fixed_before = b.generate(prompts[:16], cap)
    runs = []
    try:
        for spares in (0, 3, 16, -1):
            for k, slice_steps in ((0, 8), (2, 5)):
                b.set_serve_spares(spares)
                b.set_serve_ramp(100 if spares in (0, 16) else 0)            # (... and the first slice with every slot filled, or not)
                b.set_serve_schedule(k)
                got, st = b.serve(prompts, total, -1, slice_steps, max_new_each=budgets)
                runs.append(([g.tolist() for g in got], spares, k, st["steps"]))
    finally:
        b.set_serve_schedule(0)
        b.set_serve_spares(-1)
        b.set_serve_ramp(100)
    for got, spares, k, _ in runs[1:]:
        assert got == runs[0][0], (spares, k)
    for j, g in enumerate(runs[0][0]):
        assert len(g) == min(total, len(prompts[j]) + int(budgets[j])), (j, len(g))
    fixed_after = b.generate(prompts[:16], cap)
    assert [x.tolist() for x in fixed_after] == [x.tolist() for x in fixed_before]
    b.close()


def test_full_size_queue_is_repeatable_and_equals_the_fixed_batch(hip):
    """TinyLlama-1.1B shapes, 64 slots, prompts of 64..512 ids processed on stream 1 BESIDE the shared steps: the ids must
    not depend on what overlapped what.  They did for most of round 2 (packed-f32 instructions beside another
    stream's MFMA kernel): run 1 and run 2 differed in ~90 % of the prompts.  Three checks: a fixed admission schedule twice;
    64 prompts through 64 slots twice; and those against the fixed-batch generation (no prompt beside any step)."""
    host = load_package().load_host()
    cfg = host.default_config(Q4, Q8)
    n_seq, n_ctx = 64, 2048
    b = host.batch(cfg, n_seq)
    b.load_synthetic(1234)
    rng = np.random.default_rng(2024)
    lens = rng.integers(64, 513, 96)
    prompts = [list(host.synthetic_tokens(int(n), seed=999 + j)) for j, n in enumerate(lens)]
    budgets = rng.integers(24, 64, len(prompts)).astype(np.int32)

    def differing(x, y):
        return [j for j in range(len(x)) if not np.array_equal(x[j], y[j])]

    b.set_serve_schedule(3)
    r1, _ = b.serve(prompts, n_ctx, -1, 16, max_new_each=budgets)
    r2, _ = b.serve(prompts, n_ctx, -1, 16, max_new_each=budgets)
    assert differing(r1, r2) == []
    for j, p in enumerate(prompts):
        assert len(r1[j]) == len(p) + int(budgets[j]) and r1[j][: len(p)].tolist() == p
    b.set_serve_schedule(0)
    first = prompts[:n_seq]
    total = 560
    s1, _ = b.serve(first, total, -1, 16)
    s2, _ = b.serve(first, total, -1, 16)
    assert differing(s1, s2) == []
    g = b.generate(first, total, -1)
    assert differing(s1, g) == []
