"""-m gpu: the batch-1 decode step as ONE persistent launch (csrc/gten_decode_persist.h) against the launch chain it
replaces -- the same arithmetic with the same rounding points, so the same bytes: argmax ids, logits, and (through the
steps that follow) every K / V row; across the 256-position attention chunks, eagerly and replayed from a graph, with the
sampler on the device; and inside the band around the ORACLE on a model small enough for it.  The abort word of its
bounded polls must stay clear."""
import numpy as np
import pytest

from gpu_common import hip  # noqa: F401
from __graft_entry__ import load_package
from helpers import Q4, Q8, tiny_config
from test_model_gpu import check_logits, host_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_the_persistent_step(hip):
    """round 5: the persistent step (0.90 x the launch chain, HISTORY.md) is compiled only with -DGTEN_WITH_PERSIST=1
    (GTEN_HIP_EXTRA_FLAGS="-DGTEN_WITH_PERSIST=1" python -m pytest tests/test_persist_gpu.py); the product library refuses
    gten_hip_set_decode_persistent(1)"""
    pkg = load_package()
    try:
        hip.set_decode_persistent(True)
    except pkg.GtenHipError:
        pytest.skip("libgten_hip.so was built without the persistent step (-DGTEN_WITH_PERSIST=1)")
    hip.set_decode_persistent(False)


def _pair(hip, host, cfg, weights=None, seed=1234):
    """(persistent, chain) models on the same weights; the launch choice is made when the decoder is created"""
    out = []
    for on in (True, False):
        hip.set_decode_persistent(on)
        m = host.model(cfg)
        if weights is None:
            m.load_synthetic(seed)
        else:
            for i, w in enumerate(weights):
                m.set_weight(i, w)
        out.append(m)
    return out


def _begin(hip, models, toks):
    for on, m in zip((True, False), models):
        hip.set_decode_persistent(on)
        m.decode_begin(toks)
    hip.set_decode_persistent(False)


def test_persistent_step_equals_launch_chain_full_size(hip):
    pkg = load_package()
    host = pkg.load_host()
    cfg = host.default_config(Q4, Q8)
    toks = host.synthetic_tokens(2048, seed=12345, n_vocab=cfg.n_vocab)
    pers, chain = _pair(hip, host, cfg)
    try:
        _begin(hip, (pers, chain), toks)
        launches0 = hip.persist_status()[1]
        watch = (1, 2, 3, 17, 255, 256, 257, 258, 511, 512, 513, 700)
        ids = {True: {}, False: {}}
        for key, m in ((True, pers), (False, chain)):
            for n in range(1, 301):
                m.decode_step(n, n % 3 != 0)              # graph replays and eager launches
                if n in watch:
                    ids[key][n] = m.decode_result(n)
            m.decode_steps(301, 400, True)                # free-running, four steps per replay
            for n in watch:
                if n > 300:
                    ids[key][n] = m.decode_result(n)
        assert ids[True] == ids[False], (ids[True], ids[False])
        a = pers.logits(toks[:700], 699)
        b = chain.logits(toks[:700], 699)
        assert np.array_equal(a, b), float(np.abs(a - b).max())
        nd, launches, ab, _ = hip.persist_status()
        assert nd == 1, "the full-size q4 model must qualify for the persistent step"
        assert ab == 0, "a poll of the persistent step gave up (code 0x%x)" % ab
        assert launches > launches0, "the persistent launch never ran"
    finally:
        hip.set_decode_persistent(False)
        pers.close(); chain.close()


def test_persistent_step_mid_size_against_oracle_and_chain(hip, oracle):
    """n_embd 512 (a quarter of the prologue threads idle), 8 heads of 64 in 2 kv groups, two attention chunks, three
    blocks: the same bytes as the chain, and both inside the band around the oracle"""
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(Q4, Q8, n_heads=8, n_kv_heads=2, n_embd=512, n_ffn=1024, n_layers=3, n_vocab=1000, max_ctx=512)
    cfg = host_cfg(ocfg)
    weights = [host.synth_weight(cfg, 4242, i) for i in range(len(cfg.weight_shapes()))]
    om = oracle.model(ocfg)
    for i, w in enumerate(weights):
        om.set_weight(i, w)
    N = 270
    toks = host.synthetic_tokens(N, seed=31, n_vocab=cfg.n_vocab)
    pers, chain = _pair(hip, host, cfg, weights)
    try:
        _begin(hip, (pers, chain), toks)
        watch = (1, 2, 40, 255, 256, 257, 258, N)
        got = {}
        for n in range(1, N + 1):
            pers.decode_step(n, n % 2 == 0)
            chain.decode_step(n, n % 2 == 0)
            if n in watch:
                got[n] = (pers.decode_result(n), chain.decode_result(n), pers.logits(toks[:n], n - 1).copy(), chain.logits(toks[:n], n - 1).copy())
        for n in range(1, N + 1):
            want = om.logits(toks[:n], n - 1)
            if n not in watch:
                continue
            ia, ib, la, lb = got[n]
            assert ia == ib and np.array_equal(la, lb), n
            check_logits("q4", la, want, float(want.std()))
        nd, _, ab, _ = hip.persist_status()
        assert nd == 1, "the mid-size model must qualify for the persistent step"
        assert ab == 0
    finally:
        hip.set_decode_persistent(False)
        pers.close(); chain.close(); om.close()


def test_persistent_step_with_the_sampler_on_the_device(hip):
    """greedy generation: each step's argmax becomes the next input id on the device (the last workgroup's epilogue)"""
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(Q4, Q8, n_heads=8, n_kv_heads=2, n_embd=512, n_ffn=1024, n_layers=2, n_vocab=1000, max_ctx=512)
    cfg = host_cfg(ocfg)
    weights = [host.synth_weight(cfg, 777, i) for i in range(len(cfg.weight_shapes()))]
    prompt = host.synthetic_tokens(9, seed=3, n_vocab=cfg.n_vocab)
    outs = []
    try:
        for on in (True, False):
            hip.set_decode_persistent(on)
            m = host.model(cfg)
            for i, w in enumerate(weights):
                m.set_weight(i, w)
            outs.append(m.generate(prompt, 300))
            m.close()
        assert np.array_equal(outs[0], outs[1])
        assert hip.persist_status()[2] == 0
    finally:
        hip.set_decode_persistent(False)
