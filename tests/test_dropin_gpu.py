"""-m gpu: the drop-in boundary exercised by the REFERENCE'S OWN caller code.

oracle/_ref/libdropin.so is the reference's unmodified tinyllama.cpp (TinyLlama class, .gten
loader, module wiring, `Tensor`/`Linear`/... used exactly as upstream uses them) compiled against
this repository's gten headers and linked with libgten_hip.so (oracle/Makefile `dropin`; built
where /root/reference exists, the .so travels).  If the gten API were not intact it would not
compile; if the kernels behind it were wrong these logits would not match the reference's."""
import os

import numpy as np
import pytest

from gpu_common import hip  # noqa: F401
from __graft_entry__ import load_package
from helpers import MODES, Q4, Q8, tiny_config
from oracle import orc
from test_golden_gpu import G, band

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dropin(hip):
    lib = orc.load_dropin()
    if lib is None:
        pytest.skip("oracle/_ref/libdropin.so not built (needs /root/reference at build time)")
    return lib


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_reference_module_wiring_on_hbm_tensors(dropin, name, wd, ad):
    """Embedding / AttentionBlock / RMSNorm / EmbeddingLinear assembled as TinyLlama's ctor does,
    against the logits the real reference produced (tiny_model_golden.npz)."""
    g = np.load(os.path.join(G, "tiny_model_golden.npz"))
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(wd, ad, n_heads=4, n_kv_heads=2)
    cfg = pkg.HostConfig(**{k: getattr(ocfg, k) for k, _ in ocfg._fields_})
    m = dropin.model(ocfg)
    assert m.n_weights() == len(cfg.weight_shapes())
    for i in range(m.n_weights()):
        w = host.synth_weight(cfg, int(g["seed"][0]), i)
        assert m.weight_bytes(i) == w.size
        m.set_weight(i, w)
    toks, want = g[f"{name}.tokens"], g[f"{name}.logits.avx"]
    for step in range(want.shape[0]):
        n = 9 + step
        got = m.logits(toks[:n], 0 if step == 0 else n - 1)
        band(name, got - want[step], float(want[step].std()))
        if name == "f16":
            assert int(np.argmax(got)) == int(np.argmax(want[step]))
    m.close()


def test_reference_tinyllama_class_loads_gten_and_decodes(dropin, tmp_path):
    """the reference's TinyLlama::load_from_ckpt + logits(), full-size q4, on a synthetic .gten file"""
    path = os.path.join(G, "full_model_golden.npz")
    if not os.path.exists(path):
        pytest.skip("full_model_golden.npz not generated")
    g = np.load(path)
    pkg = load_package()
    host = pkg.load_host()
    cfg = host.default_config(Q4, Q8)
    gten_path = str(tmp_path / "tinyllama.q4.gten")
    host.write_gten(cfg, int(g["seed"][0]), gten_path)
    m = dropin.tinyllama(64, Q4, Q8)
    m.load(gten_path)
    toks, probe = g["q4.avx.tokens"], g["probe_ids"]
    for step in range(3):
        n = 15 + step
        lg = m.logits(toks[:n], 0 if step == 0 else n - 1)
        ids = g["q4.avx.top_ids"][step]
        ref_vals = np.concatenate([g["q4.avx.top_logits"][step], g["q4.avx.probes"][step]])
        band("q4", np.concatenate([lg[ids], lg[probe]]) - ref_vals, float(g["q4.avx.stats"][step][1]))
    m.close()


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_single_row_calls_run_as_one_fused_decoder_step(dropin, name, wd, ad):
    """the reference's unmodified module calls (Embedding -> AttentionBlock x L -> RMSNorm -> EmbeddingLinear with one
    new row, tinyllama.cpp:45-61) are recorded by gten/modules.h and run as ONE fused decoder step.  Against the same
    calls with the recording switched off (operator by operator): the same bytes while the context fits one attention
    chunk (n <= 256), the model band beyond (chunked softmax: f32 summation order)."""
    from test_model_gpu import check_logits
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(wd, ad, n_heads=4, n_kv_heads=2, max_ctx=320, n_layers=2)
    cfg = pkg.HostConfig(**{k: getattr(ocfg, k) for k, _ in ocfg._fields_})
    toks = host.synthetic_tokens(300, seed=91, n_vocab=cfg.n_vocab)
    runs = []
    for fused in (True, False):
        dropin.set_fused_rows(fused)
        m = dropin.model(ocfg)
        for i in range(m.n_weights()):
            m.set_weight(i, host.synth_weight(cfg, 606, i))
        out = {9: m.logits(toks[:9], 0)}                      # prompt: several rows, operator path either way
        for n in range(10, 301):
            lg = m.logits(toks[:n], n - 1)
            if n in (10, 11, 40, 255, 256, 257, 300):
                out[n] = lg
        runs.append(out)
        m.close()
    dropin.set_fused_rows(True)
    for n, a in runs[0].items():
        b = runs[1][n]
        if n <= 256:
            assert np.array_equal(a, b), (name, n, float(np.abs(a - b).max()))
        else:
            check_logits(name, a, b, float(b.std()))
    assert not np.array_equal(runs[0][10], runs[0][11])


def test_interrupted_single_row_forward_settles_through_the_operators(dropin):
    """a caller that stops after k blocks and reads the activation row (or starts another forward) gets exactly what
    the operator path computes: the recorded calls are settled, never dropped"""
    pkg = load_package()
    host = pkg.load_host()
    ocfg = tiny_config(Q4, Q8, n_heads=4, n_kv_heads=2, max_ctx=64, n_layers=2)
    cfg = pkg.HostConfig(**{k: getattr(ocfg, k) for k, _ in ocfg._fields_})
    toks = host.synthetic_tokens(24, seed=5, n_vocab=cfg.n_vocab)
    rows = []
    for fused in (True, False):
        dropin.set_fused_rows(fused)
        m = dropin.model(ocfg)
        for i in range(m.n_weights()):
            m.set_weight(i, host.synth_weight(cfg, 17, i))
        got = [m.logits(toks[:9], 0)]
        got.append(m.partial_row(toks[:10], 9, 0))            # embedding row only
        got.append(m.partial_row(toks[:10], 9, 1))            # + first block (writes K/V row 9 of block 0 again)
        got.append(m.logits(toks[:10], 9))                    # the whole row
        got.append(m.partial_row(toks[:11], 10, 2))           # both blocks, no head
        got.append(m.logits(toks[:11], 10))
        got.append(m.logits(toks[:14], 11))                   # three new rows: operators
        got.append(m.logits(toks[:15], 14))
        rows.append(got)
        m.close()
    dropin.set_fused_rows(True)
    for i, (a, b) in enumerate(zip(*rows)):
        assert np.array_equal(a, b), i


def test_reference_greedy_loop_on_the_fused_rows(dropin, tmp_path):
    """the reference's TinyLlama class + its greedy loop (logits() per token, host argmax), full-size q4: the ids with
    the single-row recording on equal the ids operator by operator"""
    pkg = load_package()
    host = pkg.load_host()
    cfg = host.default_config(Q4, Q8)
    gten_path = str(tmp_path / "tinyllama.q4.gten")
    host.write_gten(cfg, 1234, gten_path)
    prompt = host.synthetic_tokens(15, seed=12345)
    ids = []
    for fused in (True, False):
        dropin.set_fused_rows(fused)
        m = dropin.tinyllama(64, Q4, Q8)
        m.load(gten_path)
        ids.append(m.greedy(prompt, 40).tolist())
        m.close()
    dropin.set_fused_rows(True)
    assert len(ids[0]) == 40 and ids[0] == ids[1]
