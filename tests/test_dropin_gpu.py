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
