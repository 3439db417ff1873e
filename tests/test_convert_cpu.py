"""not gpu: the HF -> .gten converter (tinyllama.cpp_amd/convert.py) against the byte pins generated from the
reference's converter (tests/golden/converter_pins.npz, tinyllama_to_gten.py:24-148) and, end to end, a fake
checkpoint -> .gten -> the oracle's restatement of the reference loader and the product's own loader."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402
from helpers import MODES, tiny_config, weight_shapes  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def conv():
    pkg = load_package()
    import importlib.util
    spec = importlib.util.spec_from_file_location("gten_convert", os.path.join(os.path.dirname(pkg.__file__), "convert.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_quantizers_match_the_reference_converter_pins(conv):
    g = np.load(os.path.join(G, "converter_pins.npz"))
    w = g["w"]
    assert np.array_equal(conv.quantize_q8(w), g["q8"])
    assert np.array_equal(conv.quantize_q4(w), g["q4"])
    assert np.array_equal(conv.to_f16(w), g["f16"])


@pytest.mark.parametrize("tag,src", [("bf16", "bf16"), ("f16src", "f16")])
def test_quantizers_follow_the_checkpoint_dtype(conv, tag, src):
    """a bf16 (TinyLlama's own) or f16 checkpoint: the reference forms absmax / qmax in THAT dtype, so deltas and quants
    differ from the f32 path; pins from the reference converter run on bf16 / f16 tensors"""
    g = np.load(os.path.join(G, "converter_pins.npz"))
    w = g[f"w_{tag}"]
    assert np.array_equal(conv.quantize_q8(w, src), g[f"q8_{tag}"])
    assert np.array_equal(conv.quantize_q4(w, src), g[f"q4_{tag}"])
    assert np.array_equal(conv.to_f16(w), g[f"f16_{tag}"])
    assert not np.array_equal(conv.quantize_q8(w, "f32"), g[f"q8_{tag}"])       # the dtype matters
    with pytest.raises(ValueError):
        conv.quantize_q8(w, "f64")


def test_bf16_checkpoint_file_is_read(conv, tmp_path):
    torch = pytest.importorskip("torch")
    from safetensors.torch import save_file
    g = np.load(os.path.join(G, "converter_pins.npz"))
    t = torch.from_numpy(g["w_bf16"].copy()).to(torch.bfloat16)
    save_file({"model.embed_tokens.weight": t, "lm_head.weight": t.to(torch.float16), "model.norm.weight": t[0].to(torch.float32),
               "bad": torch.zeros(4, dtype=torch.int32)}, str(tmp_path / "m.safetensors"))
    get, n_layers = conv.open_checkpoint(str(tmp_path / "m.safetensors"))
    w, src = get("model.embed_tokens.weight")
    assert src == "bf16" and np.array_equal(w, g["w_bf16"])
    assert np.array_equal(conv.quantize(w, "q4", src), g["q4_bf16"])
    assert get("lm_head.weight")[1] == "f16" and get("model.norm.weight")[1] == "f32"
    with pytest.raises(ValueError):
        get("bad")


def test_quantizers_match_the_oracle_on_random_and_degenerate_rows(conv, oracle):
    from helpers import F16, Q4, Q8
    r = np.random.default_rng(5)
    w = (r.standard_normal((6, 256)) * 0.02).astype(np.float32)
    w[1, :32] = 0.0                                   # a zero block: delta 0, quants 0
    w[2, 32:64] = np.float32(1e-30)                   # delta underflows in fp16
    w[3, 5] = 3.0
    w[4, 64:96] = np.linspace(-1, 1, 32, dtype=np.float32) * 0.5   # exact ties after scaling
    assert np.array_equal(conv.quantize_q8(w), oracle.quantize_weight(w, Q8))
    assert np.array_equal(conv.quantize_q4(w), oracle.quantize_weight(w, Q4))
    assert np.array_equal(conv.to_f16(w), oracle.quantize_weight(w, F16))
    with pytest.raises(ValueError):
        conv.quantize_q8(np.zeros((2, 40), np.float32))


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_fake_checkpoint_round_trip(conv, oracle, tmp_path, name, wd, ad):
    """safetensors checkpoint with HF key names -> .gten -> (a) record framing and order, (b) the oracle's loader
    accepts it (magic, sizes, order: tinyllama.cpp:301-392), (c) payloads are the quantized tensors byte for byte"""
    from safetensors.numpy import save_file
    ocfg = tiny_config(wd, ad, n_layers=2)
    r = np.random.default_rng(11)
    names = conv.tensor_names(ocfg.n_layers)
    shapes = weight_shapes(ocfg)
    assert len(names) == len(shapes)
    tensors = {}
    for (key, is_lin), (rows, cols, _) in zip(names, shapes):
        tensors[key] = ((r.standard_normal((rows, cols)) * 0.02) if is_lin else (1 + 0.05 * r.standard_normal(cols))).astype(np.float32)
    # two shards, like a real checkpoint directory
    keys = list(tensors)
    ck = tmp_path / "ckpt"
    ck.mkdir()
    save_file({k: tensors[k] for k in keys[::2]}, str(ck / "model-00001-of-00002.safetensors"))
    save_file({k: tensors[k] for k in keys[1::2]}, str(ck / "model-00002-of-00002.safetensors"))
    get, n_layers = conv.open_checkpoint(str(ck))
    assert n_layers == ocfg.n_layers
    out = str(tmp_path / f"tiny.{name}.gten")
    nbytes = conv.write_gten(out, get, n_layers, name)
    assert nbytes == os.path.getsize(out)
    m = oracle.model(ocfg)
    m.load_gten(out)                                  # asserts magic, order and every payload size
    m.close()
    raw = open(out, "rb").read()
    assert raw[:8] == conv.GTEN_MAGIC.to_bytes(8, "little")
    pos = 8
    for (key, is_lin), (rows, cols, _) in zip(names, shapes):
        for _ in range(2):
            ln = int.from_bytes(raw[pos:pos + 4], "little"); pos += 4
            assert raw[pos:pos + ln].decode() == key; pos += ln
        sz = int.from_bytes(raw[pos:pos + 4], "little"); pos += 4
        want = conv.quantize(tensors[key], name) if is_lin else conv.to_f16(tensors[key].reshape(1, -1))
        assert sz == want.size and raw[pos:pos + sz] == want.tobytes(), key
        pos += sz
    assert pos == len(raw)
