"""Host tokenizer (tinyllama.cpp_amd/host/tokenizer.h) against the reference's tokenizer.h.

The vocabulary file (tokenizer.bin) is the reference's data asset and is not copied into this repository: these tests run
where it is available (/root/reference in the build container, or $GTEN_TOKENIZER_BIN) and skip elsewhere.  Pins:
  * the known answer written in the reference (tinyllama.cpp:101-104),
  * golden ids / pieces produced by the real reference code (tests/golden/make_tokenizer_golden.py),
  * where oracle/_ref is built: the live reference on random text (ASCII, UTF-8, stray continuation bytes) and on every
    id of the vocabulary.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from __graft_entry__ import load_package

VOCAB = os.environ.get("GTEN_TOKENIZER_BIN", "/root/reference/tokenizer.bin")
HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.skipif(not os.path.exists(VOCAB), reason="tokenizer.bin (the reference's vocabulary file) is not available")


@pytest.fixture(scope="module")
def tok():
    pkg = load_package()
    host = pkg.load_host()
    t = host.tokenizer(VOCAB)
    yield t
    t.close()


def test_known_answer_of_the_reference(tok):
    # tinyllama.cpp:101-104
    assert tok.encode("Who is Karl Marx?") == [1, 32001, 1404, 13, 22110, 338, 8425, 28579, 29973, 32002, 29871, 13, 32001, 20255, 13]


def test_golden_ids_and_pieces(tok):
    pins = json.load(open(os.path.join(HERE, "golden", "tokenizer_pins.json")))
    for e in pins["encode"]:
        assert tok.encode(e["prompt"]) == e["ids"], e["prompt"]
    for prev, token, piece_hex in pins["decode_hex"]:
        assert tok.decode(prev, token).hex() == piece_hex, (prev, token)


def test_chat_template_wraps_the_plain_ids(tok):
    plain = tok.encode("user\nhello there", chat_template=False)
    assert tok.encode("hello there") == [1, 32001] + plain + [32002, 29871, 13, 32001, 20255, 13]
    assert tok.encode("", chat_template=False) == []


def _reference():
    from oracle import orc
    ref = orc.load_ref("avx")
    if ref is None or not hasattr(ref.lib, "ref_tok_create"):
        return None
    L = ref.lib
    L.ref_tok_create.restype = C.c_void_p; L.ref_tok_create.argtypes = [C.c_char_p, C.c_int]
    L.ref_tok_encode.restype = C.c_int; L.ref_tok_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
    L.ref_tok_decode.restype = C.c_char_p; L.ref_tok_decode.argtypes = [C.c_void_p, C.c_int, C.c_int]
    return L, L.ref_tok_create(VOCAB.encode(), 32000)


def test_against_the_live_reference(tok):
    r = _reference()
    if r is None:
        pytest.skip("oracle/_ref is not built here")
    L, rt = r
    rng = np.random.default_rng(11)
    alphabet = list("abcdefghijklmnopqrstuvwxyz ABCDEFG 0123456789 .,;:!?'\"()-_\n\t") + ["é", "ß", "ø", "日", "本", "語", "😀", "→", "ᚠ", "  ", "the ", "ing "]
    buf = np.zeros(8192, np.int32)
    for trial in range(300):
        n = int(rng.integers(0, 120))
        text = "".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet), n)).encode("utf-8")
        if trial % 10 == 9:                       # stray continuation / truncated lead bytes
            raw = bytearray(text)
            for _ in range(3):
                raw.insert(int(rng.integers(0, len(raw) + 1)), int(rng.choice([0x80, 0xBF, 0xC3, 0xE2, 0xF0])))
            text = bytes(b for b in raw if b != 0)
        cnt = L.ref_tok_encode(rt, text, buf.ctypes.data_as(C.c_void_p), len(buf))
        assert cnt > 0
        assert tok.encode(text) == buf[:cnt].tolist(), text
    for prev in (0, 1):
        for token in list(range(0, 32000)) + [32000, 32001, 32002, 40000]:
            assert tok.decode(prev, token) == L.ref_tok_decode(rt, prev, token), (prev, token)
