"""Golden vectors of the reference's tokenizer (tokenizer.h) for host/tokenizer.h.

Run in the build container (needs /root/reference and oracle/_ref/libref_avx.so, i.e. __graft_entry__.build()):
    python tests/golden/make_tokenizer_golden.py
writes tests/golden/tokenizer_pins.json: prompts with the ids of Tokenizer::encode (chat template included) and
(prev, token, piece) triples of Tokenizer::decode, all produced by the REAL reference code on its own tokenizer.bin.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

VOCAB = "/root/reference/tokenizer.bin"
PROMPTS = [
    "Who is Karl Marx?",                                   # the known answer in tinyllama.cpp:101-104
    "Give three tips for staying healthier.",
    "",
    " ",
    "hello",
    "Hello, world! 12345 -- tabs\tand\nnewlines\r\n",
    "naïve café déjà vu — “quotes” … ¿qué?",
    "日本語のテキストと中文文本",
    "emoji 😀🚀 and rare ᚠᚢᚦ runes",
    "def f(x):\n    return x ** 2  # code\n",
    "a" * 70 + " " + "b" * 3,
    "    leading spaces and trailing   ",
    "MixedCASE_with-symbols+=/\\|<>[]{}()!@#$%^&*~`",
    "The quick brown fox jumps over the lazy dog. " * 3,
]


def main():
    ref = orc.load_ref("avx")
    assert ref is not None and os.path.exists(VOCAB)
    L = ref.lib
    L.ref_tok_create.restype = C.c_void_p; L.ref_tok_create.argtypes = [C.c_char_p, C.c_int]
    L.ref_tok_encode.restype = C.c_int; L.ref_tok_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
    L.ref_tok_decode.restype = C.c_char_p; L.ref_tok_decode.argtypes = [C.c_void_p, C.c_int, C.c_int]
    t = L.ref_tok_create(VOCAB.encode(), 32000)
    buf = np.zeros(4096, np.int32)
    enc = []
    for p in PROMPTS:
        n = L.ref_tok_encode(t, p.encode("utf-8"), buf.ctypes.data_as(C.c_void_p), len(buf))
        assert n > 0
        enc.append({"prompt": p, "ids": buf[:n].tolist()})
    dec = []
    rng = np.random.default_rng(7)
    toks = sorted(set([0, 1, 2, 3, 13, 258, 259, 1404, 22110, 29871, 31999, 32000, 32002, 40000]) | set(range(3, 259, 17)) |
                  set(int(x) for x in rng.integers(0, 32000, 80)))
    for tok in toks:
        for prev in (1, 5):
            dec.append([prev, tok, L.ref_tok_decode(t, prev, tok).hex()])
    out = {"vocab_size": 32000, "source": "reference tokenizer.h on /root/reference/tokenizer.bin (oracle/_ref/libref_avx.so)",
           "encode": enc, "decode_hex": dec}
    path = os.path.join(ROOT, "tests", "golden", "tokenizer_pins.json")
    with open(path, "w") as f:
        json.dump(out, f, ensure_ascii=True, indent=0)
    print("wrote", path, len(enc), "prompts,", len(dec), "decodes")


if __name__ == "__main__":
    main()
