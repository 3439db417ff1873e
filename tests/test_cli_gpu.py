"""-m gpu: the command line program end to end on a synthetic full-size q4 checkpoint and a synthetic vocabulary file
(neither real weights nor the reference's tokenizer.bin exist on the GPU box): the ids it prints are those of
tokenizer.encode + model.generate driven from Python."""
import os
import struct
import subprocess

import numpy as np
import pytest

from gpu_common import hip  # noqa: F401
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


def write_vocab(path):
    """a vocabulary file in the reference's format (tokenizer.h:49-86) with the layout its tokenizer assumes: <unk>, <s>,
    </s>, the 256 byte pieces, then printable characters, a few merges with scores, filler"""
    pieces = ["<unk>", "<s>", "</s>"] + [f"<0x{b:02X}>" for b in range(256)]
    singles = [chr(c) for c in range(32, 127)] + ["\n", "\t"]
    merges = ["he", "ll", "lo", "hell", "hello", " w", "or", "ld", " wor", " world", "us", "er", "user", "user\n", " h", " hello"]
    pieces += singles + merges
    while len(pieces) < 32000:
        pieces.append(f"@@{len(pieces)}@@")
    scores = [0.0] * len(pieces)
    for i, m in enumerate(merges):
        scores[pieces.index(m)] = -float(i + 1)
    with open(path, "wb") as f:
        f.write(struct.pack("<i", max(len(p.encode()) for p in pieces)))
        for p, s in zip(pieces, scores):
            b = p.encode()
            f.write(struct.pack("<fi", s, len(b)))
            f.write(b)


def test_cli_greedy_prints_the_ids_of_the_python_path(hip, tmp_path):
    pkg = load_package()
    host = pkg.load_host()
    cfg = host.default_config(4, 3)                       # full-size TinyLlama, q4 weights x q8 activations
    ckpt, vocab = str(tmp_path / "tinyllama.q4.gten"), str(tmp_path / "vocab.bin")
    host.write_gten(cfg, 4242, ckpt)
    write_vocab(vocab)
    n_pred = 48
    r = subprocess.run([pkg.build.HOST_CLI, "-q4", "-greedy", "--ids", "--npred", str(n_pred), "--model", ckpt, "--tokenizer", vocab,
                        "-p", "hello world"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = [int(x) for x in r.stdout.split()]
    tok = host.tokenizer(vocab)
    prompt = tok.encode("hello world")
    assert prompt[:2] == [1, 32001] and prompt[-6:] == [32002, 29871, 13, 32001, 20255, 13]
    cfg2 = host.default_config(4, 3)
    cfg2.max_ctx = n_pred                                  # the CLI builds TinyLlama{n_predict, dtype} (tinyllama.cpp:267)
    m = host.model(cfg2)
    m.load_gten(ckpt)
    want = m.generate(prompt, n_pred, 32002)
    m.close()
    assert got == want[len(prompt):].tolist() and len(got) > 0
    # text mode decodes the same ids
    r2 = subprocess.run([pkg.build.HOST_CLI, "-q4", "-greedy", "--npred", str(n_pred), "--model", ckpt, "--tokenizer", vocab,
                         "-p", "hello world"], capture_output=True, timeout=300)
    assert r2.returncode == 0
    text = b"".join(tok.decode(1 if i == 0 else got[i - 1], t) for i, t in enumerate(got))
    assert r2.stderr.rstrip(b"\n").endswith(text.rstrip(b"\n")) or text in r2.stderr
