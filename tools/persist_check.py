"""first contact of the persistent step with the hardware: same ids / logits as the launch chain, abort word clean, timing"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
pkg = load_package()
host = pkg.load_host()
hip = host.hip
hip.init(0)
from helpers import Q4, Q8
cfg = host.default_config(Q4, Q8)
N = int(os.environ.get("PC_N", "300"))
toks = host.synthetic_tokens(2049, seed=12345, n_vocab=cfg.n_vocab)
models = {}
for name, on in (("persist", True), ("chain", False)):
    hip.set_decode_persistent(on)
    m = host.model(cfg)
    m.load_synthetic(1234)
    m.decode_begin(toks[:2048])
    models[name] = m
hip.set_decode_persistent(True)
print("status before:", hip.persist_status()[:3], flush=True)
watch = [1, 2, 3, 17, 255, 256, 257, 258, N - 1, N]
res = {k: {} for k in models}
for name, m in models.items():
    t0 = time.time()
    for n in range(1, N + 1):
        m.decode_step(n, n > 4)            # a few eager launches first, then graph replays
        if n in watch:
            res[name][n] = m.decode_result(n)
    print(name, "ids", res[name], "%.2f s" % (time.time() - t0), flush=True)
    if name == "persist":
        print("status:", hip.persist_status()[:3], flush=True)
ok = res["persist"] == res["chain"]
la = models["persist"].logits(toks[:N], N - 1)
lb = models["chain"].logits(toks[:N], N - 1)
print("ids equal:", ok, " logits equal:", bool(np.array_equal(la, lb)), " max |d|:", float(np.abs(la - lb).max()), flush=True)
print("status:", hip.persist_status()[:3], flush=True)
# timing at the end of the context
for name, m in models.items():
    m.decode_steps(N + 1, 1984 - N, True)
    m.decode_result(1984)
    ts = []
    for rep in range(3):
        m.decode_steps(1985, 64, True) if rep == 0 else m.decode_steps(1985, 64, True)
        t0 = time.time(); m.decode_steps(1985, 64, True); m.decode_result(2048); ts.append((time.time() - t0) / 64 * 1e3)
    print(name, "ms per step at n -> 2048:", ["%.4f" % t for t in ts], " id(2048) =", m.decode_result(2048), flush=True)
nd, nl, ab, st = hip.persist_status(20 * 22 + 8)
print("status:", nd, nl, ab)
if any(st):
    st = np.array(st, dtype=np.int64)
    order = [0, 1, 2, 3, 4, 5, 6, 7, 18, 19, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 20]
    names = ["E-pub->A-poll", "issue o,down", "norm+stage", "A dots", "A publish", "q-poll+headprep", "K row, scores", "max, exp, sum (2 barriers)",
             "p, barrier, p.V", "barrier, 4-sum, publish", "B2 poll", "B2 join+publish", "C-poll", "C stage+dot+publish", "D-poll", "D prologue+dots",
             "D epilogue+publish", "E-poll", "E scatter+dots", "E publish"]
    L = 22
    t = st[: 20 * L + 1].astype(np.float64) * 0.01
    per = np.zeros((L, 20))
    for l in range(L):
        seg = np.array([t[20 * l + k] for k in order])
        per[l] = np.diff(seg)
    m = per[1:].mean(0)
    for k in range(20):
        print("  %-28s %6.2f us" % (names[k], m[k]))
    print("block us:", per[1:].sum(1).mean(), " step us:", (st[20 * L + 2] - st[0]) * 0.01, " tail (final norm+lm_head+argmax) us:", (st[20 * L + 2] - st[20 * L]) * 0.01)
for m_ in models.values():
    m_.close()
