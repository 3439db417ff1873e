"""Body of __graft_entry__.smoke(): one small invocation of the hot path on cuda:0 --
a q4 W.x through the operator C-ABI and a few greedy decode steps of a small q4 model
through the C++ driver's fused decode path -- checked against the oracle (the oracle
is only the checker here)."""
import numpy as np

from helpers import F32, Q4, Q8, act_rows, compare_rows, rng, row_bytes, tiny_config, weight_rows
from oracle import orc


def run_smoke(pkg):
    pkg.build.build_all()
    api = pkg.hipabi.load()
    api.init(0)
    oracle = orc.load_oracle()
    r = rng(2024)
    d_in, d_out = 2048, 256
    x, _ = act_rows(oracle, r, 1, d_in, Q8)
    w, _ = weight_rows(oracle, r, d_out, d_in, Q4)
    for od in (Q8, F32):
        want = np.zeros((1, row_bytes(od, d_out)), np.uint8)
        oracle.matmul_2d(x, Q8, w, Q4, want, od, 1, d_in, d_out, 0)
        out = api.alloc(want.nbytes)
        api.matmul_2d(api.upload(x), Q8, api.upload_weight(w, Q4, d_out, d_in), Q4, out, od, 1, d_in, d_out, 0)
        compare_rows(out.download(shape=want.shape), want, od, d_out, "smoke q8.q4 matmul", atol=8e-6)

    host = pkg.load_host()
    ocfg = tiny_config(Q4, Q8, n_heads=4, n_kv_heads=2)
    cfg = pkg.HostConfig(**{k: getattr(ocfg, k) for k, _ in ocfg._fields_})
    gm, om = host.model(cfg), oracle.model(ocfg)
    for i in range(gm.n_weights()):
        wt = host.synth_weight(cfg, 4321, i)
        gm.set_weight(i, wt)
        om.set_weight(i, wt)
    toks = list(host.synthetic_tokens(6, seed=7, n_vocab=cfg.n_vocab))
    worst = 0.0
    for step in range(5):
        sp = 0 if step == 0 else len(toks) - 1          # step 0: prefill (operators); then the fused decode path
        want = om.logits(toks, sp)
        got = gm.logits(toks, sp)
        d = got - want
        rms = float(np.sqrt((d * d).mean()))
        worst = max(worst, rms)
        assert np.isfinite(got).all() and rms <= 0.10 and float(np.abs(d).max()) <= 0.5, (step, rms)
        toks.append(int(np.argmax(want)))
    gm.close(); om.close()
    print(f"smoke ok: q4 W.x and 4 fused decode steps on cuda:0 match the oracle (worst logit rms {worst:.2e})")
