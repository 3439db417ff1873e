"""How far the HIP paths sit from the reference at long context, in numbers (run on the GPU box:
    gpurun -- python3 tools/parity_margins.py > gpurun_out/parity_margins.txt).

TinyLlama-1.1B q4 on the seeded synthetic weights, stepped from n = 1 with the fixture's teacher-forced ids to the probe
positions of tests/golden/full_model_golden.npz (n = 257, 1024, 2047, 2048: every attention chunk boundary, the last one
the BASELINE.json metric point).  At each probe the logits of
  * the fused decoder (default forms: one-launch attention with chunk-local statistics),
  * the fused decoder with gten_hip_set_decode_exact(1) (two launches, whole-row statistics),
  * the operator path, row by row (gten_hip_* operators, gten/modules.h without the fused rows)
are compared with the reference's AVX build on the fixture's ids (its top logits + the fixed probe ids), beside the
reference's OWN spread on the same ids -- its AVX build against its scalar build -- which is the yardstick of
tests/test_golden_gpu.py::test_long_context_probe_q4 (bar: 1.35 x that spread, max <= 0.5).  The product path only;
the fixture is data (tests/golden/make_golden.py made it from the reference)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
from helpers import Q4, Q8  # noqa: E402


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "full_model_golden.npz"))
    pkg = load_package()
    hip = pkg.hipabi.load(0)
    host = pkg.load_host()
    cfg = host.default_config(Q4, Q8)
    toks = host.synthetic_tokens(2048, seed=int(g["token_seed"][0]))
    ns = [int(n) for n in g["long.q4.ns"]]
    probe = g["probe_ids"]

    def walk(kind):
        hip.set_decode_exact(kind == "fused, exact forms")
        m = host.model(cfg)
        m.load_synthetic(int(g["seed"][0]))
        out = {}
        if kind == "operators":
            m.set_fast_decode(False)
            m.logits(toks[:1], 0, want=False)
            for n in range(2, ns[-1] + 1):
                lg = m.logits(toks[:n], n - 1, want=n in ns)
                if n in ns:
                    out[n] = lg
        else:
            m.decode_begin(toks)
            prev = 1
            for n in ns:
                for k in range(prev, n):
                    m.decode_step(k, True)
                out[n] = m.logits(toks[:n], n - 1)
                prev = n + 1
        m.close()
        hip.set_decode_exact(False)
        return out

    runs = {k: walk(k) for k in ("fused", "fused, exact forms", "operators")}
    print("TinyLlama-1.1B q4, synthetic weights (seed %d), teacher-forced ids; logits on the fixture's top ids + %d probe ids" % (int(g["seed"][0]), len(probe)))
    print("rms / max of (path - reference AVX build); 'own' = reference AVX build - reference scalar build on the same ids")
    for n in ns:
        ids = g[f"long.q4.n{n}.top_ids"]
        ref = np.concatenate([g[f"long.q4.n{n}.top_logits"], g[f"long.q4.n{n}.probes"]])
        own = g[f"long.q4.n{n}.probes"] - g[f"long.q4.n{n}.probes.scalar"]
        own_rms, own_max = float(np.sqrt((own * own).mean())), float(np.abs(own).max())
        print("n = %4d   own: rms %.4f max %.4f   top-1: avx %d scalar %d" % (n, own_rms, own_max, int(ids[0]), int(g[f"long.q4.n{n}.top_ids.scalar"][0])))
        for k, out in runs.items():
            lg = out[n]
            d = np.concatenate([lg[ids], lg[probe]]) - ref
            rms, mx = float(np.sqrt((d * d).mean())), float(np.abs(d).max())
            print("    %-20s rms %.4f (%.2f x own)  max %.4f  top-1 %d" % (k, rms, rms / own_rms, mx, int(np.argmax(lg))))
        a, b, c = runs["fused"][n], runs["fused, exact forms"][n], runs["operators"][n]
        dd = lambda x, y: (float(np.sqrt(((x - y) ** 2).mean())), float(np.abs(x - y).max()))
        print("    between the paths (all 32003 logits): fused - operators rms %.4f max %.4f | exact - operators rms %.4f max %.4f | fused - exact rms %.4f max %.4f"
              % (dd(a, c) + dd(b, c) + dd(a, b)))


if __name__ == "__main__":
    main()
