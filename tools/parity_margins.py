"""How far the HIP paths sit from the reference at long context, in numbers (run on the GPU box:
    gpurun -- python3 tools/parity_margins.py > gpurun_out/parity_margins.txt).

TinyLlama-1.1B q4 / q8 / f16 on the seeded synthetic weights, stepped from n = 1 with the fixtures' teacher-forced ids to the
probe positions of tests/golden/full_model_golden.npz (q4) and full_extra_golden.npz (q8, f16): n = 257, 1024, 2047, 2048 --
every attention chunk boundary, the last one the BASELINE.json metric point.  At each probe the logits of
  * fused        the single-sequence decoder, default forms (one-launch attention with chunk-local statistics),
  * exact        the same with gten_hip_set_decode_exact(1) (two launches, whole-row statistics),
  * operators    the operator path, row by row (gten_hip_* operators, gten/modules.h without the fused rows),
  * wide-64      a 64-sequence decoder (matrix-core W.x and attention), slots 0 / 31 / 63,
  * wide-128     one lane of 128 rows, slots 0 / 31 / 127
are compared with the reference's AVX build on the fixture's ids (its top logits + the fixed probe ids), beside the
reference's OWN spread on the same ids -- its AVX build against its scalar build -- which is the yardstick of the three
long-context tests (bar: 1.35 x that spread, max <= 0.5; f16: max |dlogit| <= 0.03).  The product path only; the
fixtures are data (tests/golden/make_golden.py made them from the reference).

    python3 tools/parity_margins.py [--configs q4,q8,f16] [--paths fused,exact,operators,wide-64,wide-128]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
from helpers import MODES  # noqa: E402

ALL_PATHS = ("fused", "exact", "operators", "wide-64", "wide-128")


def main():
    # every walk builds its own model (the exact / fast choice is made when a decoder is created): generate each configuration's
    # synthetic weights ONCE and let the other walks read them back (host/tinyllama_model.h load_synthetic_cached), 45 s -> 1 s each
    cache = None
    if os.path.isdir("/dev/shm") and "GTEN_SYNTH_CACHE_DIR" not in os.environ:
        cache = os.environ["GTEN_SYNTH_CACHE_DIR"] = "/dev/shm"
    try:
        run()
    finally:
        if cache:
            import glob
            for f in glob.glob(os.path.join(cache, "gten_synth_*")):
                try:
                    os.remove(f)
                except OSError:
                    pass


def run():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="q4,q8,f16")
    ap.add_argument("--paths", default=",".join(ALL_PATHS))
    args = ap.parse_args()
    want_cfg, want_paths = args.configs.split(","), args.paths.split(",")
    g4 = np.load(os.path.join(ROOT, "tests", "golden", "full_model_golden.npz"))
    gx = np.load(os.path.join(ROOT, "tests", "golden", "full_extra_golden.npz"))
    pkg = load_package()
    hip = pkg.hipabi.load(0)
    host = pkg.load_host()
    worst = {}
    for name, wd, ad in MODES():
        if name not in want_cfg:
            continue
        g = g4 if name == "q4" else gx
        if f"long.{name}.ns" not in g:
            print(f"{name}: no long-context probe in the fixture")
            continue
        cfg = host.default_config(wd, ad)
        toks = host.synthetic_tokens(2048, seed=int(g["token_seed"][0]))
        ns = [int(n) for n in g[f"long.{name}.ns"]]
        probe = g["probe_ids"]
        seed = int(g["seed"][0])

        def walk_single(kind):
            hip.set_decode_exact(kind == "exact")
            m = host.model(cfg)
            m.load_synthetic(seed)
            out = {}
            if kind == "operators":
                m.set_fast_decode(False)
                m.logits(toks[:1], 0, want=False)
                for n in range(2, ns[-1] + 1):
                    lg = m.logits(toks[:n], n - 1, want=n in ns)
                    if n in ns:
                        out[n] = [lg]
            else:
                m.decode_begin(toks)
                prev = 1
                for n in ns:
                    for k in range(prev, n):
                        m.decode_step(k, True)
                    out[n] = [m.logits(toks[:n], n - 1)]
                    prev = n + 1
            m.close()
            hip.set_decode_exact(False)
            return out

        def walk_wide(S):
            batch = host.batch(cfg, S)
            batch.load_synthetic(seed)
            for q in range(S):
                batch.decode_begin(q, toks)
            out = {}
            for n in range(1, ns[-1] + 1):
                batch.decode_step(n, True)
                if n in ns:
                    out[n] = [batch.logits(q) for q in (0, 31, S - 1)]
            batch.close()
            return out

        runs = {}
        for p in ALL_PATHS:
            if p not in want_paths:
                continue
            t0 = time.time()
            runs[p] = walk_wide(int(p.split("-")[1])) if p.startswith("wide") else walk_single(p)
            print(f"# {name} {p}: walked in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
        print(f"TinyLlama-1.1B {name}, synthetic weights (seed {seed}), teacher-forced ids; logits on the fixture's top ids + {len(probe)} probe ids")
        print("rms / max of (path - reference AVX build); 'own' = reference AVX build - reference scalar build on the same ids")
        for n in ns:
            ids = g[f"long.{name}.n{n}.top_ids"]
            ref = np.concatenate([g[f"long.{name}.n{n}.top_logits"], g[f"long.{name}.n{n}.probes"]])
            own = g[f"long.{name}.n{n}.probes"] - g[f"long.{name}.n{n}.probes.scalar"]
            own_rms, own_max = float(np.sqrt((own * own).mean())), float(np.abs(own).max())
            gap = float(g[f"long.{name}.n{n}.top_logits"][0] - g[f"long.{name}.n{n}.top_logits"][1])
            print("n = %4d   own: rms %.4f max %.4f   top-1: avx %d scalar %d (avx gap %.3f)"
                  % (n, own_rms, own_max, int(ids[0]), int(g[f"long.{name}.n{n}.top_ids.scalar"][0]), gap))
            for k, out in runs.items():
                for i, lg in enumerate(out[n]):
                    d = np.concatenate([lg[ids], lg[probe]]) - ref
                    rms, mx = float(np.sqrt((d * d).mean())), float(np.abs(d).max())
                    slot = "" if len(out[n]) == 1 else " slot %d" % i
                    print("    %-20s rms %.4f (%.2f x own)  max %.4f  top-1 %d" % (k + slot, rms, rms / own_rms, mx, int(np.argmax(lg))))
                    key = (name, k)
                    w = worst.get(key, (0.0, 0.0, 0.0))
                    worst[key] = (max(w[0], rms), max(w[1], rms / own_rms), max(w[2], mx))
            if all(k in runs for k in ("fused", "exact", "operators")):
                a, b, c = runs["fused"][n][0], runs["exact"][n][0], runs["operators"][n][0]
                dd = lambda x, y: (float(np.sqrt(((x - y) ** 2).mean())), float(np.abs(x - y).max()))  # noqa: E731
                print("    between the paths (all 32003 logits): fused - operators rms %.4f max %.4f | exact - operators rms %.4f max %.4f | fused - exact rms %.4f max %.4f"
                      % (dd(a, c) + dd(b, c) + dd(a, b)))
        print()
    print("worst over the probes, per configuration and path: rms | x the reference's own spread | max")
    for (name, k), (rms, x, mx) in worst.items():
        print("    %-4s %-10s rms %.4f  %.2f x own  max %.4f%s" % (name, k, rms, x, mx, "   <-- above 1.25 x" if x > 1.25 and name != "f16" else ""))


if __name__ == "__main__":
    main()
