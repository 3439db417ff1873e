"""Experiment (round 2, profiles/README.md "packed f32 beside MFMA"): the fused single-sequence decode step beside each kind of neighbour -- MFMA f16 / i8, one-wave MFMA, dot4, plain VALU (r02_packed_f32/d11.txt, d17.txt).
Run on the GPU box from the repository root:  python tools/mfma_neighbour/decode_beside_neighbours.py  (builds libneighbour.so when missing)."""
import os, sys, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from __graft_entry__ import load_package
pkg = load_package(); hip = pkg.hipabi.load(); hip.init(0)
host = pkg.load_host()
from neighbour import load_neighbour
bg = load_neighbour()
TR = int(sys.argv[1]) if len(sys.argv) > 1 else 20
TAG = os.environ.get("TAG", "")
cfg = host.default_config(4, 3)
m = host.model(cfg); m.load_synthetic(1234)
toks = host.synthetic_tokens(2048, seed=1000)
m.logits(toks[:299], 0, want=False)
def one(): return m.logits(toks[:300], 299)
ref = one()
names = {1: "mfma f16 512x256", 7: "mfma f16 512x64 (1 wave)", 6: "mfma i8", 5: "dot4 spin", 3: "valu spin"}
for kind, grid in ((1, 512), (1, 64), (7, 512), (6, 512), (5, 512), (3, 512)):
    bad = 0; mx = 0.0; nd = 0
    for t in range(TR):
        assert bg.nb_run(kind, 12, grid, 40) == 0
        got = one()
        assert bg.nb_sync() == 0
        if not np.array_equal(got, ref):
            bad += 1; mx = max(mx, float(np.abs(got - ref).max())); nd = max(nd, int((got != ref).sum()))
    print(f"{TAG:10s} fused decoder beside {names[kind]:26s} grid {grid:4d}: {bad} of {TR} differ, max |d| {mx:.3g}, up to {nd} logits", flush=True)
assert np.array_equal(one(), ref)
