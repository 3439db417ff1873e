"""Body of __graft_entry__.smoke(): one small invocation of the hot path on
cuda:0, checked against the oracle (the oracle is only the checker here)."""
import numpy as np

from helpers import F32, Q4, Q8, act_rows, compare_rows, rng, row_bytes, weight_rows
from oracle import orc


def run_smoke(pkg):
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
        compare_rows(out.download(shape=want.shape), want, od, d_out, "smoke q8.q4 matmul")
    print("smoke ok: q4 W.x on cuda:0 matches the oracle")
