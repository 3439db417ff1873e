"""Experiment (round 2, profiles/README.md "packed f32 beside MFMA"): every operator of the C-ABI, the fused decoder and the small shared steps, each repeated beside the register-only MFMA neighbour and compared with its quiet run (r02_packed_f32/d10.txt).
Run on the GPU box from the repository root:  python tools/mfma_neighbour/ops_beside_neighbour.py  (builds libneighbour.so when missing)."""
import os, sys, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from __graft_entry__ import load_package
pkg = load_package(); hip = pkg.hipabi.load(); hip.init(0)
host = pkg.load_host()
from neighbour import load_neighbour
bg = load_neighbour()
I32, F16, F32, Q8, Q4 = 0, 1, 2, 3, 4
rng = np.random.default_rng(0)
def qblocks(rows, cols, dt):
    nb = cols // 32; per = 18 if dt == Q4 else 34
    b = rng.integers(0, 256, (rows, nb, per), dtype=np.uint8)
    d = np.float16(0.01).view(np.uint16); b[:, :, 0] = d & 0xff; b[:, :, 1] = d >> 8
    return b.reshape(rows, nb * per)
TR = int(sys.argv[1]) if len(sys.argv) > 1 else 40
KIND = int(os.environ.get("BGKIND", "1"))
def trial(name, fn, out, prep=None):
    if prep: prep()
    fn(); hip.sync(); ref = out.download().copy()
    bad = 0
    for t in range(TR):
        if prep: prep()
        assert bg.nb_run(KIND, 12, 512, 40) == 0
        fn(); hip.sync()
        got = out.download()
        assert bg.nb_sync() == 0
        bad += int(not np.array_equal(got, ref))
    if prep: prep()
    fn(); hip.sync(); quiet_ok = np.array_equal(out.download(), ref)
    print(f"{name:34s}: {bad} of {TR} runs beside the mfma neighbour differ; quiet again equal: {quiet_ok}", flush=True)
D, F, NR = 2048, 5632, 16
w4 = hip.upload_weight(qblocks(D, D, Q4), Q4, D, D)
x1 = hip.upload(qblocks(64, D, Q8)); y1 = hip.alloc(64 * hip.row_bytes(Q8, D))
trial("matmul rows kernel (1 row, q4)", lambda: hip.matmul_2d(x1, Q8, w4, Q4, y1, Q8, 1, D, D), y1)
trial("matmul rows kernel (8 rows, q4)", lambda: hip.matmul_2d(x1, Q8, w4, Q4, y1, Q8, 8, D, D), y1)
trial("matmul mfma (64 rows, q4)", lambda: hip.matmul_2d(x1, Q8, w4, Q4, y1, Q8, 64, D, D), y1)
nw = hip.upload((rng.standard_normal(D) * 0.1 + 1).astype(np.float16))
xa = qblocks(NR, D, Q8); xn = hip.upload(xa); yn = hip.alloc(NR * hip.row_bytes(Q8, D))
trial("rms_norm (16 rows)", lambda: hip.rms_norm(xn, Q8, nw, yn, NR, D), yn)
xr = hip.upload(xa)
trial("rotary_emb (16 rows)", lambda: hip.rotary_emb(xr, Q8, NR, D, 64, 5), xr, prep=lambda: xr.upload(xa))
fa = hip.upload(qblocks(NR, F, Q8)); fb = hip.upload(qblocks(NR, F, Q8)); fo = hip.alloc(NR * hip.row_bytes(Q8, F))
trial("silu (16 rows)", lambda: hip.silu(fa, fo, Q8, NR, F), fo)
trial("mul (16 rows)", lambda: hip.mul(fa, fb, fo, Q8, NR, F), fo)
trial("add (16 rows)", lambda: hip.add(fa, fb, fo, Q8, NR, F), fo)
NC = 300
q = hip.upload(qblocks(NC, D, Q8)); k = hip.upload(qblocks(NC, 256, Q8)); v = hip.upload(qblocks(NC, 256, Q8)); o = hip.alloc(NC * hip.row_bytes(Q8, D))
trial("qkv_attn (1 new row at 299)", lambda: hip.qkv_attn(q, k, v, o, Q8, NC, 32, 4, 64, NC - 1), o)
trial("qkv_attn tiled (300 rows)", lambda: hip.qkv_attn(q, k, v, o, Q8, NC, 32, 4, 64, 0), o)
# fused single-sequence decoder and the small shared step
cfg = host.default_config(4, 3)
m = host.model(cfg); m.load_synthetic(1234)
toks = host.synthetic_tokens(2048, seed=1000)
m.logits(toks[:299], 0, want=False)
class L:
    def __init__(s): s.v = None
    def download(s): return s.v
lo = L()
def one(): lo.v = m.logits(toks[:300], 299)
trial("fused decoder, 1 sequence", one, lo)
m.set_fast_decode(False)
trial("operator path, 1 new row", one, lo)
for S in (2, 8):
    b = host.batch(cfg, S); b.load_synthetic(1234)
    for s_ in range(S): b.prefill(s_, toks[:299], want=False)
    for s_ in range(S): b.decode_begin(s_, toks)
    def stp():
        b.decode_steps(300, 1, True); hip.sync(); lo.v = np.stack([b.logits(s_) for s_ in range(S)])
    trial(f"shared step, {S} sequences", stp, lo)
