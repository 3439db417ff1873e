"""gpu: results must not depend on what ANOTHER stream is running on the chip.  (Named to run last: full-size models.)

Round 2 found that they did: beside a kernel that keeps every SIMD's matrix core busy (a prompt's GEMMs on stream 1, or the
register-only MFMA loop of tools/mfma_neighbour/neighbour.hip), the fused decoder's q|k|v and lm_head launches scaled their
outputs by +0.3..0.5 % in about a quarter of their workgroups -- packed-f32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32,
made by hipcc's SLP vectorizer) returned an older value in the last 16 lanes of a wave.  Alone on the GPU every run was
bit-identical, so no parity test saw it; continuous batching (prompts beside the shared steps) did, as ids that changed from
run to run.  build.py now keeps those instructions out of every kernel (tests/test_no_packed_f32_cpu.py); this test is the
run-time half: one decode step, repeated beside the neighbour, must give the bits of the quiet step.
Full-size TinyLlama shapes on purpose: the launches must be long enough to overlap the neighbour on most CUs."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from gpu_common import hip  # noqa: F401

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB_DIR = os.path.join(ROOT, "tools", "mfma_neighbour")
NB_SRC = os.path.join(NB_DIR, "neighbour.hip")
NB_LIB = os.path.join(NB_DIR, "libneighbour.so")


@pytest.fixture(scope="module")
def neighbour(hip):  # noqa: F811
    if not os.path.exists(NB_LIB) or os.path.getmtime(NB_LIB) < os.path.getmtime(NB_SRC):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            pytest.skip("no hipcc to build the neighbour kernels")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", NB_LIB, NB_SRC], capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            pytest.skip("neighbour kernels did not build: " + r.stderr[-500:])
    nb = ctypes.CDLL(NB_LIB)
    for f in (nb.nb_init, nb.nb_sync):
        f.restype = ctypes.c_int
    nb.nb_run.restype = ctypes.c_int
    nb.nb_run.argtypes = [ctypes.c_int] * 4
    assert nb.nb_init() == 0
    return nb


@pytest.fixture(scope="module")
def host_api(hip):  # noqa: F811
    from __graft_entry__ import load_package
    return load_package().load_host()


MFMA_F16, MFMA_I8 = 1, 6


def beside(nb, kind, fn, launches=12):
    assert nb.nb_run(kind, launches, 512, 40) == 0          # ~1 ms of 512 x 256-thread workgroups: every SIMD busy
    out = fn()
    assert nb.nb_sync() == 0
    return out


@pytest.mark.parametrize("wdtype,adtype", [(4, 3), (3, 3), (1, 1)], ids=["q4", "q8", "f16"])
def test_fused_decode_step_beside_a_matrix_core_neighbour(hip, host_api, neighbour, wdtype, adtype):  # noqa: F811
    cfg = host_api.default_config(wdtype, adtype)
    m = host_api.model(cfg)
    m.load_synthetic(1234)
    toks = host_api.synthetic_tokens(400, seed=1000)
    m.logits(toks[:299], 0, want=False)

    def step():
        return m.logits(toks[:300], 299)

    quiet = step()
    assert np.array_equal(step(), quiet)
    for kind in (MFMA_F16, MFMA_I8):
        for _ in range(6):
            got = beside(neighbour, kind, step)
            assert np.array_equal(got, quiet), f"decode step beside neighbour kind {kind}: max |d| {np.abs(got - quiet).max():.3g}"
    assert np.array_equal(step(), quiet)


@pytest.mark.parametrize("n_seq,wdtype,adtype", [(8, 4, 3), (16, 4, 3), (64, 4, 3), (32, 1, 1)], ids=["q4-8", "q4-16", "q4-64", "f16-32"])
def test_shared_step_beside_a_matrix_core_neighbour(hip, host_api, neighbour, n_seq, wdtype, adtype):  # noqa: F811
    """GEMV kernels (8), k_dec_mmvh + k_dec_attn_mm_g (16, 64), k_dec_mmv_f16 + the f16 grouped attention (f16, 32)"""
    cfg = host_api.default_config(wdtype, adtype)
    b = host_api.batch(cfg, n_seq)
    b.load_synthetic(1234)
    toks = host_api.synthetic_tokens(400, seed=1000)
    for q in range(n_seq):
        b.prefill(q, toks[:199], want=False)
    for q in range(n_seq):
        b.decode_begin(q, toks)

    def step():
        b.decode_steps(200, 1, True)
        hip.sync()
        return np.stack([b.logits(q) for q in range(n_seq)])

    quiet = step()
    assert np.array_equal(step(), quiet)
    for kind in (MFMA_F16, MFMA_I8):
        for _ in range(3):
            got = beside(neighbour, kind, step, launches=30)
            assert np.array_equal(got, quiet), f"{int((got != quiet).any(axis=1).sum())} of {n_seq} sequences changed beside neighbour kind {kind}"


@pytest.mark.parametrize("wdtype,adtype", [(4, 3), (3, 3), (1, 1)], ids=["q4", "q8", "f16"])
def test_prompt_kernels_beside_a_matrix_core_neighbour(hip, host_api, neighbour, wdtype, adtype):  # noqa: F811
    cfg = host_api.default_config(wdtype, adtype)
    m = host_api.model(cfg)
    m.load_synthetic(1234)
    toks = host_api.synthetic_tokens(300, seed=7)
    quiet = m.logits(toks[:256], 0)
    for kind in (MFMA_F16, MFMA_I8):
        for _ in range(3):
            got = beside(neighbour, kind, lambda: m.logits(toks[:256], 0), launches=40)
            assert np.array_equal(got, quiet), f"prompt beside neighbour kind {kind}"


def test_decode_step_beside_the_librarys_own_prompt_kernels(hip, host_api):  # noqa: F811
    """the real serving case as the neighbour: a 512-id q4 prompt (f16 MFMA W.x, int8 MFMA scores) on the library's second
    stream while the fused decode step of ANOTHER model object runs on stream 0"""
    cfg = host_api.default_config(4, 3)
    dec, pre = host_api.model(cfg), host_api.model(cfg)
    dec.load_synthetic(1234)
    pre.load_synthetic(1234)
    toks = host_api.synthetic_tokens(600, seed=1000)
    dec.logits(toks[:299], 0, want=False)

    def step():
        return dec.logits(toks[:300], 299)

    quiet = step()
    assert np.array_equal(step(), quiet)
    for _ in range(6):
        hip.select_stream(1)
        pre.logits(toks[:512], 0, want=False)           # asynchronous: queued on stream 1
        hip.select_stream(0)
        got = step()
        hip.select_stream(1); hip.sync(); hip.select_stream(0)
        assert np.array_equal(got, quiet), f"max |d| {np.abs(got - quiet).max():.3g}"
    dec.close(); pre.close()


def test_standalone_reproducer(hip, tmp_path):  # noqa: F811
    """tools/mfma_neighbour/repro.hip, built and run here: the scalar build of the victim and every packed f16 form the
    library contains must be clean beside both neighbours.  (The packed-f32 forms with src1's halves swapped fail on the
    MI355X boxes of rounds 2 and 3 -- reported, not asserted: a later chip or firmware may well fix them.)"""
    script = os.path.join(NB_DIR, "run_repro.sh")
    if not os.path.exists(shutil.which("hipcc") or "/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this box")
    r = subprocess.run(["bash", script, "12"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.splitlines()
    scalar = [l for l in lines if l.startswith("victim") and "scalar" in l]
    assert len(scalar) >= 8 and all(" 0 of " in l for l in scalar), "\n".join(scalar)
    forms = [l for l in lines if l.strip().startswith("by form")]
    for l in forms:                                         # [f16 forms 0..3 | f32 forms 4..7]
        counts = [int(x) for x in l[l.rindex("[") + 1:l.rindex("]")].split()]
        assert counts[:4] == [0, 0, 0, 0], l
    print("\n".join(l for l in lines if "differ" in l and " 0 of " not in l))
