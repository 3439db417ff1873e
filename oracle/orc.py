"""ctypes bindings for the CPU checker libraries.  TEST INFRASTRUCTURE ONLY.

`load_oracle()` -> oracle/liborc.so   (own C restatement, prefix orc_)
`load_ref(kind)` -> oracle/_ref/libref_{avx,scalar}.so (the real reference, prefix
ref_), or None when it has not been built (it needs /root/reference at build
time; a prebuilt copy travels with gpurun snapshots).

Both expose the same operator surface, so a `CpuLib` can be used
interchangeably as "the checker".  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

I32, F16, F32, Q8, Q4 = 0, 1, 2, 3, 4
DTYPE_NAMES = {I32: "i32", F16: "f16", F32: "f32", Q8: "q8", Q4: "q4"}


def row_bytes(dtype, cols):
    if dtype in (I32, F32):
        return cols * 4
    if dtype == F16:
        return cols * 2
    if dtype == Q8:
        return ((cols + 31) // 32) * 34
    if dtype == Q4:
        return (cols // 32) * 18
    raise ValueError(dtype)


class Config(C.Structure):
    _fields_ = [(k, C.c_int) for k in
                ("n_vocab", "max_ctx", "n_embd", "n_ffn", "n_layers", "n_heads", "n_kv_heads", "wdtype", "adtype")]


def _p(a):
    """void* of a numpy array (must be C-contiguous)."""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class CpuLib:
    """Uniform wrapper over liborc.so (prefix 'orc_') and libref_*.so (prefix 'ref_')."""

    def __init__(self, path, prefix):
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path)
        L, pre = self.lib, prefix
        vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int

        def sig(name, res, args):
            f = getattr(L, pre + name)
            f.restype = res
            f.argtypes = args
            return f

        self._f2h = sig("fp32_to_fp16", C.c_uint16, [C.c_float])
        self._h2f = sig("fp16_to_fp32", C.c_float, [C.c_uint16])
        self._q8q = sig("q8_quantize_row", None, [vp, vp, ci])
        self._q8d = sig("q8_dequantize_row", None, [vp, vp, ci])
        self._q4d = sig("q4_dequantize_row", None, [vp, vp, ci])
        self._dot = sig("vec_dot", C.c_float, [vp, ci, vp, ci, ci])
        self._embed = sig("token_embed", None, [vp, ci, sz, vp, vp, ci, sz, ci, ci, ci])
        self._matmul = sig("matmul_2d", None, [vp, ci, sz, vp, ci, sz, vp, ci, sz, ci, ci, ci, ci])
        self._rms = sig("rms_norm", None, [vp, ci, sz, vp, vp, sz, ci, ci, ci])
        self._rope = sig("rotary_emb", None, [vp, ci, sz, ci, ci, ci, ci])
        self._silu = sig("silu", None, [vp, vp, ci, sz, ci, ci, ci])
        self._mul = sig("mul", None, [vp, vp, vp, ci, sz, ci, ci, ci])
        self._add = sig("add", None, [vp, vp, vp, ci, sz, ci, ci, ci])
        self._attn = sig("qkv_attn", None, [vp, vp, vp, vp, ci, sz, sz, sz, ci, ci, ci, ci, ci])
        self._mcreate = sig("model_create", vp, [C.POINTER(Config)])
        self._mfree = sig("model_free", None, [vp])
        self._mnw = sig("model_n_weights", ci, [vp])
        self._mwb = sig("model_weight_bytes", sz, [vp, ci])
        self._msetw = sig("model_set_weight", None, [vp, ci, vp, sz])
        self._mlogits = sig("model_logits", None, [vp, vp, ci, ci, vp])
        if prefix == "orc_":
            self._setsimd = sig("set_simd", None, [ci])
            self._wq8 = sig("weight_quantize_q8", None, [vp, ci, ci, vp])
            self._wq4 = sig("weight_quantize_q4", None, [vp, ci, ci, vp])
            self._wf16 = sig("weight_to_f16", None, [vp, sz, vp])
            self._mload = sig("model_load_gten", ci, [vp, C.c_char_p])
        else:
            self.avx = bool(sig("built_with_avx", ci, [])())
            self._tl_create = sig("tl_create", vp, [ci, ci, ci])
            self._tl_free = sig("tl_free", None, [vp])
            self._tl_load = sig("tl_load", ci, [vp, C.c_char_p])
            self._tl_logits = sig("tl_logits", None, [vp, vp, ci, ci, vp])

    # -- scalar helpers
    def set_simd(self, avx_order):
        if self.prefix == "orc_":
            self._setsimd(int(avx_order))

    def fp32_to_fp16(self, f):
        return int(self._f2h(float(f)))

    def fp16_to_fp32(self, h):
        return float(self._h2f(int(h)))

    # -- row codecs: float rows <-> storage bytes (uint8 arrays)
    def quantize_rows(self, x, dtype):
        """f32 [n][d] -> uint8 [n][row_bytes] using the ACTIVATION writer."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n, d = x.shape
        out = np.zeros((n, row_bytes(dtype, d)), dtype=np.uint8)
        for r in range(n):
            if dtype == Q8:
                self._q8q(_p(x[r]), _p(out[r]), d)
            elif dtype == F16:
                out[r] = x[r].astype(np.float16).view(np.uint8)
            elif dtype == F32:
                out[r] = x[r].view(np.uint8)
            else:
                raise ValueError(dtype)
        return out

    def dequantize_rows(self, b, dtype, d):
        b = np.ascontiguousarray(b, dtype=np.uint8)
        n = b.shape[0]
        out = np.zeros((n, d), dtype=np.float32)
        for r in range(n):
            if dtype == Q8:
                self._q8d(_p(b[r]), _p(out[r]), d)
            elif dtype == Q4:
                self._q4d(_p(b[r]), _p(out[r]), d)
            elif dtype == F16:
                out[r] = b[r].view(np.float16).astype(np.float32)
            elif dtype == F32:
                out[r] = b[r].view(np.float32)
            else:
                raise ValueError(dtype)
        return out

    def quantize_weight(self, w, dtype):
        """f32 [rows][cols] -> uint8 [rows][row_bytes] using the OFFLINE converter rules."""
        assert self.prefix == "orc_"
        w = np.ascontiguousarray(w, dtype=np.float32)
        rows, cols = w.shape
        out = np.zeros((rows, row_bytes(dtype, cols)), dtype=np.uint8)
        if dtype == Q8:
            self._wq8(_p(w), rows, cols, _p(out))
        elif dtype == Q4:
            self._wq4(_p(w), rows, cols, _p(out))
        elif dtype == F16:
            self._wf16(_p(w), w.size, _p(out))
        else:
            raise ValueError(dtype)
        return out

    # -- operators (all arrays uint8 [rows][row_bytes], dense)
    def vec_dot(self, a, a_dtype, b, b_dtype, n):
        return float(self._dot(_p(a), a_dtype, _p(b), b_dtype, n))

    def token_embed(self, w, w_dtype, tokens, out, out_dtype, d, start_pos=0):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        self._embed(_p(w), w_dtype, w.shape[1], _p(tokens), _p(out), out_dtype, out.shape[1],
                    len(tokens), d, start_pos)

    def matmul_2d(self, x, x_dtype, w, w_dtype, out, out_dtype, n, d_in, d_out, start_pos=0):
        self._matmul(_p(x), x_dtype, x.shape[1], _p(w), w_dtype, w.shape[1],
                     _p(out), out_dtype, out.shape[1], n, d_in, d_out, start_pos)

    def rms_norm(self, x, dtype, w_f16, out, n, d, start_pos=0):
        self._rms(_p(x), dtype, x.shape[1], _p(w_f16), _p(out), out.shape[1], n, d, start_pos)

    def rotary_emb(self, x, dtype, n, d, d_head, start_pos=0):
        self._rope(_p(x), dtype, x.shape[1], n, d, d_head, start_pos)

    def silu(self, x, out, dtype, n, d, start_pos=0):
        self._silu(_p(x), _p(out), dtype, x.shape[1], n, d, start_pos)

    def mul(self, a, b, out, dtype, n, d, start_pos=0):
        self._mul(_p(a), _p(b), _p(out), dtype, a.shape[1], n, d, start_pos)

    def add(self, a, b, out, dtype, n, d, start_pos=0):
        self._add(_p(a), _p(b), _p(out), dtype, a.shape[1], n, d, start_pos)

    def qkv_attn(self, q, k, v, out, dtype, n, n_heads, n_kv_heads, d_head, start_pos=0):
        self._attn(_p(q), _p(k), _p(v), _p(out), dtype, q.shape[1], k.shape[1], out.shape[1],
                   n, n_heads, n_kv_heads, d_head, start_pos)

    # -- model
    def model(self, cfg):
        return CpuModel(self, cfg)

    def tinyllama(self, n_ctx, wdtype, adtype):
        assert self.prefix == "ref_"
        return RefTinyLlama(self, n_ctx, wdtype, adtype)


class CpuModel:
    def __init__(self, lib, cfg):
        self.lib = lib
        self.cfg = cfg
        self.h = lib._mcreate(C.byref(cfg))

    def n_weights(self):
        return self.lib._mnw(self.h)

    def weight_bytes(self, idx):
        return self.lib._mwb(self.h, idx)

    def set_weight(self, idx, data):
        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        self.lib._msetw(self.h, idx, _p(data), data.size)

    def load_gten(self, path):
        rc = self.lib._mload(self.h, path.encode())
        if rc != 0:
            raise RuntimeError(f"load_gten({path}) failed rc={rc}")

    def logits(self, tokens, start_pos):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.zeros(self.cfg.n_vocab, dtype=np.float32)
        self.lib._mlogits(self.h, _p(tokens), len(tokens), start_pos, _p(out))
        return out

    def partial_row(self, tokens, start_pos, n_blocks):
        """drop-in build only: embedding + the first n_blocks blocks, then a host read of the last activation row"""
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.zeros(1 << 16, np.uint8)
        got = self.lib._mpartial(self.h, _p(tokens), len(tokens), start_pos, n_blocks, _p(out), out.size)
        assert got > 0
        return out[:got].copy()

    def close(self):
        if self.h:
            self.lib._mfree(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RefTinyLlama:
    """The reference's own hard-coded 1.1B TinyLlama class (tinyllama.cpp:23-76)."""
    N_VOCAB = 32003

    def __init__(self, lib, n_ctx, wdtype, adtype):
        self.lib = lib
        self.h = lib._tl_create(n_ctx, wdtype, adtype)

    def load(self, path):
        rc = self.lib._tl_load(self.h, path.encode())
        if rc != 0:
            raise RuntimeError(f"reference load_from_ckpt({path}) failed rc={rc}")

    def logits(self, tokens, start_pos):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.zeros(self.N_VOCAB, dtype=np.float32)
        self.lib._tl_logits(self.h, _p(tokens), len(tokens), start_pos, _p(out))
        return out

    def greedy(self, prompt, max_tokens, eos=-1):
        """drop-in build only: the reference's greedy loop (logits() per token, host argmax) on token ids"""
        buf = np.zeros(max_tokens, np.int32)
        buf[: len(prompt)] = prompt
        total = self.lib._tl_greedy(self.h, _p(buf), len(prompt), max_tokens, eos)
        return buf[:total].copy()

    def close(self):
        if self.h:
            self.lib._tl_free(self.h)
            self.h = None


def build(ref=True):
    """Compile liborc.so (and, where /root/reference exists, oracle/_ref)."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", HERE] + targets, check=True)


_cache = {}


def load_oracle():
    if "orc" not in _cache:
        path = os.path.join(HERE, "liborc.so")
        if not os.path.exists(path):
            build(ref=False)
        _cache["orc"] = CpuLib(path, "orc_")
    return _cache["orc"]


def load_ref(kind="avx"):
    key = "ref_" + kind
    if key not in _cache:
        path = os.path.join(HERE, "_ref", f"libref_{kind}.so")
        _cache[key] = CpuLib(path, "ref_") if os.path.exists(path) else None
    return _cache[key]


class DropinLib:
    """oracle/_ref/libdropin.so: the reference's UNMODIFIED tinyllama.cpp (TinyLlama class,
    .gten loader, module wiring) compiled against this repository's HBM-backed gten API
    and linked with libgten_hip.so (oracle/Makefile, target `dropin`).  Needs a GPU to run."""

    def __init__(self, path):
        self.lib = L = C.CDLL(path)
        vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int

        def sig(name, res, args):
            f = getattr(L, "ref_" + name)
            f.restype = res
            f.argtypes = args
            return f

        self._mcreate = sig("model_create", vp, [C.POINTER(Config)])
        self._mfree = sig("model_free", None, [vp])
        self._mnw = sig("model_n_weights", ci, [vp])
        self._mwb = sig("model_weight_bytes", sz, [vp, ci])
        self._msetw = sig("model_set_weight", None, [vp, ci, vp, sz])
        self._mlogits = sig("model_logits", None, [vp, vp, ci, ci, vp])
        self._tl_create = sig("tl_create", vp, [ci, ci, ci])
        self._tl_free = sig("tl_free", None, [vp])
        self._tl_load = sig("tl_load", ci, [vp, C.c_char_p])
        self._tl_logits = sig("tl_logits", None, [vp, vp, ci, ci, vp])
        self._tl_greedy = sig("tl_greedy", ci, [vp, vp, ci, ci, ci])
        self._mpartial = sig("model_partial_row", sz, [vp, vp, ci, ci, ci, vp, sz])
        self._set_fused = sig("set_fused_rows", None, [ci])
        self.prefix = "ref_"

    def set_fused_rows(self, on):
        """single-row recording of gten/modules.h (fused decoder behind the unmodified module calls) on / off"""
        self._set_fused(1 if on else 0)

    def model(self, cfg):
        return CpuModel(self, cfg)

    def tinyllama(self, n_ctx, wdtype, adtype):
        return RefTinyLlama(self, n_ctx, wdtype, adtype)


def build_dropin():
    subprocess.run(["make", "-s", "-C", HERE, "dropin"], check=True)


def load_dropin():
    if "dropin" not in _cache:
        path = os.path.join(HERE, "_ref", "libdropin.so")
        _cache["dropin"] = DropinLib(path) if os.path.exists(path) else None
    return _cache["dropin"]
