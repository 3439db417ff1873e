"""ctypes binding of include/gten_hip.h (libgten_hip.so).

This is plumbing for tests and bench.py: device buffers are plain HBM
allocations owned by the library, moved with explicit h2d/d2h copies.  There
is no CPU fallback: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

I32, F16, F32, Q8, Q4 = 0, 1, 2, 3, 4


class GtenHipError(RuntimeError):
    pass


def _sig(lib, name, res, args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = args
    return f


class DeviceBuffer:
    """A byte range in HBM.  `.ptr` is the raw device address (int)."""

    def __init__(self, api, nbytes):
        self.api = api
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        api._check(api._malloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, api, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(api, arr.nbytes)
        buf.upload(arr)
        return buf

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        self.api._check(self.api._h2d(self.ptr + offset, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def download(self, dtype=np.uint8, shape=None, offset=0, nbytes=None):
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(nbytes, dtype=np.uint8)
        self.api._check(self.api._d2h(out.ctypes.data_as(C.c_void_p), self.ptr + offset, nbytes))
        out = out.view(dtype)
        return out.reshape(shape) if shape is not None else out

    def zero(self, byte=0):
        self.api._check(self.api._memset(self.ptr, byte, self.nbytes))

    def free(self):
        if self.ptr:
            self.api._check(self.api._free(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class BlockDesc(C.Structure):
    """gten_hip_block_desc (include/gten_hip.h)"""
    _fields_ = ([(k, C.c_int) for k in ("adtype", "wdtype", "n_embd", "n_heads", "n_kv_heads", "n_ffn")] +
                [(k, C.c_void_p) for k in ("attn_norm_w", "wq", "wk", "wv", "wo", "ffn_norm_w", "wgate", "wup", "wdown", "inp",
                                           "attn_norm_out", "q", "k", "v", "attn_out", "o", "h", "ffn_norm_out", "gate", "up", "down", "out")])


class GtenHip:
    """Loaded libgten_hip.so with every symbol of include/gten_hip.h bound."""

    SYMBOLS = [
        "gten_hip_device_count", "gten_hip_init", "gten_hip_last_error", "gten_hip_stream", "gten_hip_sync", "gten_hip_select_stream", "gten_hip_stream_wait", "gten_hip_stream_idle",
        "gten_hip_malloc", "gten_hip_free", "gten_hip_memset", "gten_hip_memcpy_h2d", "gten_hip_memcpy_d2h",
        "gten_hip_memcpy_d2d", "gten_hip_prof_enable", "gten_hip_prof_read", "gten_hip_prof_family_name",
        "gten_hip_selftest_q8scale", "gten_hip_row_bytes", "gten_hip_pack_weight", "gten_hip_token_embed",
        "gten_hip_block_rows", "gten_hip_set_block_rows", "gten_hip_matmul_2d", "gten_hip_rms_norm", "gten_hip_rotary_emb", "gten_hip_silu", "gten_hip_mul",
        "gten_hip_add", "gten_hip_qkv_attn", "gten_hip_argmax_row", "gten_hip_set_prefill_exact", "gten_hip_set_decode_exact", "gten_hip_set_decode_persistent", "gten_hip_persist_status", "gten_hip_set_row_segments", "gten_hip_row_segments_ok", "gten_hip_copy_ranges",
        # fused single-token decoder: driven from C++ (host/tinyllama_model.h), listed here so that
        # the export check covers the whole header
        "gten_hip_decoder_create", "gten_hip_decoder_destroy", "gten_hip_decoder_set_tokens",
        "gten_hip_decoder_step", "gten_hip_decoder_steps", "gten_hip_decoder_generate", "gten_hip_decoder_generate_multi", "gten_hip_decoder_step_ragged", "gten_hip_decoder_result", "gten_hip_decoder_time_family",
        "gten_hip_decoder_create_multi", "gten_hip_decoder_set_tokens_seq", "gten_hip_decoder_result_seq",
        "gten_hip_decoder_logits_seq",
        "gten_hip_decoder_lane_info", "gten_hip_set_lane_skip", "gten_hip_decoder_slot_start", "gten_hip_decoder_slot_start_until", "gten_hip_decoder_slot_park", "gten_hip_decoder_slot_bind", "gten_hip_decoder_slots_apply", "gten_hip_decoder_run", "gten_hip_decoder_run_lanes", "gten_hip_decoder_slot_ids", "gten_hip_decoder_slot_ids_all",
        "gten_hip_set_kv_head_major", "gten_hip_decoder_kv_info", "gten_hip_kv_watch_selftest", "gten_hip_set_ffn_streamed", "gten_hip_set_wx_planes",
    ]

    def __init__(self, path=None):
        path = path or _build.HIP_LIB
        if not os.path.exists(path):
            raise GtenHipError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        self.path = path
        self.lib = L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
        self._count = _sig(L, "gten_hip_device_count", ci, [])
        self._init = _sig(L, "gten_hip_init", ci, [ci])
        self._err = _sig(L, "gten_hip_last_error", C.c_char_p, [])
        self._stream = _sig(L, "gten_hip_stream", vp, [])
        self._sync = _sig(L, "gten_hip_sync", ci, [])
        self._selftest_q8 = _sig(L, "gten_hip_selftest_q8scale", ci, [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)])
        self._malloc = _sig(L, "gten_hip_malloc", ci, [C.POINTER(vp), sz])
        self._free = _sig(L, "gten_hip_free", ci, [vp])
        self._memset = _sig(L, "gten_hip_memset", ci, [vp, ci, sz])
        self._h2d = _sig(L, "gten_hip_memcpy_h2d", ci, [vp, vp, sz])
        self._d2h = _sig(L, "gten_hip_memcpy_d2h", ci, [vp, vp, sz])
        self._d2d = _sig(L, "gten_hip_memcpy_d2d", ci, [vp, vp, sz])
        self._row_bytes = _sig(L, "gten_hip_row_bytes", sz, [ci, ci])
        self._prof_enable = _sig(L, "gten_hip_prof_enable", ci, [ci])
        self._prof_read = _sig(L, "gten_hip_prof_read", ci, [ci, C.POINTER(ci), C.POINTER(C.c_double)])
        self._prof_name = _sig(L, "gten_hip_prof_family_name", C.c_char_p, [ci])
        self._pack = _sig(L, "gten_hip_pack_weight", ci, [vp, ci, ci, ci, vp])
        self._embed = _sig(L, "gten_hip_token_embed", ci, [vp, ci, ci, vp, vp, ci, sz, ci, ci, ci])
        self._matmul = _sig(L, "gten_hip_matmul_2d", ci, [vp, ci, sz, vp, ci, vp, ci, sz, ci, ci, ci, ci])
        self._rms = _sig(L, "gten_hip_rms_norm", ci, [vp, ci, sz, vp, vp, sz, ci, ci, ci])
        self._rope = _sig(L, "gten_hip_rotary_emb", ci, [vp, ci, sz, ci, ci, ci, ci])
        self._silu = _sig(L, "gten_hip_silu", ci, [vp, vp, ci, sz, ci, ci, ci])
        self._mul = _sig(L, "gten_hip_mul", ci, [vp, vp, vp, ci, sz, ci, ci, ci])
        self._add = _sig(L, "gten_hip_add", ci, [vp, vp, vp, ci, sz, ci, ci, ci])
        self._attn = _sig(L, "gten_hip_qkv_attn", ci, [vp, vp, vp, vp, ci, sz, sz, sz, ci, ci, ci, ci, ci])
        self._prefill_exact = _sig(L, "gten_hip_set_prefill_exact", ci, [ci])
        self._block_rows = _sig(L, "gten_hip_block_rows", ci, [C.POINTER(BlockDesc), ci, ci])
        self._set_block_rows = _sig(L, "gten_hip_set_block_rows", ci, [ci])
        self._decode_exact = _sig(L, "gten_hip_set_decode_exact", ci, [ci])
        self._decode_persistent = _sig(L, "gten_hip_set_decode_persistent", ci, [ci])
        self._kv_head_major = _sig(L, "gten_hip_set_kv_head_major", ci, [ci])
        self._kv_watch_selftest = _sig(L, "gten_hip_kv_watch_selftest", ci, [])
        self._ffn_streamed = _sig(L, "gten_hip_set_ffn_streamed", ci, [ci])
        self._wx_planes = _sig(L, "gten_hip_set_wx_planes", ci, [ci])
        self._lane_skip = _sig(L, "gten_hip_set_lane_skip", ci, [ci])
        self._persist_status = _sig(L, "gten_hip_persist_status", ci, [C.POINTER(ci), C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint), C.c_void_p, ci])
        self._set_row_segments = _sig(L, "gten_hip_set_row_segments", ci, [C.c_void_p, ci])
        self._copy_ranges = _sig(L, "gten_hip_copy_ranges", ci, [C.c_void_p, ci])
        self.initialised = False

    # -- runtime
    def _check(self, rc):
        if rc != 0:
            raise GtenHipError(f"gten_hip error {rc}: {self._err().decode(errors='replace')}")

    def device_count(self):
        return self._count()

    def init(self, device=0):
        if self.device_count() <= device:
            raise GtenHipError(f"no MI355X visible (device_count={self.device_count()}, asked for {device}); "
                               "there is no CPU fallback for the gten_hip path")
        self._check(self._init(device))
        self.initialised = True
        return self

    def sync(self):
        self._check(self._sync())

    def selftest_q8scale(self):
        """(mismatches of div127, mismatches of recip_rn) against the IEEE division expansion, on the device"""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        self._check(self._selftest_q8(C.byref(a), C.byref(b)))
        return a.value, b.value

    def stream(self):
        return self._stream()

    def set_row_segments(self, starts):
        """rows [starts[k], starts[k + 1]) of the next gten_hip_block_rows calls are prompt k (None / []: one prompt)"""
        if not starts:
            self._check(self._set_row_segments(None, 0))
            return
        a = np.ascontiguousarray(starts, dtype=np.int32)
        self._check(self._set_row_segments(a.ctypes.data_as(C.c_void_p), len(a) - 1))

    def copy_ranges(self, ranges):
        """ranges: [(dst_ptr, src_ptr, nbytes)] device-to-device, one launch"""
        class R(C.Structure):
            _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("bytes", C.c_size_t)]
        arr = (R * len(ranges))(*[R(int(d), int(s), int(b)) for d, s, b in ranges])
        self._check(self._copy_ranges(C.cast(arr, C.c_void_p), len(ranges)))

    def set_decode_exact(self, on):
        """exact forms of the decode step for decoders created from now on (include/gten_hip.h)"""
        self._check(self._decode_exact(1 if on else 0))

    def set_kv_head_major(self, on):
        """decoders of 16+ sequences created from now on: head-major shadows of the K / V caches (default) or the cache rows as they lie"""
        self._check(self._kv_head_major(1 if on else 0))

    def set_ffn_streamed(self, on):
        """gate | up of full 128-row q4 lanes as the streamed kernel (default) or as k_dec_mmvh: the same bits"""
        self._check(self._ffn_streamed(1 if on else 0))

    def set_wx_planes(self, on):
        """f16 wide decoders: o and down in eight K planes of 64-feature workgroups (default) or as k_dec_mmv_f16 in two"""
        self._check(self._wx_planes(1 if on else 0))

    def kv_watch_selftest(self):
        """host-only check of the watched-cache registry (no GPU needed): 0 = every case as expected, else the failing case"""
        return int(self._kv_watch_selftest())

    def set_decode_persistent(self, on):
        """single-sequence q4 decoders created from now on: the step as ONE persistent launch, or the launch chain (default)"""
        self._check(self._decode_persistent(1 if on else 0))

    def set_lane_skip(self, on):
        """1: gten_hip_decoder_run leaves lanes without a live slot out of the step; 0 (the library's default: skipping measured
        slower on the bench's queue): every run takes every lane"""
        self._check(self._lane_skip(1 if on else 0))

    def persist_status(self, n_stamps=0):
        """(decoders running the persistent step, launches enqueued, abort code [cleared], stamps) -- waits for the stream"""
        nd, nl, ab = C.c_int(0), C.c_ulonglong(0), C.c_uint(0)
        st = (C.c_uint * max(n_stamps, 1))()
        self._check(self._persist_status(C.byref(nd), C.byref(nl), C.byref(ab), C.cast(st, C.c_void_p) if n_stamps else None, n_stamps))
        return nd.value, nl.value, ab.value, list(st)[:n_stamps]

    def set_prefill_exact(self, on):
        """prompt-sized W.x with quantized weights: exact form (scalar-build order, bit for bit) instead of the fast one"""
        self._check(self._prefill_exact(1 if on else 0))

    def select_stream(self, idx):
        """queue the following calls on the library's stream 0 or 1 (include/gten_hip.h)"""
        self._check(self.lib.gten_hip_select_stream(int(idx)))

    def set_block_rows(self, on):
        """prompt-sized AttentionBlock calls as one composed call (default) or module by module"""
        self._check(self._set_block_rows(1 if on else 0))

    def block_rows(self, n, start_pos, ints, bufs):
        """gten_hip_block_rows: `ints` the six integers of the descriptor, `bufs` name -> DeviceBuffer.
        Returns False when the library does not take the configuration (GTEN_HIP_NOT_HANDLED)."""
        d = BlockDesc()
        for k, v in ints.items():
            setattr(d, k, int(v))
        for k, v in bufs.items():
            setattr(d, k, v.ptr)
        rc = self._block_rows(C.byref(d), n, start_pos)
        if rc == 1:
            return False
        self._check(rc)
        return True

    def prof_enable(self, on):
        self._check(self._prof_enable(1 if on else 0))

    def prof_family_index(self, wanted):
        fam = 0
        while True:
            name = self._prof_name(fam)
            if name is None:
                raise KeyError(wanted)
            if name.decode() == wanted:
                return fam
            fam += 1

    def prof_read(self):
        """{family name: (launches, total_ms)} for every family with at least one launch."""
        out = {}
        fam = 0
        while True:
            name = self._prof_name(fam)
            if name is None:
                break
            n, ms = C.c_int(0), C.c_double(0.0)
            self._check(self._prof_read(fam, C.byref(n), C.byref(ms)))
            if n.value:
                out[name.decode()] = (n.value, ms.value)
            fam += 1
        return out

    def row_bytes(self, dtype, cols):
        return self._row_bytes(dtype, cols)

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def upload(self, arr):
        return DeviceBuffer.from_numpy(self, arr)

    def upload_weight(self, blocks, dtype, rows, cols):
        """uint8 [rows][row_bytes] in .gten block order -> packed device weight."""
        src = self.upload(blocks)
        dst = self.alloc(src.nbytes)
        self._check(self._pack(src.ptr, dtype, rows, cols, dst.ptr))
        self.sync()
        src.free()
        return dst

    # -- operators (DeviceBuffers; pitches default to dense rows)
    def token_embed(self, w, w_dtype, n_vocab, tokens, out, out_dtype, n, d, start_pos=0, out_pitch=None):
        self._check(self._embed(w.ptr, w_dtype, n_vocab, tokens.ptr, out.ptr, out_dtype,
                                out_pitch or self.row_bytes(out_dtype, d), n, d, start_pos))

    def matmul_2d(self, x, x_dtype, w, w_dtype, out, out_dtype, n, d_in, d_out, start_pos=0,
                  x_pitch=None, out_pitch=None):
        self._check(self._matmul(x.ptr, x_dtype, x_pitch or self.row_bytes(x_dtype, d_in), w.ptr, w_dtype,
                                 out.ptr, out_dtype, out_pitch or self.row_bytes(out_dtype, d_out),
                                 n, d_in, d_out, start_pos))

    def rms_norm(self, x, dtype, w, out, n, d, start_pos=0):
        p = self.row_bytes(dtype, d)
        self._check(self._rms(x.ptr, dtype, p, w.ptr, out.ptr, p, n, d, start_pos))

    def rotary_emb(self, x, dtype, n, d, d_head, start_pos=0):
        self._check(self._rope(x.ptr, dtype, self.row_bytes(dtype, d), n, d, d_head, start_pos))

    def silu(self, x, out, dtype, n, d, start_pos=0):
        self._check(self._silu(x.ptr, out.ptr, dtype, self.row_bytes(dtype, d), n, d, start_pos))

    def mul(self, a, b, out, dtype, n, d, start_pos=0):
        self._check(self._mul(a.ptr, b.ptr, out.ptr, dtype, self.row_bytes(dtype, d), n, d, start_pos))

    def add(self, a, b, out, dtype, n, d, start_pos=0):
        self._check(self._add(a.ptr, b.ptr, out.ptr, dtype, self.row_bytes(dtype, d), n, d, start_pos))

    def qkv_attn(self, q, k, v, out, dtype, n, n_heads, n_kv_heads, d_head, start_pos=0):
        qp = self.row_bytes(dtype, n_heads * d_head)
        kp = self.row_bytes(dtype, n_kv_heads * d_head)
        self._check(self._attn(q.ptr, k.ptr, v.ptr, out.ptr, dtype, qp, kp, qp, n, n_heads, n_kv_heads, d_head, start_pos))


_api = None


def load(device=None):
    """Process-wide GtenHip; initialised on `device` when given (needs a GPU)."""
    global _api
    if _api is None:
        _api = GtenHip()
    if device is not None and not _api.initialised:
        _api.init(device)
    return _api
