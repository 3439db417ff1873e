"""ctypes binding of include/gten_host.h (libgten_host.so): the C++ model driver."""
import ctypes as C
import os

import numpy as np

from . import build as _build
from .hipabi import GtenHipError, load as _load_hip


class HostConfig(C.Structure):
    _fields_ = [(k, C.c_int) for k in
                ("n_vocab", "max_ctx", "n_embd", "n_ffn", "n_layers", "n_heads", "n_kv_heads", "wdtype", "adtype")]

    def weight_shapes(self):
        E, F, V = self.n_embd, self.n_ffn, self.n_vocab
        KV = (E // self.n_heads) * self.n_kv_heads
        W, F16 = self.wdtype, 1
        out = [(V, E, W)]
        for _ in range(self.n_layers):
            out += [(E, E, W), (KV, E, W), (KV, E, W), (E, E, W), (F, E, W), (F, E, W), (E, F, W), (1, E, F16), (1, E, F16)]
        return out + [(1, E, F16), (V, E, W)]


def _sig(lib, name, res, args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = args
    return f


class GtenHost:
    SYMBOLS = [
        "gten_host_default_config", "gten_host_model_create", "gten_host_model_free", "gten_host_model_n_weights",
        "gten_host_model_weight_bytes", "gten_host_model_set_weight", "gten_host_model_load_gten",
        "gten_host_model_load_synthetic", "gten_host_model_logits", "gten_host_model_greedy", "gten_host_model_generate",
        "gten_host_tokenizer_create", "gten_host_tokenizer_free", "gten_host_tokenizer_encode", "gten_host_tokenizer_decode",
        "gten_host_synth_weight", "gten_host_write_gten", "gten_host_synthetic_tokens",
        "gten_host_model_set_fast_decode", "gten_host_model_decode_begin", "gten_host_model_decode_step", "gten_host_model_decode_steps", "gten_host_batch_decode_steps",
        "gten_host_model_decode_result", "gten_host_model_time_family",
        "gten_host_batch_create", "gten_host_batch_free", "gten_host_batch_load_synthetic", "gten_host_batch_set_weight",
        "gten_host_batch_prefill", "gten_host_batch_prefill_many", "gten_host_batch_generate", "gten_host_batch_serve", "gten_host_batch_serve2", "gten_host_batch_set_serve_schedule", "gten_host_batch_set_serve_spares", "gten_host_batch_set_serve_ramp", "gten_host_batch_decode_begin", "gten_host_batch_decode_step", "gten_host_batch_decode_step_ragged",
        "gten_host_batch_decode_result", "gten_host_batch_logits", "gten_host_batch_time_family", "gten_host_batch_kv_info", "gten_host_batch_seq_steps",
    ]

    def __init__(self, path=None):
        path = path or _build.HOST_LIB
        if not os.path.exists(path):
            raise GtenHipError(f"{path} is missing: run __graft_entry__.build()")
        self.hip = _load_hip()                     # libgten_hip.so first (RTLD_GLOBAL)
        self.lib = L = C.CDLL(path)
        vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
        cfgp = C.POINTER(HostConfig)
        self._defcfg = _sig(L, "gten_host_default_config", None, [cfgp, ci, ci])
        self._create = _sig(L, "gten_host_model_create", vp, [cfgp])
        self._free = _sig(L, "gten_host_model_free", None, [vp])
        self._nw = _sig(L, "gten_host_model_n_weights", ci, [vp])
        self._wb = _sig(L, "gten_host_model_weight_bytes", sz, [vp, ci])
        self._setw = _sig(L, "gten_host_model_set_weight", ci, [vp, ci, vp, sz])
        self._loadg = _sig(L, "gten_host_model_load_gten", ci, [vp, C.c_char_p])
        self._loads = _sig(L, "gten_host_model_load_synthetic", ci, [vp, C.c_uint64])
        self._logits = _sig(L, "gten_host_model_logits", ci, [vp, vp, ci, ci, vp])
        self._greedy = _sig(L, "gten_host_model_greedy", ci, [vp, vp, ci, ci, ci])
        self._generate = _sig(L, "gten_host_model_generate", ci, [vp, vp, ci, ci, ci])
        self._tok_create = _sig(L, "gten_host_tokenizer_create", vp, [C.c_char_p, ci])
        self._tok_free = _sig(L, "gten_host_tokenizer_free", None, [vp])
        self._tok_encode = _sig(L, "gten_host_tokenizer_encode", ci, [vp, C.c_char_p, ci, vp, ci])
        self._tok_decode = _sig(L, "gten_host_tokenizer_decode", C.c_char_p, [vp, ci, ci])
        self._setfast = _sig(L, "gten_host_model_set_fast_decode", ci, [vp, ci])
        self._dbegin = _sig(L, "gten_host_model_decode_begin", ci, [vp, vp, ci])
        self._dstep = _sig(L, "gten_host_model_decode_step", ci, [vp, ci, ci])
        self._dsteps = _sig(L, "gten_host_model_decode_steps", ci, [vp, ci, ci, ci])
        self._bsteps = _sig(L, "gten_host_batch_decode_steps", ci, [vp, ci, ci, ci])
        self._dresult = _sig(L, "gten_host_model_decode_result", ci, [vp, ci, C.POINTER(C.c_int32)])
        self._timefam = _sig(L, "gten_host_model_time_family", ci, [vp, ci, ci, ci, C.POINTER(C.c_double), C.POINTER(ci)])
        self._bcreate = _sig(L, "gten_host_batch_create", vp, [cfgp, ci])
        self._bfree = _sig(L, "gten_host_batch_free", None, [vp])
        self._bloads = _sig(L, "gten_host_batch_load_synthetic", ci, [vp, C.c_uint64])
        self._bsetw = _sig(L, "gten_host_batch_set_weight", ci, [vp, ci, vp, sz])
        self._bprefill = _sig(L, "gten_host_batch_prefill", ci, [vp, ci, vp, ci, vp])
        self._bprefill_many = _sig(L, "gten_host_batch_prefill_many", ci, [vp, vp, vp, vp, ci, vp])
        self._bbegin = _sig(L, "gten_host_batch_decode_begin", ci, [vp, ci, vp, ci])
        self._bstep = _sig(L, "gten_host_batch_decode_step", ci, [vp, ci, ci])
        self._bstepr = _sig(L, "gten_host_batch_decode_step_ragged", ci, [vp, vp, ci])
        self._bgen = _sig(L, "gten_host_batch_generate", ci, [vp, vp, vp, ci, ci, ci, vp, vp])
        self._bserve = _sig(L, "gten_host_batch_serve2", ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, ci])
        self._bsched = _sig(L, "gten_host_batch_set_serve_schedule", ci, [vp, ci])
        self._bspares = _sig(L, "gten_host_batch_set_serve_spares", ci, [vp, ci])
        self._bramp = _sig(L, "gten_host_batch_set_serve_ramp", ci, [vp, ci])
        self._bresult = _sig(L, "gten_host_batch_decode_result", ci, [vp, ci, ci, C.POINTER(C.c_int32)])
        self._blogits = _sig(L, "gten_host_batch_logits", ci, [vp, ci, vp])
        self._btime = _sig(L, "gten_host_batch_time_family", ci, [vp, ci, ci, ci, C.POINTER(C.c_double), C.POINTER(ci)])
        self._bseqsteps = _sig(L, "gten_host_batch_seq_steps", ci, [vp, ci, vp, ci, ci, ci])
        self._bkvinfo = _sig(L, "gten_host_batch_kv_info", ci, [vp, C.POINTER(ci), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)])
        self._synthw = _sig(L, "gten_host_synth_weight", ci, [cfgp, C.c_uint64, ci, vp, sz])
        self._writeg = _sig(L, "gten_host_write_gten", ci, [cfgp, C.c_uint64, C.c_char_p])
        self._stoks = _sig(L, "gten_host_synthetic_tokens", None, [vp, ci, C.c_uint32, ci])

    def default_config(self, wdtype, adtype):
        cfg = HostConfig()
        self._defcfg(C.byref(cfg), wdtype, adtype)
        return cfg

    def synth_weight(self, cfg, seed, idx):
        rows, cols, dt = cfg.weight_shapes()[idx]
        out = np.zeros(rows * self.hip.row_bytes(dt, cols), np.uint8)
        rc = self._synthw(C.byref(cfg), seed, idx, out.ctypes.data_as(C.c_void_p), out.size)
        if rc:
            raise GtenHipError(f"gten_host_synth_weight rc={rc}")
        return out

    def write_gten(self, cfg, seed, path):
        rc = self._writeg(C.byref(cfg), seed, path.encode())
        if rc:
            raise GtenHipError(f"gten_host_write_gten rc={rc}")

    def synthetic_tokens(self, count, seed=12345, n_vocab=32003):
        out = np.zeros(count, np.int32)
        self._stoks(out.ctypes.data_as(C.c_void_p), count, seed, n_vocab)
        return out

    def model(self, cfg):
        return HostModel(self, cfg)

    def batch(self, cfg, n_seq):
        return HostBatch(self, cfg, n_seq)

    def tokenizer(self, path, vocab_size=32000):
        """host/tokenizer.h on a vocabulary file (host only: no GPU needed)"""
        return HostTokenizer(self, path, vocab_size)


class HostTokenizer:
    def __init__(self, host, path, vocab_size):
        self.host = host
        self.h = host._tok_create(str(path).encode(), vocab_size)
        if not self.h:
            raise GtenHipError(f"tokenizer: cannot open {path}")

    def encode(self, prompt, chat_template=True):
        data = prompt if isinstance(prompt, bytes) else prompt.encode("utf-8")
        cap = 16 + 2 * len(data)
        buf = np.zeros(cap, np.int32)
        n = self.host._tok_encode(self.h, data, 1 if chat_template else 0, buf.ctypes.data_as(C.c_void_p), cap)
        if n < 0:
            raise GtenHipError(f"tokenizer_encode rc={n}")
        return buf[:n].tolist()

    def decode(self, prev_token, token):
        return self.host._tok_decode(self.h, prev_token, token)        # bytes

    def close(self):
        if self.h:
            self.host._tok_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostModel:
    """The C++ TinyLlama driver on the GPU (needs an initialised device)."""

    def __init__(self, host, cfg):
        self.host = host
        self.cfg = cfg
        if host.hip.device_count() < 1:
            raise GtenHipError("no MI355X visible: the model runs only on the gten_hip path")
        self.h = host._create(C.byref(cfg))

    def n_weights(self):
        return self.host._nw(self.h)

    def weight_bytes(self, idx):
        return self.host._wb(self.h, idx)

    def set_weight(self, idx, data):
        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        rc = self.host._setw(self.h, idx, data.ctypes.data_as(C.c_void_p), data.size)
        if rc:
            raise GtenHipError(f"set_weight({idx}) rc={rc}")

    def load_gten(self, path):
        rc = self.host._loadg(self.h, path.encode())
        if rc:
            raise GtenHipError(f"load_gten({path}) rc={rc}")

    def load_synthetic(self, seed):
        rc = self.host._loads(self.h, seed)
        if rc:
            raise GtenHipError(f"load_synthetic rc={rc}")

    def logits(self, tokens, start_pos, want=True):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.zeros(self.cfg.n_vocab, np.float32) if want else None
        rc = self.host._logits(self.h, tokens.ctypes.data_as(C.c_void_p), len(tokens), start_pos,
                               out.ctypes.data_as(C.c_void_p) if want else None)
        if rc:
            raise GtenHipError(f"logits rc={rc}")
        return out

    def set_fast_decode(self, on):
        self.host._setfast(self.h, 1 if on else 0)

    def decode_begin(self, tokens):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        rc = self.host._dbegin(self.h, tokens.ctypes.data_as(C.c_void_p), len(tokens))
        if rc:
            raise GtenHipError(f"decode_begin rc={rc}")

    def decode_step(self, n, use_graph=True):
        rc = self.host._dstep(self.h, n, 1 if use_graph else 0)
        if rc:
            raise GtenHipError(f"decode_step({n}) rc={rc}")

    def decode_steps(self, n_first, count, use_graph=True):
        """asynchronous: steps n_first .. n_first + count - 1 (ids from decode_begin), four steps per graph replay"""
        rc = self.host._dsteps(self.h, n_first, count, 1 if use_graph else 0)
        if rc:
            raise GtenHipError(f"decode_steps({n_first}, {count}) rc={rc}")

    def decode_result(self, n):
        out = C.c_int32(-1)
        rc = self.host._dresult(self.h, n, C.byref(out))
        if rc:
            raise GtenHipError(f"decode_result({n}) rc={rc}")
        return out.value

    def time_family(self, family, n, reps=20):
        """(average us per launch, launches per replay) of one kernel family, HIP-event timed"""
        us, cnt = C.c_double(0.0), C.c_int(0)
        rc = self.host._timefam(self.h, family, n, reps, C.byref(us), C.byref(cnt))
        if rc:
            raise GtenHipError(f"time_family rc={rc}")
        return us.value, cnt.value

    def greedy(self, prompt, max_tokens, eos=-1):
        buf = np.zeros(max_tokens, np.int32)
        buf[: len(prompt)] = prompt
        total = self.host._greedy(self.h, buf.ctypes.data_as(C.c_void_p), len(prompt), max_tokens, eos)
        return buf[:total].copy()

    def generate(self, prompt, max_tokens, eos=-1):
        """greedy ids with the sampler on the device (gten_host_model_generate): same ids as greedy()"""
        buf = np.zeros(max_tokens, np.int32)
        buf[: len(prompt)] = prompt
        total = self.host._generate(self.h, buf.ctypes.data_as(C.c_void_p), len(prompt), max_tokens, eos)
        return buf[:total].copy()

    def close(self):
        if self.h:
            self.host._free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostBatch:
    """n_seq sequences sharing one copy of the weights; one step advances all of them."""

    def __init__(self, host, cfg, n_seq):
        self.host, self.cfg, self.n_seq = host, cfg, n_seq
        if host.hip.device_count() < 1:
            raise GtenHipError("no MI355X visible: the model runs only on the gten_hip path")
        self.h = host._bcreate(C.byref(cfg), n_seq)
        if not self.h:
            raise GtenHipError(f"batch_create(n_seq={n_seq}) failed")

    def _ck(self, rc, what):
        if rc:
            raise GtenHipError(f"{what} rc={rc}")

    def load_synthetic(self, seed):
        self._ck(self.host._bloads(self.h, seed), "batch_load_synthetic")

    def set_weight(self, idx, data):
        data = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
        self._ck(self.host._bsetw(self.h, idx, data.ctypes.data_as(C.c_void_p), data.size), f"batch_set_weight({idx})")

    def prefill(self, seq, tokens, want=True):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.zeros(self.cfg.n_vocab, np.float32) if want else None
        self._ck(self.host._bprefill(self.h, seq, tokens.ctypes.data_as(C.c_void_p), len(tokens),
                                     out.ctypes.data_as(C.c_void_p) if want else None), "batch_prefill")
        return out

    def prefill_many(self, seqs, prompts, want=True):
        """several prompts (>= 16 ids each) as segments of one row matrix, prompt k onto the caches of sequence seqs[k];
        returns [n_prompts][n_vocab] logits (wide batches only: raises otherwise)"""
        seqs = np.ascontiguousarray(seqs, dtype=np.int32)
        starts = np.zeros(len(prompts) + 1, np.int32)
        starts[1:] = np.cumsum([len(p) for p in prompts])
        toks = np.ascontiguousarray(np.concatenate([np.asarray(p, np.int32) for p in prompts]), dtype=np.int32)
        out = np.zeros((len(prompts), self.cfg.n_vocab), np.float32) if want else None
        self._ck(self.host._bprefill_many(self.h, seqs.ctypes.data_as(C.c_void_p), toks.ctypes.data_as(C.c_void_p), starts.ctypes.data_as(C.c_void_p),
                                          len(prompts), out.ctypes.data_as(C.c_void_p) if want else None), "batch_prefill_many")
        return out

    def decode_begin(self, seq, tokens):
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        self._ck(self.host._bbegin(self.h, seq, tokens.ctypes.data_as(C.c_void_p), len(tokens)), "batch_decode_begin")

    def decode_step(self, n, use_graph=True):
        self._ck(self.host._bstep(self.h, n, 1 if use_graph else 0), f"batch_decode_step({n})")

    def decode_steps(self, n_first, count, use_graph=True):
        self._ck(self.host._bsteps(self.h, n_first, count, 1 if use_graph else 0), f"batch_decode_steps({n_first}, {count})")

    def decode_step_ragged(self, ns, use_graph=True):
        """sequence q decodes row ns[q] - 1 (continuous batching)"""
        ns = np.ascontiguousarray(ns, dtype=np.int32)
        assert len(ns) == self.n_seq
        self._ck(self.host._bstepr(self.h, ns.ctypes.data_as(C.c_void_p), 1 if use_graph else 0), "batch_decode_step_ragged")

    def generate(self, prompts, max_tokens, eos=-1):
        """greedy generation of every sequence (sampler on the device): list of id arrays, one per sequence, prompt included"""
        assert len(prompts) == self.n_seq
        mp = max(len(p) for p in prompts)
        pr = np.zeros((self.n_seq, mp), np.int32)
        npr = np.zeros(self.n_seq, np.int32)
        for q, p in enumerate(prompts):
            pr[q, : len(p)] = p
            npr[q] = len(p)
        out = np.zeros((self.n_seq, max_tokens), np.int32)
        tot = np.zeros(self.n_seq, np.int32)
        self._ck(self.host._bgen(self.h, pr.ctypes.data_as(C.c_void_p), npr.ctypes.data_as(C.c_void_p), mp, max_tokens, eos,
                                 out.ctypes.data_as(C.c_void_p), tot.ctypes.data_as(C.c_void_p)), "batch_generate")
        return [out[q, : tot[q]].copy() for q in range(self.n_seq)]

    def set_serve_schedule(self, k):
        """tests: exactly k prompts beside every slice (0: as many as fit while it runs)"""
        self._ck(self.host._bsched(self.h, int(k)), "set_serve_schedule")

    def set_serve_spares(self, n):
        """cache sets that serve() fills ahead of the slots that will take them (-1: default, 0: none)"""
        self._ck(self.host._bspares(self.h, int(n)), "set_serve_spares")

    def set_serve_ramp(self, percent):
        """percent of the slots that get a processed prompt before a queue's first slice starts (default 100)"""
        self._ck(self.host._bramp(self.h, int(percent)), "set_serve_ramp")

    def serve(self, prompts, max_tokens, eos=-1, slice_steps=16, max_new=0, max_new_each=None):
        """continuous batching: the queue `prompts` (any number) through this batch's slots; returns (list of id
        arrays -- prompt + new ids, one per prompt, in queue order -- and a dict of counters)"""
        mp = max(len(p) for p in prompts)
        width = max(max_tokens, mp)
        pr = np.zeros((len(prompts), mp), np.int32)
        npr = np.zeros(len(prompts), np.int32)
        for j, p in enumerate(prompts):
            pr[j, : len(p)] = p
            npr[j] = len(p)
        out = np.zeros((len(prompts), width), np.int32)
        tot = np.zeros(len(prompts), np.int32)
        st = np.zeros(9, np.float64)
        each = None if max_new_each is None else np.ascontiguousarray(max_new_each, dtype=np.int32)
        assert each is None or len(each) == len(prompts)
        self._ck(self.host._bserve(self.h, pr.ctypes.data_as(C.c_void_p), npr.ctypes.data_as(C.c_void_p), len(prompts), mp, max_tokens, eos,
                                   slice_steps, max_new, None if each is None else each.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), tot.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), len(st)),
                 "batch_serve")
        keys = ("prompt_tokens", "new_tokens", "steps", "admissions", "prefill_s", "decode_s", "lane_steps", "lane_rows", "moved")
        return [out[j, : tot[j]].copy() for j in range(len(prompts))], dict(zip(keys, st.tolist()))

    def decode_result(self, seq, n):
        out = C.c_int32(-1)
        self._ck(self.host._bresult(self.h, seq, n, C.byref(out)), "batch_decode_result")
        return out.value

    def logits(self, seq):
        out = np.zeros(self.cfg.n_vocab, np.float32)
        self._ck(self.host._blogits(self.h, seq, out.ctypes.data_as(C.c_void_p)), "batch_logits")
        return out

    def seq_steps(self, seq, tokens, n_first, steps):
        """steps n_first .. n_first + steps - 1 of ONE sequence on its own single-sequence decoder (same caches as the shared one)"""
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        self._ck(self.host._bseqsteps(self.h, seq, tokens.ctypes.data_as(C.c_void_p), len(tokens), n_first, steps), "batch_seq_steps")

    def kv_info(self):
        """(head-major K / V shadows kept, sequence imports launched so far, import launches) of the shared decoder"""
        hm, imp, lau = C.c_int(0), C.c_ulonglong(0), C.c_ulonglong(0)
        self._ck(self.host._bkvinfo(self.h, C.byref(hm), C.byref(imp), C.byref(lau)), "batch_kv_info")
        return bool(hm.value), imp.value, lau.value

    def time_family(self, family, n, reps=20):
        us, cnt = C.c_double(0.0), C.c_int(0)
        self._ck(self.host._btime(self.h, family, n, reps, C.byref(us), C.byref(cnt)), "batch_time_family")
        return us.value, cnt.value

    def close(self):
        if self.h:
            self.host._bfree(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_host = None


def load_host():
    global _host
    if _host is None:
        _host = GtenHost()
    return _host
