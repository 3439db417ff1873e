"""HF TinyLlama checkpoint -> .gten file.

Counterpart of the reference's `tinyllama_to_gten.py` (behaviour restated, nothing copied):
  * record order and framing: tinyllama_to_gten.py:94-109,150-201 / reader tinyllama.cpp:301-392
  * weight quantizers: tinyllama_to_gten.py:24-91 -- per 32-wide block delta32 = absmax / 127 (Q8) or / 7 (Q4),
    q = round-half-to-EVEN(x / delta32) (torch.round), stored delta = fp16(delta32); Q4 adds 7 and packs elements
    0..15 into the high nibbles, 16..31 into the low nibbles; a zero block stores delta 0 and q 0
  * norm vectors stay fp16 in every mode; embedding and lm_head are quantized like any linear

Checkpoint dtype: the reference divides absmax by 127 (7) IN THE CHECKPOINT'S OWN DTYPE before the delta goes to f32
(tinyllama_to_gten.py:38-39, 68-69), so for a bf16 checkpoint -- TinyLlama's own -- the f32 delta is a bf16 value, and
the quants follow from it.  `quantize_q8 / quantize_q4(w, src=...)` reproduce that for src in ("f32", "bf16", "f16");
`open_checkpoint` reads bf16 / f16 / f32 safetensors (through torch for bf16, which numpy cannot hold) and reports each
tensor's dtype.  Other dtypes are refused.

numpy only (plus `safetensors`, and torch for bf16 files, to read a checkpoint); pinned byte for byte by tests/golden/converter_pins.npz
(tests/test_convert_cpu.py).  The GPU library repacks these blocks at load (gten_hip_pack_weight).

    python -m tinyllama.cpp_amd.convert <model.safetensors | checkpoint dir> out.q4.gten --dtype q4
"""
import os
import struct
import sys

import numpy as np

GTEN_MAGIC = 0x454C49464E455447
DTYPES = ("f16", "q8", "q4")


def to_f16(w):
    return np.ascontiguousarray(w, dtype=np.float32).astype(np.float16).view(np.uint8).reshape(w.shape[0], -1)


def _blocks(w):
    w = np.ascontiguousarray(w, dtype=np.float32)
    rows, cols = w.shape
    if cols % 32:
        raise ValueError(f"row length {cols} is not a multiple of the 32-wide quant block")
    return w.reshape(rows, cols // 32, 32), rows, cols // 32


SRC_DTYPES = ("f32", "bf16", "f16")


def _round_to(x, src):
    """f32 values rounded (to nearest even) to the checkpoint's dtype, returned as f32"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if src == "f32":
        return x
    if src == "f16":
        return x.astype(np.float16).astype(np.float32)
    if src == "bf16":
        u = x.view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000          # round to nearest even on the upper 16 bits
        return u.astype(np.uint32).view(np.float32)
    raise ValueError(f"checkpoint dtype {src!r} not in {SRC_DTYPES}")


def _quants(blocks, qmax, src="f32"):
    amax = np.abs(blocks).max(axis=2, keepdims=True)
    delta = _round_to((amax / np.float32(qmax)).astype(np.float32), src)   # the division happens in the checkpoint's dtype
    scale = np.divide(np.float32(1.0), delta, out=np.zeros_like(delta), where=delta != 0)
    q = np.rint(blocks * scale).astype(np.int32)          # np.rint: half to even, like torch.round
    return q, delta.astype(np.float16)


def quantize_q8(w, src="f32"):
    """[rows][cols] f32 -> [rows][cols/32 * 34] bytes: {f16 delta, int8 x 32} per block"""
    blocks, rows, nb = _blocks(w)
    q, d16 = _quants(blocks, 127, src)
    out = np.empty((rows, nb, 34), np.uint8)
    out[:, :, :2] = d16.view(np.uint8).reshape(rows, nb, 2)
    out[:, :, 2:] = q.astype(np.int8).view(np.uint8)
    return out.reshape(rows, nb * 34)


def quantize_q4(w, src="f32"):
    """[rows][cols] f32 -> [rows][cols/32 * 18] bytes: {f16 delta, 16 bytes: hi nibble = element j + 7, lo = element j + 16 + 7}"""
    blocks, rows, nb = _blocks(w)
    q, d16 = _quants(blocks, 7, src)
    q = (q + 7).astype(np.uint8)
    out = np.empty((rows, nb, 18), np.uint8)
    out[:, :, :2] = d16.view(np.uint8).reshape(rows, nb, 2)
    out[:, :, 2:] = (q[:, :, :16] << 4) | q[:, :, 16:]
    return out.reshape(rows, nb * 18)


def quantize(w, dtype, src="f32"):
    if dtype == "f16":
        return to_f16(w)
    return {"q8": quantize_q8, "q4": quantize_q4}[dtype](w, src)


def tensor_names(n_layers):
    """HF keys in .gten record order (tinyllama.cpp:345-391); second field: quantized like a linear?"""
    names = [("model.embed_tokens.weight", True)]
    for i in range(n_layers):
        p = f"model.layers.{i}."
        names += [(p + f"self_attn.{k}_proj.weight", True) for k in ("q", "k", "v", "o")]
        names += [(p + f"mlp.{k}_proj.weight", True) for k in ("gate", "up", "down")]
        names += [(p + "input_layernorm.weight", False), (p + "post_attention_layernorm.weight", False)]
    names += [("model.norm.weight", False), ("lm_head.weight", True)]
    return names


def write_gten(path, get_tensor, n_layers, dtype, progress=None):
    """get_tensor(name) -> 2-D (linear) or 1-D (norm) float array, or a pair (array, checkpoint dtype in SRC_DTYPES).
    Returns the number of bytes written."""
    if dtype not in DTYPES:
        raise ValueError(f"dtype {dtype!r} not in {DTYPES}")
    total = 0
    with open(path, "wb") as f:
        f.write(struct.pack("<q", GTEN_MAGIC))
        total += 8
        for name, is_linear in tensor_names(n_layers):
            got = get_tensor(name)
            w, src = got if isinstance(got, tuple) else (got, "f32")
            w = np.asarray(w, dtype=np.float32)
            if is_linear:
                if w.ndim != 2:
                    raise ValueError(f"{name}: expected a matrix, got shape {w.shape}")
                payload = quantize(w, dtype, src)
            else:
                payload = to_f16(w.reshape(1, -1))
            raw = name.encode()
            # the reference writes every name twice (length + bytes), then the payload size
            for _ in range(2):
                f.write(struct.pack("<i", len(raw)))
                f.write(raw)
            buf = payload.tobytes()
            f.write(struct.pack("<i", len(buf)))
            f.write(buf)
            total += 2 * (4 + len(raw)) + 4 + len(buf)
            if progress:
                progress(name, payload.shape)
    return total


def open_checkpoint(src):
    """name -> (f32 array, checkpoint dtype) accessor over one .safetensors file or a directory of shards"""
    from safetensors import safe_open
    files = [src] if os.path.isfile(src) else sorted(os.path.join(src, f) for f in os.listdir(src) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no .safetensors under {src}")
    handles = [safe_open(f, framework="np") for f in files]
    index = {k: (h, f) for h, f in zip(handles, files) for k in h.keys()}
    pt_handles = {}

    def get(name):
        if name not in index:
            raise KeyError(f"{name} missing from the checkpoint")
        h, f = index[name]
        dt = h.get_slice(name).get_dtype()
        if dt in ("F32", "F16"):
            return np.asarray(h.get_tensor(name), dtype=np.float32), ("f32" if dt == "F32" else "f16")
        if dt == "BF16":
            import torch                       # numpy has no bf16: read through torch, hand on the exact f32 values
            if f not in pt_handles:
                pt_handles[f] = safe_open(f, framework="pt")
            return pt_handles[f].get_tensor(name).to(torch.float32).numpy(), "bf16"
        raise ValueError(f"{name}: checkpoint dtype {dt} is not supported (f32, f16 or bf16)")

    n_layers = 1 + max((int(k.split(".")[2]) for k in index if k.startswith("model.layers.")), default=-1)
    return get, n_layers


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("src")
    ap.add_argument("out")
    ap.add_argument("--dtype", choices=DTYPES, default="q4")
    a = ap.parse_args(argv)
    get, n_layers = open_checkpoint(a.src)
    n = write_gten(a.out, get, n_layers, a.dtype, progress=lambda name, shape: print(f"  {name:56s} {shape}", file=sys.stderr))
    print(f"{a.out}: {n} bytes, {n_layers} layers, {a.dtype}")


if __name__ == "__main__":
    main()
