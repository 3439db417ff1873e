"""not gpu: no kernel of libgten_hip.so may contain a packed-f32 VALU instruction (v_pk_mul_f32 / v_pk_add_f32 /
v_pk_fma_f32 / v_pk_mov_b32).  Round 2 found them unsafe beside another stream's matrix-core kernel on the MI355X: with the
SIMDs saturated by a neighbour's MFMA stream, their results came back wrong in the last 16 lanes of a wave (the fused
decoder's RMSNorm prologue, profiles/README.md "packed f32 beside MFMA").  build.py switches them off in the code generator
(-target-feature -packed-fp32-ops, -fno-slp-vectorize); this test compiles every kernel file to gfx950 assembly with the
build's own flags and looks at what came out.  tests/test_zz_neighbour_gpu.py is the run-time half."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

PACKED = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32|v_pk_mov_b32)\b", re.M)
# Mixed-precision fused forms with an fp16 RESULT (v_fma_mixlo_f16 / v_fma_mixhi_f16): ONE rounding of a product (+ addend)
# where the sources -- and the reference, gten/ops.h:73-96 -- have a separate f32 operation and an fp16 convert.  Round 2
# shipped a build whose fused decoder had them in its RMSNorm prologue while the operator kernel did not (a flag change
# re-fused `v * inv * w` with the convert): fused != operator path by one fp16 ulp on ties.  gten_dev.h's f2hv() makes
# the f32 value opaque ahead of every convert, so no flag can re-fuse it; this test reads the generated code.
# (v_fma_mix_f32 -- f32 result, an fp16 source widened on the way in -- is exact widening + an fma the sources spell out
# with __builtin_fmaf; -ffp-contract=off keeps the compiler from inventing others.  Not a rounding change, allowed.)
MIXED = re.compile(r"^\s*(v_(?:fma|mad)_mix(?:lo|hi)_f16)\b", re.M)


def test_generated_code_has_no_packed_f32_instructions(tmp_path):
    from __graft_entry__ import load_package
    b = load_package().build
    if shutil.which(b.HIPCC) is None and not os.path.exists(b.HIPCC):
        pytest.skip("no hipcc")
    assert "-packed-fp32-ops" in b.HIP_FLAGS and "-fno-slp-vectorize" in b.HIP_FLAGS
    flags = [f for f in b.HIP_FLAGS if f not in ("-shared", "-fPIC")]
    srcs = b._sources(b.CSRC, (".hip",))
    assert len(srcs) >= 5

    def asm(src):
        out = str(tmp_path / (os.path.basename(src) + ".s"))
        r = subprocess.run([b.HIPCC] + flags + b.HIP_FILE_FLAGS.get(os.path.basename(src), []) + ["-S", "--cuda-device-only", "-o", out, src],
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return open(out).read()

    with ThreadPoolExecutor(max_workers=4) as pool:
        texts = list(pool.map(asm, srcs))
    n_kernels = 0
    for src, text in zip(srcs, texts):
        n_kernels += text.count(".amdhsa_kernel ")
        found = PACKED.findall(text)
        assert not found, f"{os.path.basename(src)}: {len(found)} packed-f32 instructions ({sorted(set(found))})"
        mixed = MIXED.findall(text)
        assert not mixed, f"{os.path.basename(src)}: {len(mixed)} mixed-precision fused instructions ({sorted(set(mixed))})"
    assert n_kernels > 100          # every template instance of the library went through the check
    # No FLAT memory instruction in the decode, prompt-GEMM and prompt-attention translation units (round 5): a pointer that went through an
    # integer (an LDS pointer aligned via uintptr_t) or out of an argument struct comes back generic, hipcc then emits flat_load / flat_store,
    # which may alias LDS: they count on both counters, order against every LDS access, and sent k_dec_wxp_f16's weight pieces through
    # scratch.  (gten_ops.hip's self-test kernel takes a generic pointer on purpose.)
    for src, text in zip(srcs, texts):
        if os.path.basename(src) in ("gten_decode.hip", "gten_mfma.hip", "gten_attn_tiled.hip"):
            flat = re.findall(r"^\s*(flat_(?:load|store|atomic)\w*)", text, re.M)
            assert not flat, f"{os.path.basename(src)}: {len(flat)} flat memory instructions ({sorted(set(flat))})"
    # ... and no kernel of the decode unit spills more than a handful of registers (k_dec_ffn_q8<false>: 4) or indexes a register array at run time
    dec = texts[[os.path.basename(x) for x in srcs].index("gten_decode.hip")]
    worst = max((int(m) for m in re.findall(r"; ScratchSize: (\d+)", dec)), default=0)
    assert worst <= 32, f"gten_decode.hip: a kernel with {worst} bytes of scratch"
    # row16_max (gten_attn_tiled.hip) is a hand-written DPP chain: the hazard recogniser does not look inside inline assembly,
    # so whatever the compiler places behind it -- possibly a DPP or readlane consumer of the result -- gets no wait states of
    # its own.  The text therefore ends with `s_nop 1`; here: every last step of the chain is followed by exactly that.
    tiled = texts[[os.path.basename(x) for x in srcs].index("gten_attn_tiled.hip")]
    lines = [ln.strip() for ln in tiled.splitlines() if ln.strip() and not ln.strip().startswith((";", "//", "."))]
    last_steps = [i for i, ln in enumerate(lines) if ln.startswith("v_max_f32_dpp") and "row_mirror" in ln and "row_half_mirror" not in ln]
    assert len(last_steps) >= 8
    for i in last_steps:
        assert lines[i + 1].startswith("s_nop 1"), (lines[i], lines[i + 1])
