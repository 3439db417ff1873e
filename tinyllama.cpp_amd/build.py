"""Build the native libraries of this package in-tree.

  csrc/libgten_hip.so   hand-written HIP kernels + the C-ABI of include/gten_hip.h
                        (hipcc --offload-arch=gfx950; cross-compiles without a GPU)
  host/libgten_host.so  C++ host side: gten API mirror, TinyLlama driver, synthetic
                        weights (g++; links libgten_hip.so)

The .so files are git-ignored but travel with gpurun snapshots, so the GPU box
uses what was built here; `build_all()` rebuilds only what is out of date.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
INCLUDE = os.path.join(ROOT, "include")

HIP_LIB = os.path.join(CSRC, "libgten_hip.so")
HOST_LIB = os.path.join(HOST, "libgten_host.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: the reference rounds every multiply and add separately
# (gten/simd_ops.h:59-61); fused multiply-adds would move results off its grid.
# -amdgpu-kernarg-preload-count: the command processor loads the leading scalar kernel arguments into SGPRs while
# the waves are created, so a kernel can form its first addresses without a scalar-load round trip (measured:
# -0.3 us per dependent launch, tools/microbench_launch_floor.hip; 14 dwords fit beside the segment pointer).
# -fno-slp-vectorize, -packed-fp32-ops: NO packed-f32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) in any kernel.  hipcc's
# SLP vectorizer pairs neighbouring scalar f32 operations into them; on this chip they are half rate anyway, and -- found
# in round 2 -- they are NOT SAFE beside another stream's matrix-core kernel: with the SIMDs saturated by a neighbour's
# MFMA stream (a prompt's GEMMs beside the decode step), v_pk_*_f32 results of the fused decoder's RMSNorm prologue came
# back wrong in the last 16 lanes of a wave (a deterministic "older" value, so 1 / rms was off by +0.3..0.5 % in ~25 % of
# the workgroups of the q|k|v and lm_head launches; alone on the GPU every run was bit-identical).  Without the packed
# instructions the same test is clean and the step is not slower (profiles/README.md, "packed f32 beside MFMA";
# tests/test_no_packed_f32_cpu.py checks the generated code of every kernel).
# (-target-feature -packed-fp32-ops makes the code generator itself refuse them -- float2 arithmetic in the sources and
# the loop vectorizer produce them too; the host half of the compilation prints "not a recognized feature", harmless)
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
             "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-fPIC", "-shared",
             "-mllvm", "-amdgpu-kernarg-preload-count=16",
             "-Wall", "-Wno-unused-function", "-I" + INCLUDE]
CXX_FLAGS = ["-std=c++17", "-O2", "-fopenmp", "-fPIC", "-shared", "-Wall", "-I" + INCLUDE, "-I" + PKG]


# A/B builds on the GPU box: extra hipcc flags (e.g. "-DHM_OCC=") from the environment; a change of the flags rebuilds everything
_EXTRA = os.environ.get("GTEN_HIP_EXTRA_FLAGS", "").split()
HIP_FLAGS += _EXTRA
_FLAGS_STAMP = os.path.join(CSRC, "_obj", "flags.txt")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources(d, exts):
    out = []
    for base, _, files in os.walk(d):
        out += [os.path.join(base, f) for f in files if f.endswith(exts)]
    return sorted(out)


# per-file additions (none at present: -fno-slp-vectorize used to be one, for the MFMA kernels only)
HIP_FILE_FLAGS = {}
HIP_OBJ = os.path.join(CSRC, "_obj")


def build_hip(force=False):
    """one object per .hip file (only stale ones are recompiled, up to 4 at a time), then one link"""
    from concurrent.futures import ThreadPoolExecutor
    srcs = _sources(CSRC, (".hip",))
    hdrs = _sources(CSRC, (".h",)) + _sources(INCLUDE, (".h",)) + [os.path.abspath(__file__)]
    os.makedirs(HIP_OBJ, exist_ok=True)
    compile_flags = [f for f in HIP_FLAGS if f != "-shared"]
    stamp = " ".join(_EXTRA)
    if (open(_FLAGS_STAMP).read() if os.path.exists(_FLAGS_STAMP) else "") != stamp:
        force = True
        with open(_FLAGS_STAMP, "w") as f:
            f.write(stamp)
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(HIP_OBJ, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            jobs.append([HIPCC] + compile_flags + HIP_FILE_FLAGS.get(os.path.basename(src), []) + ["-c", "-o", obj, src])
    if jobs:
        with ThreadPoolExecutor(max_workers=4) as pool:
            for r in pool.map(lambda cmd: subprocess.run(cmd, capture_output=True, text=True), jobs):
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed: " + " ".join(r.args) + "\n" + r.stderr[-4000:])
    if jobs or force or _newer(HIP_LIB, objs):
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs, check=True)
    return HIP_LIB


HOST_CLI = os.path.join(HOST, "tinyllama_cli")
CLI_SRC = os.path.join(HOST, "tinyllama_cli.cpp")


def build_host(force=False):
    srcs = [s for s in _sources(HOST, (".cpp",)) if s != CLI_SRC]
    if not srcs:
        return None
    deps = srcs + _sources(HOST, (".h",)) + _sources(os.path.join(PKG, "gten"), (".h",)) + _sources(INCLUDE, (".h",))
    if force or _newer(HOST_LIB, deps) or _newer(HOST_LIB, [HIP_LIB]):
        subprocess.run(["g++"] + CXX_FLAGS + ["-o", HOST_LIB] + srcs +
                       ["-L" + CSRC, "-lgten_hip", "-Wl,-rpath,$ORIGIN/../csrc"], check=True)
    # the command line program (host/tinyllama_cli.cpp): the reference's main() on this repository's gten API
    if os.path.exists(CLI_SRC) and (force or _newer(HOST_CLI, deps + [CLI_SRC]) or _newer(HOST_CLI, [HIP_LIB])):
        cli_flags = [f for f in CXX_FLAGS if f not in ("-shared", "-fPIC")]
        subprocess.run(["g++"] + cli_flags + ["-o", HOST_CLI, CLI_SRC, "-L" + CSRC, "-lgten_hip", "-Wl,-rpath,$ORIGIN/../csrc"], check=True)
    return HOST_LIB


def build_all(force=False):
    build_hip(force)
    build_host(force)
