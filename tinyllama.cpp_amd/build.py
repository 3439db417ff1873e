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
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
             "-Wall", "-Wno-unused-function", "-I" + INCLUDE]
CXX_FLAGS = ["-std=c++17", "-O2", "-fopenmp", "-fPIC", "-shared", "-Wall", "-I" + INCLUDE, "-I" + PKG]


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


def build_hip(force=False):
    srcs = _sources(CSRC, (".hip",))
    deps = srcs + _sources(CSRC, (".h",)) + _sources(INCLUDE, (".h",))
    if force or _newer(HIP_LIB, deps):
        subprocess.run([HIPCC] + HIP_FLAGS + ["-o", HIP_LIB] + srcs, check=True)
    return HIP_LIB


def build_host(force=False):
    srcs = _sources(HOST, (".cpp",))
    if not srcs:
        return None
    deps = srcs + _sources(HOST, (".h",)) + _sources(os.path.join(PKG, "gten"), (".h",)) + _sources(INCLUDE, (".h",))
    if force or _newer(HOST_LIB, deps) or _newer(HOST_LIB, [HIP_LIB]):
        subprocess.run(["g++"] + CXX_FLAGS + ["-o", HOST_LIB] + srcs +
                       ["-L" + CSRC, "-lgten_hip", "-Wl,-rpath,$ORIGIN/../csrc"], check=True)
    return HOST_LIB


def build_all(force=False):
    build_hip(force)
    build_host(force)
