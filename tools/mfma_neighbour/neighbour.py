"""Loader of the test-only neighbour kernels (neighbour.hip): nb_init / nb_run(kind, launches, grid, spin) / nb_sync."""
import ctypes
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "neighbour.hip")
LIB = os.path.join(HERE, "libneighbour.so")


def load_neighbour():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", LIB, SRC], check=True)
    nb = ctypes.CDLL(LIB)
    nb.nb_run.argtypes = [ctypes.c_int] * 4
    for f in (nb.nb_init, nb.nb_run, nb.nb_sync):
        f.restype = ctypes.c_int
    if nb.nb_init() != 0:
        raise RuntimeError("nb_init failed")
    return nb
