"""not gpu: the host-side C++ (tokenizer, synthetic weights + quantizers, .gten writer; host/capi.cpp and the headers it
pulls in) compiled with -fsanitize=address,undefined and run on the CPU (tests/host_sanitize.cpp).  No GPU call is made:
the binary links libgten_hip.so only because capi.cpp references it."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    from __graft_entry__ import load_package
    pkg = load_package()
    pkg.build.build_hip()
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_sanitize")
    csrc = os.path.join(ROOT, "tinyllama.cpp_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tinyllama.cpp_amd"),
           os.path.join(ROOT, "tests", "host_sanitize.cpp"), os.path.join(ROOT, "tinyllama.cpp_amd", "host", "capi.cpp"),
           "-o", exe, "-L" + csrc, "-lgten_hip", "-Wl,-rpath," + csrc]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-4000:])
    assert "host_sanitize ok" in r.stdout


def test_modules_and_scheduler_under_sanitizers_against_a_stub_device(tmp_path):
    """The host C++ ABOVE the HIP C-ABI -- gten modules (recorded single-row chains, the composed block call), TinyLlama /
    TinyLlamaBatch, the continuous-batching scheduler, capi.cpp -- under ASan + UBSan, linked against tests/hip_stub.cpp (a
    test-only stand-in for libgten_hip.so on host memory whose 'model' is a fixed next-id rule).  tests/host_sanitize_serve.cpp
    checks that the reference's loop, the device sampler, the fixed batch and the queue through the slots (4 admission
    schedules x 3 slice lengths, per-prompt budgets, 2 and 8 slots) all return the same ids."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_sanitize_serve")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tinyllama.cpp_amd"),
           os.path.join(ROOT, "tests", "host_sanitize_serve.cpp"), os.path.join(ROOT, "tests", "hip_stub.cpp"),
           os.path.join(ROOT, "tinyllama.cpp_amd", "host", "capi.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-4000:])
    assert "host_sanitize_serve ok" in r.stdout


def test_stub_device_covers_the_whole_header(tmp_path):
    """tests/hip_stub.cpp stands in for EVERY symbol include/gten_hip.h declares (so that new entry points cannot silently
    fall out of the sanitizer run), and only tests/ refers to it"""
    import ctypes
    import re
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    so = str(tmp_path / "libhip_stub.so")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "hip_stub.cpp"), "-o", so], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = ctypes.CDLL(so)
    text = open(os.path.join(ROOT, "include", "gten_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(gten_hip_[a-z0-9_]+)\s*\(", text)))
    assert len(names) >= 40
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in include/gten_hip.h but tests/hip_stub.cpp does not define it"
    # the product never refers to the stub
    for base, _, files in os.walk(os.path.join(ROOT, "tinyllama.cpp_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                assert "hip_stub" not in open(os.path.join(base, f), errors="replace").read(), os.path.join(base, f)
    for f in ("bench.py", "__graft_entry__.py"):
        assert "hip_stub" not in open(os.path.join(ROOT, f)).read()
