"""not gpu: the C-ABI library builds for gfx950, loads, and exports every symbol
include/gten_hip.h declares (no compute calls: there is no GPU here)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402


def declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gten_(?:hip|host)_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    pkg = load_package()
    pkg.build.build_hip()
    api = pkg.hipabi.GtenHip()
    names = declared_symbols("gten_hip.h")
    assert len(names) >= 20
    for name in names:
        assert hasattr(api.lib, name), f"{name} declared in include/gten_hip.h but not exported"
    assert sorted(api.SYMBOLS) == names, "python binding out of sync with the header"


def test_host_library_exports_every_declared_symbol():
    pkg = load_package()
    pkg.build.build_all()
    host = pkg.hostabi.GtenHost()
    names = declared_symbols("gten_host.h")
    for name in names:
        assert hasattr(host.lib, name), f"{name} declared in include/gten_host.h but not exported"
    assert sorted(host.SYMBOLS) == names


def test_no_silent_cpu_fallback():
    """without a GPU the product path must fail loudly, not compute on the CPU"""
    pkg = load_package()
    api = pkg.hipabi.GtenHip()
    if api.device_count() > 0:
        return
    import pytest
    with pytest.raises(pkg.GtenHipError):
        api.init(0)
    with pytest.raises(pkg.GtenHipError):
        api.alloc(64)


def test_watched_cache_registry_selftest():
    """the registry behind "a head-major K / V shadow cannot be stale" (csrc/gten_rt.h kv_watch_*, include/gten_hip.h
    gten_hip_set_kv_head_major): which writes hit which watched rows, whose flag they set, what slot_bind's re-registration
    leaves -- host-only logic, exercised without a GPU"""
    pkg = load_package()
    api = pkg.hipabi.GtenHip()
    assert api.kv_watch_selftest() == 0
