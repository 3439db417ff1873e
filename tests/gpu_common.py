"""Fixtures shared by the -m gpu tests: the loaded C-ABI and the oracle."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402


@pytest.fixture(scope="session")
def hip():
    pkg = load_package()
    pkg.build.build_all()               # no-op when the in-tree .so files are newer than their sources
    api = pkg.hipabi.load()
    if api.device_count() < 1:
        pytest.fail("no GPU visible: the gten_hip path has no CPU fallback")
    api.init(0)
    return api
