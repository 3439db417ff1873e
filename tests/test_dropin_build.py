"""not gpu: the reference's unmodified tinyllama.cpp compiles against this repository's gten
headers (the API is intact).  Only where /root/reference exists."""
import os

import pytest


def test_reference_translation_unit_compiles_against_our_gten_headers():
    if not os.path.isfile("/root/reference/tinyllama.cpp"):
        pytest.skip("/root/reference not present")
    from __graft_entry__ import load_package
    pkg = load_package()
    pkg.build.build_hip()
    from oracle import orc
    path = os.path.join(orc.HERE, "_ref", "libdropin.so")
    if os.path.exists(path):
        os.remove(path)
    orc.build_dropin()
    assert os.path.exists(path)
