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


class ParityMargin(UserWarning):
    """a long-context parity margin: raised as a WARNING so that the summary of `pytest -q` (the driver's GPU log) carries the
    numbers, not only a pass / fail"""


def record_margin(what, rms, own_rms, mx, extra=""):
    """one long-context comparison with the reference's goldens: `rms` against the reference's own AVX-vs-scalar spread `own_rms`
    at that length (the yardstick of DESIGN.md section 5: bar 1.35 x, watch line 1.25 x) -- appended to
    gpurun_out/parity_margins_tests.txt (scratch on the GPU box, merged back by gpurun) and emitted as a warning"""
    import warnings
    x = rms / own_rms if own_rms > 0 else float("inf")
    line = f"{what}: rms {rms:.4f} = {x:.2f} x the reference's own spread {own_rms:.4f}, max {mx:.4f}{' ' + extra if extra else ''}"
    if x > 1.25:
        line += "   <-- above 1.25 x"
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_margins_tests.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    warnings.warn(ParityMargin(line))
    return x
