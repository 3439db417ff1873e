import os
import sys

import pytest

# The checker libraries use OpenMP over output features like the reference
# (gten/ops.h:635-637); on a many-core GPU host hundreds of threads on tiny test
# matrices cost far more than they give, so bound them before libgomp loads.
os.environ.setdefault("OMP_NUM_THREADS", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import orc
    orc.build(ref=os.path.isdir("/root/reference"))
    return orc.load_oracle()


@pytest.fixture(scope="session", params=["avx", "scalar"])
def ref_pair(request, oracle):
    """(oracle configured for the matching summation order, real reference build) or skip."""
    from oracle import orc
    ref = orc.load_ref(request.param)
    if ref is None:
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    oracle.set_simd(request.param == "avx")
    yield oracle, ref
    oracle.set_simd(True)
