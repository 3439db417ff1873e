"""tinyllama.cpp_amd -- MI355X-native forward path behind tinyllama.cpp's gten API.

The product is native: csrc/ (HIP kernels + C-ABI, include/gten_hip.h), gten/
(C++ mirror of the reference's tensor/module/operator API) and host/ (model
driver).  The Python here only builds those libraries and binds their C-ABI for
tests and bench.py.  The directory name contains a dot, so load it with
`__graft_entry__.load_package()` rather than a plain import statement.
"""
from . import build, hipabi, hostabi  # noqa: F401
from .hipabi import F16, F32, I32, Q4, Q8, GtenHip, GtenHipError, load  # noqa: F401
from .hostabi import GtenHost, HostConfig, load_host  # noqa: F401,E402
