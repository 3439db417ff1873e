"""not gpu: the product's synthetic-weight generator / quantizers / .gten writer
(tinyllama.cpp_amd/host/synth.h, capi.cpp) against the oracle's restatement of the
reference converter (tinyllama_to_gten.py:24-148).  Host-only code, no GPU needed."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402
from helpers import F16, MODES, Q4, Q8, tiny_config  # noqa: E402


@pytest.fixture(scope="module")
def host():
    pkg = load_package()
    pkg.build.build_all()
    return pkg.hostabi.load_host()


def host_cfg(pkg_cfg_cls, c):
    return pkg_cfg_cls(**{k: getattr(c, k) for k, _ in c._fields_})


@pytest.mark.parametrize("name,wd,ad", MODES())
def test_synthetic_weights_follow_the_converter_rules(host, oracle, name, wd, ad):
    pkg = load_package()
    cfg = host_cfg(pkg.HostConfig, tiny_config(wd, ad))
    f16cfg = host_cfg(pkg.HostConfig, tiny_config(F16, F16))
    for idx, (rows, cols, dt) in enumerate(cfg.weight_shapes()):
        got = host.synth_weight(cfg, 1234, idx)
        # the f16 form is the f32 draw rounded once; re-quantizing values that
        # are exactly representable is NOT the same as quantizing the f32 draw,
        # so rebuild the f32 draw from statistics instead: check format + stats
        if dt == F16:
            v = got.view(np.float16).astype(np.float32)
            if rows == 1:
                assert abs(v.mean() - 1.0) < 0.02 and 0.03 < v.std() < 0.07
            else:
                assert abs(v.mean()) < 2e-3 and 0.018 < v.std() < 0.022
            continue
        blk = got.reshape(rows * (cols // 32), -1)
        d = blk[:, :2].copy().view(np.float16).astype(np.float32).reshape(-1)
        if dt == Q8:
            q = blk[:, 2:].copy().view(np.int8).astype(np.int32)
            assert q.min() >= -127 and q.max() <= 127 and (np.abs(q).max(axis=1) == 127).all()
        else:
            hi = (blk[:, 2:] >> 4).astype(np.int32)
            lo = (blk[:, 2:] & 15).astype(np.int32)
            assert hi.max() <= 14 and lo.max() <= 14          # nibble 15 is never produced
            q = np.concatenate([hi, lo], axis=1) - 7
            assert (np.abs(q).max(axis=1) == 7).all()
        vals = q * d[:, None]
        assert abs(vals.mean()) < 2e-3 and 0.017 < vals.std() < 0.023
    # determinism: same seed -> same bytes, different seed -> different bytes
    a = host.synth_weight(cfg, 1234, 1)
    assert np.array_equal(a, host.synth_weight(cfg, 1234, 1))
    assert not np.array_equal(a, host.synth_weight(cfg, 1235, 1))
    del f16cfg


def test_written_gten_file_is_readable_by_the_oracle_loader(host, oracle, tmp_path):
    """the product's .gten writer (tinyllama_to_gten.py:94-201 layout) against the oracle's
    restatement of the reference loader (tinyllama.cpp:301-392): magic, record order,
    payload sizes; and the payloads are the synthetic tensors byte for byte."""
    pkg = load_package()
    for name, wd, ad in MODES():
        cfg = host_cfg(pkg.HostConfig, tiny_config(wd, ad, n_layers=1))
        path = str(tmp_path / f"tiny.{name}.gten")
        host.write_gten(cfg, 99, path)
        m = oracle.model(tiny_config(wd, ad, n_layers=1))
        m.load_gten(path)
        m.close()
        raw = open(path, "rb").read()
        assert raw[:8] == (0x454c49464e455447).to_bytes(8, "little")
        pos = 8
        for idx in range(len(cfg.weight_shapes())):
            for _ in range(2):
                n = int.from_bytes(raw[pos:pos + 4], "little"); pos += 4 + n
            nbytes = int.from_bytes(raw[pos:pos + 4], "little"); pos += 4
            want = host.synth_weight(cfg, 99, idx)
            assert nbytes == want.size
            assert raw[pos:pos + nbytes] == want.tobytes()
            pos += nbytes
        assert pos == len(raw)


def test_quantizer_equivalence_on_random_matrix(host, oracle):
    """host/synth.h::quantize_weight == oracle (== tinyllama_to_gten.py rules) on identical f32 input.
    The product exposes its quantizer only through synthetic tensors, so compare through the
    generator: regenerate the f32 draw in numpy with the same hash and feed it to the oracle."""
    pkg = load_package()
    cfg = host_cfg(pkg.HostConfig, tiny_config(Q8, Q8))
    rows, cols, _ = cfg.weight_shapes()[1]

    def mix64(z):
        z = (z + np.uint64(0x9e3779b97f4a7c15)) & np.uint64(0xffffffffffffffff)
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)) & np.uint64(0xffffffffffffffff)
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)) & np.uint64(0xffffffffffffffff)
        return z ^ (z >> np.uint64(31))

    with np.errstate(over="ignore"):
        seed, tensor = np.uint64(1234), np.uint64(1)
        base = mix64(seed * np.uint64(0x2545f4914f6cdd1d) + tensor)
        pairs = np.arange(rows * cols // 2, dtype=np.uint64)
        h = mix64(base ^ (pairs * np.uint64(0x9e3779b97f4a7c15)))
    u1 = ((h >> np.uint64(32)).astype(np.float64) + 1.0) / 4294967297.0
    u2 = ((h & np.uint64(0xffffffff)).astype(np.float64) + 0.5) / 4294967296.0
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 6.283185307179586476925 * u2
    draw = np.empty(rows * cols, np.float32)
    draw[0::2] = (rad * np.cos(ang)).astype(np.float32)
    draw[1::2] = (rad * np.sin(ang)).astype(np.float32)
    w = (np.float32(0.0) + np.float32(0.02) * draw).reshape(rows, cols)
    for wd in (Q8, Q4, F16):
        c = host_cfg(pkg.HostConfig, tiny_config(wd, Q8 if wd != F16 else F16))
        got = host.synth_weight(c, 1234, 1)
        want = oracle.quantize_weight(w, wd).reshape(-1)
        exact = float((got == want).mean())
        # identical unless libm's cos/sin/log differ in the last ulp of a draw
        assert exact > 0.9999, (wd, exact)


def test_synthetic_tokens(host):
    t = host.synthetic_tokens(64)
    assert t[0] == 1 and t[1:].min() >= 3 and t[1:].max() < 31993
    assert np.array_equal(t, host.synthetic_tokens(64))
    assert len(set(t.tolist())) > 50
