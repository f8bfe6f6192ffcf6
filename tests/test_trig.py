"""The contract's fp32 trig (include/sigma_trig_f32.h) on the host side, through the oracle library.

sin / cos / tan / atan == (float)libm(double) -- the correctly rounded value -- on every sampled argument, and within one ulp of what
torch computed in the build container (tests/golden/trig_f32.npz; torch's vector math is not correctly rounded).
"""
import os

import numpy as np
import pytest

import oracle_binding as ob

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trig_f32.npz")


def _call(lib, kind, a, b=None):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b if b is not None else a, np.float32)
    out = np.empty_like(a)
    lib.fn_trig(kind, a.size, ob.ptr(a), ob.ptr(b), ob.ptr(out))
    return out


def _ulp_diff(a, b):
    def key(v):
        i = v.view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)
    return np.abs(key(a) - key(b))


@pytest.fixture(scope="module")
def lib():
    return ob.load_oracle()


@pytest.mark.parametrize("kind,name,npf", [(0, "sin", np.sin), (1, "cos", np.cos), (2, "tan", np.tan), (3, "atan", np.arctan)])
def test_unary_is_correctly_rounded_and_within_one_ulp_of_torch(lib, kind, name, npf):
    z = np.load(GOLDEN)
    x = z["x"]
    got = _call(lib, kind, x)
    want = npf(x.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert _ulp_diff(got, z[name]).max() <= 1
    # a wider sweep against libm only: all magnitudes up to 2^29, signed zeros, the arctangent's interval boundaries
    rng = np.random.default_rng(kind)
    w = rng.integers(0, 0x4E000000, 1 << 20, dtype=np.uint32).view(np.float32) * rng.choice([-1.0, 1.0], 1 << 20).astype(np.float32)
    w = np.concatenate([w, np.array([0.0, -0.0, 0.4375, 0.6875, 1.1875, 2.4375, np.pi, -np.pi / 2], np.float32)])
    got = _call(lib, kind, w)
    want = npf(w.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
