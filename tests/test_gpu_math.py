"""Accuracy of the kernels' own fp64 elementary functions (bssm_amd/csrc/pf_math.h) against libm,
and of the Box-Muller stream against the oracle's libm-based one."""
import ctypes as C

import numpy as np
import pytest

from bssm_amd import _lib

pytestmark = pytest.mark.gpu


def _run(fn, x, mult=1):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(x.size * mult)
    _lib.check(_lib.lib().bssm_debug_math(fn, _lib.dptr(x), _lib.dptr(out), x.size))
    return out


def _ulps(a, b):
    return np.abs(a - b) / np.spacing(np.abs(b))


def test_fast_exp():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-745, 709.7, 200000), rng.uniform(-2, 2, 200000), rng.normal(0, 1e-8, 1000),
                        [0.0, -0.0, 1.0, -1.0, 709.78, -745.1, 800.0, -800.0, 1e5, -1e5, np.inf, -np.inf]])
    got = _run(0, x)
    ref = np.exp(x)
    fin = np.isfinite(ref) & (ref > 1e-300)
    assert _ulps(got[fin], ref[fin]).max() <= 2.0
    assert np.all(got[ref == np.inf] == np.inf) and np.all(got[ref == 0] == 0)
    assert np.isnan(_run(0, np.array([np.nan]))[0])


def test_fast_log():
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(-700, 700, 200000)), rng.uniform(0.5, 2.0, 200000), 1 + rng.normal(0, 1e-9, 1000),
                        (np.arange(1, 2001) * 2.0 ** -53), [1.0, 2.0, 0.5, 5e-324, 1e-310, np.inf, 0.0]])
    got = _run(1, x)
    with np.errstate(divide="ignore"):
        ref = np.log(x)
    fin = np.isfinite(ref) & (ref != 0)
    # table-driven log: <= 1 ulp except next to the interval around 1 (x just below 1 - 2^-9), where log c and r cancel
    # to half their size: 2.1 ulp worst case (scripts/ubench/math_check.cpp)
    assert _ulps(got[fin], ref[fin]).max() <= 2.5
    assert got[x == 1.0][0] == 0.0 and got[-2] == np.inf and got[-1] == -np.inf
    assert np.isnan(_run(1, np.array([-1.0]))[0])


def test_fast_sincos_2pi():
    rng = np.random.default_rng(2)
    bits = np.concatenate([rng.integers(0, 2 ** 32, 400000), np.arange(8) * 2 ** 29, np.arange(1, 9) * 2 ** 29 - 1,
                           np.arange(8) * 2 ** 29 + 1, [0, 1, 2 ** 32 - 1]]).astype(np.uint64)
    u = bits.astype(np.float64) * 2.0 ** -32
    got = _run(2, u, mult=2).reshape(-1, 2)
    # reference in extended precision: angle = 2 pi u exactly via long double
    ang = (np.longdouble(6.283185307179586476925286766559) * u.astype(np.longdouble))
    rs, rc = np.sin(ang).astype(np.float64), np.cos(ang).astype(np.float64)
    # theta = 2 pi frac is rounded once (0.5 ulp of an angle <= pi/4) and the fdlibm kernels add < 1 ulp:
    # 3.4e-16 absolute worst case; libm's sin(2*pi*u) on the ROUNDED double angle (the oracle) is off by up to 4.4e-16
    assert np.abs(got[:, 0] - rs).max() <= 4e-16
    assert np.abs(got[:, 1] - rc).max() <= 4e-16
    assert np.abs(got[:, 0] ** 2 + got[:, 1] ** 2 - 1).max() <= 5e-16


def test_box_muller_stream_matches_oracle(oracle):
    n = 50000
    got = _run(3, np.zeros(n), mult=2).reshape(-1, 2)
    L = oracle.lib()
    rng, _ = oracle.make_rng(seed=1, stream=0)
    z0, z1 = C.c_double(), C.c_double()
    ref = np.zeros((n, 2))
    for i in range(n):
        L.orc_rng_normal_pair(C.byref(rng), 2, 0, i, 0, C.byref(z0), C.byref(z1))
        ref[i] = (z0.value, z1.value)
    assert np.abs(got - ref).max() <= 4e-15      # |z| <= 8.6; a few ulp of the radius
    assert np.abs(got - ref).mean() <= 3e-16


def test_packed_normals_and_shared_uniforms_match_oracle(oracle):
    """round 4: draws of more than two normals pack two Box-Muller pairs into a Philox block (40-bit radius, 24-bit angle),
    and the resampling uniforms of such a filter come from blocks shared by four consecutive slots (pf_device.h)"""
    n = 50000
    got = _run(4, np.zeros(n), mult=4).reshape(-1, 4)
    L = oracle.lib()
    rng, _ = oracle.make_rng(seed=1, stream=0)
    ref = np.zeros((n, 4))
    L.orc_rng_normals_bulk.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
    L.orc_rng_normals_bulk(C.byref(rng), 2, 0, 0, n, 4, ref.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.abs(got - ref).max() <= 4e-15      # |z| <= 7.45; a few ulp of the radius
    assert np.abs(got - ref).mean() <= 3e-16
    gu = _run(5, np.zeros(n), mult=2)[:n]
    ru = np.array([L.orc_rng_uniform(C.byref(rng), 1, 3, i, 4) for i in range(n)])
    assert np.array_equal(gu, ru)


def test_wave_max_key_is_the_exact_maximum():
    """wave_max_key (pf_device.h): the maximum of a wave's 64 doubles through two 32-bit DPP reductions on an order-preserving
    integer key -- exact for every sign / magnitude mix, -inf rows, signed zeros, denormals and equal high words"""
    rng = np.random.default_rng(5)
    blocks = [rng.normal(0, 1e3, 64), -np.abs(rng.normal(0, 1e-3, 64)), np.full(64, -np.inf), rng.normal(0, 1, 64) * 1e-310,
              -1.0 - rng.integers(0, 2 ** 30, 64) * 2.0 ** -52, 1.0 + rng.integers(0, 2 ** 30, 64) * 2.0 ** -52,
              np.where(rng.random(64) < 0.5, -np.inf, rng.normal(-700, 30, 64)), np.r_[np.full(63, -5.0), 1e308], np.r_[0.0, np.full(63, -0.0)]]
    for _ in range(23):
        blocks.append(rng.normal(0, 1, 64) * 10.0 ** rng.integers(-300, 300))
    x = np.concatenate(blocks)
    got = _run(6, x)
    want = np.repeat(x.reshape(-1, 64).max(axis=1), 64)
    assert np.array_equal(got, want)
