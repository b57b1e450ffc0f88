"""The plain-float acos / atan2 / sin / cos shared by the kernels and (on request) the oracle
(cudaraytracer_amd/csrc/det_math.h) against the C library in double precision: they stand in for
the reference's acosf / atan2f / sinf / cosf calls (hittable.cuh:183-199, render.cuh:78-80), so
they must be as accurate as a single-precision math library (a few ulp), and exact at the special
points the (u, v) formulas hit."""
import math

import numpy as np
import pytest

import oracle_lib


@pytest.fixture(scope="module")
def lib():
    return oracle_lib.load()


def _ulp_error(got, want64):
    want32 = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    ulp = np.maximum(ulp, np.finfo(np.float32).tiny)
    return np.abs(got.astype(np.float64) - want64) / ulp


def test_acos(lib):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-1, 1, 200000), np.linspace(-1, 1, 20001), [-1.0, -0.5, 0.0, 0.5, 1.0]]).astype(np.float32)
    got = np.array([lib.oracle_det_acos(float(v)) for v in x], np.float32)
    err = _ulp_error(got, np.arccos(x.astype(np.float64)))
    assert err.max() <= 4.0, (err.max(), x[err.argmax()])
    assert lib.oracle_det_acos(1.0) == 0.0
    assert lib.oracle_det_acos(-1.0) == np.float32(math.pi)
    assert math.isnan(lib.oracle_det_acos(1.5))


def test_atan2(lib):
    rng = np.random.default_rng(2)
    n = 200000
    y = rng.normal(size=n).astype(np.float32)
    x = rng.normal(size=n).astype(np.float32)
    x[:1000] *= 1e-6
    y[1000:2000] *= 1e-6
    got = np.array([lib.oracle_det_atan2(float(a), float(b)) for a, b in zip(y, x)], np.float32)
    err = _ulp_error(got, np.arctan2(y.astype(np.float64), x.astype(np.float64)))
    assert err.max() <= 4.0, (err.max(), y[err.argmax()], x[err.argmax()])
    pi = np.float32(math.pi)
    assert lib.oracle_det_atan2(0.0, 1.0) == 0.0
    assert lib.oracle_det_atan2(0.0, -1.0) == pi
    assert lib.oracle_det_atan2(-0.0, -1.0) == -pi
    assert lib.oracle_det_atan2(1.0, 0.0) == np.float32(math.pi / 2)
    assert lib.oracle_det_atan2(-1.0, 0.0) == -np.float32(math.pi / 2)
    assert lib.oracle_det_atan2(0.0, 0.0) == 0.0
    assert math.isnan(lib.oracle_det_atan2(float("nan"), 1.0))


def test_sin_cos(lib):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(0, 2 * math.pi, 200000), rng.uniform(-50, 50, 50000),
                        [0.0, math.pi / 2, math.pi, 3 * math.pi / 2, 2 * math.pi]]).astype(np.float32)
    s = np.array([lib.oracle_det_sin(float(v)) for v in x], np.float32)
    c = np.array([lib.oracle_det_cos(float(v)) for v in x], np.float32)
    # near a zero of the function a float library is judged by absolute error (the argument's own rounding dominates)
    es = np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64)))
    ec = np.abs(c.astype(np.float64) - np.cos(x.astype(np.float64)))
    assert es.max() <= 2.5e-7 and ec.max() <= 2.5e-7, (es.max(), ec.max())
    assert lib.oracle_det_sin(0.0) == 0.0 and lib.oracle_det_cos(0.0) == 1.0
