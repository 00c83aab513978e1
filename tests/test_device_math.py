"""Accuracy of the step kernel's own math routines (highwayenv_amd/csrc/hwy_math.h) on their stated
domains, against numpy, in ulps.  `emu` runs the same source on the CPU (seed = exact 1/x there);
`hip` runs it on the MI355X through hwy_debug_math (real v_rcp_f64 / v_rsq_f64 seeds)."""
import numpy as np
import pytest

from highwayenv_amd import _abi
from tests.backends import BACKENDS, make_engine

LOG, EXP, SIN, COS, ASIN, RCP, RSQRT, WRAP = range(8)


def ulps(got, want):
    want = np.asarray(want, np.float64)
    return np.abs(got - want) / np.spacing(np.abs(want))


@pytest.fixture(params=BACKENDS)
def eng(request):
    e = make_engine(request.param, _abi.make_config(_abi.highway_fast_default_config(), 1, fast=True))
    yield e
    e.close()


def test_log_exp(eng):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(1e-3, 2.5, 200000), 10.0 ** rng.uniform(-60, 2, 50000), [1.0, 0.5, 2.0, 0.7071067811865476]])
    assert ulps(eng.debug_math(LOG, x), np.log(x)).max() <= 2.0
    y = np.concatenate([rng.uniform(-5, 4, 200000), rng.uniform(-690, 40, 50000), [0.0, -0.0, 1.0]])
    assert ulps(eng.debug_math(EXP, y), np.exp(y)).max() <= 2.0
    assert (eng.debug_math(EXP, np.array([-701.0, -1e9, -np.inf])) == 0).all()
    # the composite the kernel uses: (v/v0)^delta = exp(delta*log r)
    r, d = rng.uniform(0.05, 1.8, 100000), rng.uniform(3.5, 4.5, 100000)
    got = eng.debug_math(EXP, d * eng.debug_math(LOG, r))
    assert (np.abs(got - np.power(r, d)) / np.power(r, d)).max() < 2e-15


def test_sincos(eng):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-1.6, 1.6, 200000), rng.uniform(-50, 50, 100000), rng.uniform(-1e5, 1e5, 20000),
                        [0.0, -0.0, np.pi / 4, -np.pi / 2, np.pi]])
    s, c = eng.debug_math(SIN, x), eng.debug_math(COS, x)
    # absolute error bounded by ~1 ulp of 1 (relative ulps blow up at the zeros of sin/cos for any libm)
    assert np.abs(s - np.sin(x)).max() < 2.5e-16 * (1 + np.abs(x).max() * 1e-6)
    assert np.abs(c - np.cos(x)).max() < 2.5e-16 * (1 + np.abs(x).max() * 1e-6)
    small = np.abs(x) < 0.78
    assert ulps(s[small], np.sin(x[small]))[x[small] != 0].max() <= 2.0
    assert ulps(c[small], np.cos(x[small])).max() <= 2.0
    assert np.abs(s * s + c * c - 1).max() < 5e-16


def test_asin_rcp_rsqrt_wrap(eng):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-0.5, 0.5, 200000), rng.uniform(-1, 1, 100000), [0.0, 0.5, -0.5, 0.7071067811865476, 1.0, -1.0]])
    u = ulps(eng.debug_math(ASIN, x), np.arcsin(x))
    assert u[np.abs(x) <= 0.5][x[np.abs(x) <= 0.5] != 0].max() <= 2.0 and u[x != 0].max() <= 4.0
    v = np.concatenate([rng.uniform(0.01, 50, 100000), -rng.uniform(0.01, 50, 100000), 10.0 ** rng.uniform(-8, 8, 50000)])
    assert ulps(eng.debug_math(RCP, v), 1.0 / v).max() <= 1.5
    w = np.concatenate([rng.uniform(1e-12, 4, 100000), 10.0 ** rng.uniform(-20, 20, 50000)])
    assert ulps(eng.debug_math(RSQRT, w), 1.0 / np.sqrt(w)).max() <= 2.0
    a = np.concatenate([rng.uniform(-10, 10, 100000), rng.uniform(-1e4, 1e4, 20000), [np.pi, -np.pi, 0.0, 3 * np.pi]])
    ref = ((a + np.pi) % (2 * np.pi)) - np.pi  # utils.wrap_to_pi
    assert np.abs(eng.debug_math(WRAP, a) - ref).max() < 4e-12  # |a| up to 1e4: one ulp of a + pi
    inside = np.abs(a) < 3
    assert (eng.debug_math(WRAP, a[inside]) == ref[inside]).all()  # no reduction needed: bit-identical


def test_atan_atan2(eng):
    """fdlibm-style atan / atan2 of the intersection kernel (CircularLane.local_coordinates, lane.py:355-362)."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-3, 3, 200000), 10.0 ** rng.uniform(-12, 6, 50000), -(10.0 ** rng.uniform(-12, 6, 50000)),
                        [0.0, 0.4375, 0.6875, 1.1875, 2.4375, 1.0, -1.0]])
    u = ulps(eng.debug_math(8, x), np.arctan(x))
    assert u[x != 0].max() <= 2.0
    y = np.concatenate([rng.uniform(-40, 40, 200000), 10.0 ** rng.uniform(-9, 3, 20000), [0.0, 0.75, -0.75]])
    assert np.abs(eng.debug_math(9, y) - np.arctan2(y, 0.75)).max() < 4.5e-16       # |result| < pi/2: < 2 ulp of pi/2
    assert ulps(eng.debug_math(9, y), np.arctan2(y, 0.75))[y != 0].max() <= 2.5
    for op, yy in ((10, 0.5), (11, -0.5)):
        got = eng.debug_math(op, y)
        assert np.abs(got - np.arctan2(yy, y)).max() < 9e-16                       # |result| <= pi: 2 ulp of pi


def test_paired_routines_are_bit_identical_to_the_scalar_ones(eng):
    """hwy_math.h's paired forms (log_pos2, exp_bounded2, sincos_bounded2, asin_bounded2, fast_rcp2, fast_rsqrt2: the two vehicles
    of a thread in hwy_wave2.h share every coefficient's register pair, one asm statement per Horner step) run the same
    operations on the same values as two scalar calls: compared bit for bit, in both positions of the pair."""
    rng = np.random.default_rng(4)
    pos = np.concatenate([rng.uniform(1e-3, 2.5, 50000), 10.0 ** rng.uniform(-30, 2, 10000), [1.0, 0.5, 2.0]])
    ey = np.concatenate([rng.uniform(-5, 4, 50000), rng.uniform(-690, 40, 10000), [0.0, -701.0, -np.inf]])
    ang = np.concatenate([rng.uniform(-1.6, 1.6, 50000), rng.uniform(-50, 50, 10000), [0.0, np.pi / 4]])
    unit = np.concatenate([rng.uniform(-1, 1, 50000), [0.0, 0.5, -0.5, 1.0, -1.0, 0.7071067811865476]])
    for first, second, scalar, x in ((20, 21, LOG, pos), (22, 23, EXP, ey), (28, 29, ASIN, unit), (30, 31, RCP, pos), (32, 33, RSQRT, pos)):
        want = eng.debug_math(scalar, x)
        np.testing.assert_array_equal(eng.debug_math(first, x), want, err_msg=f"op {first}")
        np.testing.assert_array_equal(eng.debug_math(second, x), want, err_msg=f"op {second}")
    s, c = eng.debug_math(SIN, ang), eng.debug_math(COS, ang)
    for op, want in ((24, s), (25, c), (26, s), (27, c)):
        np.testing.assert_array_equal(eng.debug_math(op, ang), want, err_msg=f"op {op}")


def test_reach_key_rounds_up(eng):
    """hwy_device.h reach_key / reach_from_keys: the double a high-word key is rounded up to is never below the value it stands for
    (a bound too small would let the forward collision walk stop before a partner inside the pre-check sphere)."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-60, 60, 20000), 10.0 ** rng.uniform(-300, 6, 20000), [0.0, -0.0, 2.0, 1e-310, 5e-324]])
    up = eng.debug_math(41, x)
    assert (up >= np.abs(x)).all()
    big = np.abs(x) > 1e-300
    assert (up[big] <= np.abs(x[big]) * (1 + 2.0 ** -19)).all()  # (the high word keeps 20 bits of the mantissa)


@pytest.mark.gpu
def test_wave_max_u32_on_the_device():
    """hwy_device.h wave_max_u32 (six DPP v_max_u32 + v_readlane): every lane of a wavefront gets the maximum of the 64 keys --
    whole wavefronts of random values, the maximum planted in every lane position once."""
    from highwayenv_amd.engine import Engine
    e = Engine(_abi.make_config(_abi.highway_fast_default_config(), 1, fast=True))
    rng = np.random.default_rng(4)
    x = rng.uniform(0.0, 50.0, (128, 64))
    for lane in range(64):
        x[lane, lane] = 100.0 + lane
    x[64:, :] *= 10.0 ** rng.integers(-8, 3, (64, 1))
    got = e.debug_math(40, x.ravel()).reshape(x.shape)
    key = (np.abs(x).view(np.uint64) >> 32).astype(np.float64)
    assert (got == key.max(axis=1, keepdims=True)).all()
    e.close()
