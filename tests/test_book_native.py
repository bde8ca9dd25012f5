"""The native book compilers (csrc/book_host.cpp; host threads, no GPU) against the NumPy forms they replace and, through
those, the object path (tests/test_schedule_np.py checks the NumPy forms date for date against `Schedule` /
`SwapFloatLeg`): coupon schedules of many legs, and the foreign-leg batches of a cross-currency book - bit for bit."""
import dataclasses

import numpy as np
import pytest

from adrates_amd import _native
from adrates_amd.market.position import xccy_engine as XE
from adrates_amd.trades import synthetic_xccy as SX
from adrates_amd.utils import BusDayAdjustTypes, DayCountTypes, schedule_np as S
from adrates_amd.utils.error import LibError

from . import _fixtures as F


def _random_legs(n, seed):
    rng = np.random.default_rng(seed)
    eff = rng.integers(36526, 55000, n)                                   # 2000 .. 2050, weekends and month ends included
    eff[: n // 8] = S.serial_of(rng.integers(2000 * 12, 2050 * 12, n // 8), np.full(n // 8, 31))       # month ends
    months = rng.choice([1, 2, 3, 5, 6, 9, 12, 18, 24, 60, 120, 361, 600], n)
    y, m, d = S.ymd_from_serial(eff)
    term = S.serial_of(y * 12 + m - 1 + months, d) + rng.integers(0, 3, n) * rng.integers(0, 2, n)     # some stubs
    mpp = rng.choice([1, 3, 6, 12], n)
    lag = rng.choice([0, 0, 1, 2, 5, -2], n)
    den = rng.choice([365.0, 360.0], n)
    return eff, term, mpp, lag, den


@pytest.mark.parametrize("bd", list(BusDayAdjustTypes))
@pytest.mark.parametrize("weekend", [True, False])
def test_native_leg_times_equal_the_array_form(bd, weekend):
    eff, term, mpp, lag, den = _random_legs(20011, seed=bd.value + 10 * weekend)
    value_serial = 44000
    for pay_den in (None, 365):
        want = S.leg_times_np(eff, term, mpp, lag, bd, weekend, den, value_serial, pay_den)
        got = S.leg_times(eff, term, mpp, lag, bd, weekend, den, value_serial, pay_den)
        assert len(got) == len(want) == 6
        for a, b in zip(got, want):
            assert a.dtype == b.dtype and np.array_equal(a, b)
    assert got[5].sum() > 0.9 * got[5].size
    if weekend and bd == BusDayAdjustTypes.PRECEDING:
        assert not got[5].all()          # a stub date rolled back onto the effective date: the reference's de-duplication applies


def test_native_leg_times_edge_cases():
    bd = BusDayAdjustTypes.FOLLOWING
    off, tp, ts, te, al, plain = S.leg_times(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64),
                                             np.zeros(0, dtype=np.int64), bd, True, np.zeros(0), 44000)
    assert off.tolist() == [0] and tp.size == 0 and plain.size == 0
    with pytest.raises(LibError, match="before termination"):
        S.leg_times(np.array([45000]), np.array([45000]), np.array([12]), np.array([0]), bd, True, np.array([365.0]), 44000)
    with pytest.raises(LibError, match="1900|range|supported"):
        S.leg_times(np.array([40]), np.array([45000]), np.array([12]), np.array([0]), bd, True, np.array([365.0]), 44000)
    with pytest.raises(LibError):
        S.leg_times(np.array([45000]), np.array([45000 + 400 * 366]), np.array([12]), np.array([0]), bd, True, np.array([365.0]), 44000)
    # one period shorter than the frequency: a single coupon from the effective date to the adjusted termination date
    off, tp, ts, te, al, plain = S.leg_times(np.array([45292]), np.array([45299]), np.array([12]), np.array([2]), bd, True,
                                             np.array([360.0]), 45292)
    want = S.leg_times_np(np.array([45292]), np.array([45299]), np.array([12]), np.array([2]), bd, True, np.array([360.0]), 45292)
    assert off.tolist() == [0, 1] and all(np.array_equal(a, b) for a, b in zip((off, tp, ts, te, al, plain), want))


@pytest.mark.parametrize("n,seed", [(1, 1), (257, 2), (12001, 3)])
def test_native_xccy_assembly_equals_the_array_form(n, seed):
    vd = F.README_VALUE_DT
    terms, _ = SX.draw_terms(vd, n, seed=seed) if "seed" in SX.draw_terms.__code__.co_varnames else SX.draw_terms(vd, n)
    raw = XE.raw_from_terms(terms, vd, DayCountTypes.ACT_365F)
    # put some flows exactly at the value time and before it, and switch some exchanges off
    rng = np.random.default_rng(seed)
    raw.for_tpx = raw.for_tpx.copy()
    pick = rng.random(raw.for_tpx.size) < 0.03
    raw.for_tpx[pick] = rng.choice([0.0, -0.25], int(pick.sum()))
    raw.for_exch_t = raw.for_exch_t.copy()
    raw.for_exch_t[rng.random(n) < 0.2, 0] = 0.0
    raw.for_exch_t[rng.random(n) < 0.05, 0] = -0.1
    raw.for_exch = rng.random(n) < 0.8
    raw.for_al = raw.for_al.copy()
    raw.for_al[rng.random(raw.for_al.size) < 0.01] = 0.0
    df_x = lambda t: np.exp(-0.031 * np.asarray(t, dtype=np.float64) - 0.0004 * np.asarray(t, dtype=np.float64) ** 2)
    df_f = lambda t: np.exp(-0.043 * np.asarray(t, dtype=np.float64))
    got = XE.compile_xccy(raw, 1.2731, df_x, df_f)
    want = XE.compile_xccy_np(raw, 1.2731, df_x, df_f)
    for a, b in zip(got[:3], want[:3]):
        for f in dataclasses.fields(a):
            x, y = getattr(a, f.name), getattr(b, f.name)
            if x is None or y is None:
                assert x is None and y is None, f.name
            else:
                assert np.array_equal(np.asarray(x), np.asarray(y)), f.name
    assert np.array_equal(got[3], want[3])


def test_library_exports_the_book_compilers():
    lib = _native.load()
    for name in ("adr_leg_counts_host", "adr_leg_times_host", "adr_xccy_assemble_host"):
        assert hasattr(lib, name)
