"""Deterministic synthetic OIS portfolios for benchmarks and large-size tests.

Specification: SURVEY.md section 8(d) - spot-starting swaps, maturity in whole
months U{1..360} ("offgrid": annual coupons with a front stub, so most cash-flow
times fall between curve knots) or whole years U{1..30} ("ongrid": every time
snaps to a knot), coupon U(0.01, 0.07), notional round(U(1e6, 5e7), -5),
pay/receive 50/50, float spread 0, both legs ANNUAL on the curve's day count,
MODIFIED_FOLLOWING on the WEEKEND calendar, ``numpy.random.default_rng(seed)``.

Every distinct maturity has one schedule, so the schedule of each maturity is
built once through the ordinary `OIS(...)` constructor and the per-trade arrays
are assembled from those templates with vectorised NumPy - the result is
bit-identical to compiling `OIS` objects one by one (tests/test_synthetic.py).
"""
from __future__ import annotations

import numpy as np

from ..utils.calendar import BusDayAdjustTypes
from ..utils.currency import CurrencyTypes
from ..utils.day_count import DayCountTypes
from ..utils.frequency import FrequencyTypes
from ..utils.global_types import CurveTypes, SwapTypes
from .compiler import TradeBatch, compile_ois
from .rates.ois import OIS

DEFAULT_SEED = 20240430


def _make_swap(value_dt, tenor, coupon, notional, pay_fixed, dc_type, curve_type, currency, freq):
    return OIS(effective_dt=value_dt, term_dt_or_tenor=tenor,
               fixed_leg_type=SwapTypes.PAY if pay_fixed else SwapTypes.RECEIVE,
               fixed_coupon=float(coupon), fixed_freq_type=freq, fixed_dc_type=dc_type,
               floating_index=curve_type, currency=currency, notional=float(notional),
               bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING, float_freq_type=freq, float_dc_type=dc_type)


def draw_terms(n, kind="offgrid", seed=DEFAULT_SEED):
    """Random economic terms: maturity (months), coupon, notional, pay-fixed flag."""
    rng = np.random.default_rng(seed)
    if kind == "offgrid":
        months = rng.integers(1, 361, size=n)
    elif kind == "ongrid":
        months = 12 * rng.integers(1, 31, size=n)
    else:
        raise ValueError("kind must be 'offgrid' or 'ongrid'")
    coupon = rng.uniform(0.01, 0.07, size=n)
    notional = np.round(rng.uniform(1e6, 5e7, size=n), -5)
    pay_fixed = rng.random(n) < 0.5
    return months, coupon, notional, pay_fixed


def swaps_from_terms(value_dt, months, coupon, notional, pay_fixed, dc_type=DayCountTypes.ACT_365F,
                     curve_type=CurveTypes.GBP_OIS_SONIA, currency=CurrencyTypes.GBP,
                     freq=FrequencyTypes.ANNUAL):
    """The same trades as `OIS` objects (slow path; used to validate the fast one)."""
    return [_make_swap(value_dt, f"{int(m)}M", c, nn, bool(p), dc_type, curve_type, currency, freq)
            for m, c, nn, p in zip(months, coupon, notional, pay_fixed)]


def synthesize(value_dt, n, kind="offgrid", seed=DEFAULT_SEED, dc_type=DayCountTypes.ACT_365F,
               curve_type=CurveTypes.GBP_OIS_SONIA, currency=CurrencyTypes.GBP,
               freq=FrequencyTypes.ANNUAL) -> TradeBatch:
    """Batch of ``n`` synthetic trades as arrays, without creating n objects."""
    months, coupon, notional, pay_fixed = draw_terms(n, kind, seed)

    # schedule templates: unit notional and unit coupon give the accrual fractions
    uniq = np.unique(months)
    tmpl = {}
    for m in uniq:
        b = compile_ois([_make_swap(value_dt, f"{int(m)}M", 1.0, 1.0, True, dc_type, curve_type, currency, freq)],
                        value_dt)
        tmpl[int(m)] = b
    n_fix = np.array([tmpl[int(m)].fix_tp.shape[0] for m in uniq])
    n_flt = np.array([tmpl[int(m)].flt_tp.shape[0] for m in uniq])
    pos = np.searchsorted(uniq, months)

    def gather(counts, field):
        lens = counts[pos]
        off = np.concatenate(([0], np.cumsum(lens)))
        # concatenated templates + start offset of each template
        cat = np.concatenate([getattr(tmpl[int(m)], field) for m in uniq])
        starts = np.concatenate(([0], np.cumsum(counts)))[:-1]
        idx = np.repeat(starts[pos] - off[:-1], lens) + np.arange(off[-1])
        return off.astype(np.int64), cat[idx], lens

    fix_off, fix_tp, fix_len = gather(n_fix, "fix_tp")
    _, fix_alpha, _ = gather(n_fix, "fix_pay")            # unit notional * unit coupon = accrual fraction
    flt_off, flt_tp, _ = gather(n_flt, "flt_tp")
    _, flt_ts, _ = gather(n_flt, "flt_ts")
    _, flt_te, _ = gather(n_flt, "flt_te")
    _, flt_alpha, _ = gather(n_flt, "flt_alpha")
    # payment = year_frac * notional * coupon, in the leg's evaluation order (swap_fixed_leg.py:190)
    fix_pay = fix_alpha * np.repeat(notional, fix_len) * np.repeat(coupon, fix_len)
    sign_fix = np.where(pay_fixed, -1.0, 1.0)
    return TradeBatch(fix_off, flt_off, fix_tp, fix_pay, flt_tp, flt_ts, flt_te, flt_alpha,
                      notional.astype(np.float64), np.zeros(n), sign_fix, -sign_fix)
