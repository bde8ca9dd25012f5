"""Deterministic synthetic OIS portfolios for benchmarks and large-size tests.

Specification: SURVEY.md section 8(d) - spot-starting swaps, maturity in whole
months U{1..360} ("offgrid": annual coupons with a front stub, so most cash-flow
times fall between curve knots) or whole years U{1..30} ("ongrid": every time
snaps to a knot), coupon U(0.01, 0.07), notional round(U(1e6, 5e7), -5),
pay/receive 50/50, float spread 0, both legs ANNUAL on the curve's day count,
MODIFIED_FOLLOWING on the WEEKEND calendar, ``numpy.random.default_rng(seed)``.

Every distinct maturity has one schedule, so the terms go through the vectorised
compiler (trades/compiler.py::compile_ois_terms): one template swap per distinct
schedule, per-trade arrays gathered with NumPy - bit-identical to compiling `OIS`
objects one by one (tests/test_synthetic.py).
"""
from __future__ import annotations

import numpy as np

from ..utils.calendar import BusDayAdjustTypes
from ..utils.currency import CurrencyTypes
from ..utils.day_count import DayCountTypes
from ..utils.frequency import FrequencyTypes
from ..utils.global_types import CurveTypes, SwapTypes
from .compiler import OISTerms, TradeBatch, compile_ois, compile_ois_terms
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
    """Batch of ``n`` synthetic trades as arrays, without creating n objects: the terms go through the
    vectorised compiler (`compile_ois_terms`), which builds one schedule per distinct maturity."""
    months, coupon, notional, pay_fixed = draw_terms(n, kind, seed)
    tenor_of = {int(m): f"{int(m)}M" for m in np.unique(months)}
    terms = OISTerms(effective_dt=value_dt, tenor=[tenor_of[int(m)] for m in months], coupon=coupon,
                     notional=notional, pay_fixed=pay_fixed, fixed_freq_type=freq, fixed_dc_type=dc_type,
                     floating_index=curve_type, currency=currency, float_freq_type=freq, float_dc_type=dc_type,
                     bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
    return compile_ois_terms(terms, value_dt)


def shard_of_portfolio(value_dt, n_total, rank, world_size, kind="offgrid", seed=DEFAULT_SEED, **kw):
    """Rank ``rank``'s contiguous share of ONE synthetic portfolio of ``n_total`` trades, cut where
    `adrates_amd.distributed.shard_batch` cuts it (near-equal cash-flow counts), without compiling the other
    ranks' trades: the cash-flow counts follow from the drawn maturities alone (annual legs with a front stub:
    ceil(months / 12) coupons per leg), so only the slice ``lo:hi`` of the terms is compiled.  Equal, bit for bit,
    to ``shard_batch(synthesize(value_dt, n_total, ...), rank, world_size)`` (tests/test_synthetic.py).
    Returns ``(batch, (lo, hi))``."""
    from ..distributed import canonical_chunks, shard_bounds
    freq = kw.get("freq", FrequencyTypes.ANNUAL)
    if freq != FrequencyTypes.ANNUAL:
        raise ValueError("shard_of_portfolio counts coupons for annual legs only")
    months, coupon, notional, pay_fixed = draw_terms(n_total, kind, seed)
    coupons = (months + 11) // 12
    cum = np.concatenate(([0], np.cumsum(coupons))).astype(np.int64)
    if kw.get("canonical_chunks"):
        # the rank's run of CANONICAL chunks (distributed.canonical_chunks): cut points that do not depend on the world
        # size; `chunks` (trade ranges relative to the returned batch) is handed back through kw["chunks_out"]
        ranges = canonical_chunks(cum, cum, rank, world_size)
        lo, hi = ranges[0][0], ranges[-1][1]
        kw["chunks_out"].extend((a - lo, b - lo) for a, b in ranges)
    else:
        lo, hi = shard_bounds(cum, cum, world_size)[rank]
    dc_type = kw.get("dc_type", DayCountTypes.ACT_365F)
    curve_type = kw.get("curve_type", CurveTypes.GBP_OIS_SONIA)
    currency = kw.get("currency", CurrencyTypes.GBP)
    tenor_of = {int(m): f"{int(m)}M" for m in np.unique(months[lo:hi])}
    terms = OISTerms(effective_dt=value_dt, tenor=[tenor_of[int(m)] for m in months[lo:hi]], coupon=coupon[lo:hi],
                     notional=notional[lo:hi], pay_fixed=pay_fixed[lo:hi], fixed_freq_type=freq, fixed_dc_type=dc_type,
                     floating_index=curve_type, currency=currency, float_freq_type=freq, float_dc_type=dc_type,
                     bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
    return compile_ois_terms(terms, value_dt), (lo, hi)
