"""Vectorised trade compiler (trades/compiler.py::compile_ois_terms) == the object path, bit for bit."""
import time

import numpy as np
import pytest

from adrates_amd.trades.compiler import OISTerms, compile_ois, compile_ois_terms
from adrates_amd.trades.rates.ois import OIS
from adrates_amd.utils import (BusDayAdjustTypes, CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes,
                               SwapTypes)
from adrates_amd.utils.error import LibError

from . import _fixtures as F

FIELDS = ("fix_off", "flt_off", "fix_tp", "fix_pay", "flt_tp", "flt_ts", "flt_te", "flt_alpha", "notional",
          "spread", "fix_sign", "flt_sign")


def _same(a, b):
    for f in FIELDS:
        assert np.array_equal(getattr(a, f), getattr(b, f)), f


def test_mixed_terms_match_objects():
    vd = F.README_VALUE_DT
    rng = np.random.default_rng(5)
    n = 300
    effs = [vd, vd.add_weekdays(2), vd.add_months(3)]
    eff = [effs[i] for i in rng.integers(0, 3, n)]
    tenors = [["1W", "3M", "18M", "2Y", "87M", "10Y", "30Y"][i] for i in rng.integers(0, 7, n)]
    ffreq = [[FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL][i] for i in rng.integers(0, 2, n)]
    lfreq = [[FrequencyTypes.ANNUAL, FrequencyTypes.QUARTERLY][i] for i in rng.integers(0, 2, n)]
    fdc = [[DayCountTypes.ACT_365F, DayCountTypes.ACT_360][i] for i in rng.integers(0, 2, n)]
    lag = rng.integers(0, 3, n)
    coupon = rng.uniform(0.01, 0.07, n)
    notional = np.round(rng.uniform(1e6, 5e7, n), -5)
    pay = rng.random(n) < 0.5
    spread = np.where(rng.random(n) < 0.3, 0.0025, 0.0)
    terms = OISTerms(effective_dt=eff, tenor=tenors, coupon=coupon, notional=notional, pay_fixed=pay,
                     fixed_freq_type=ffreq, fixed_dc_type=fdc, floating_index=CurveTypes.GBP_OIS_SONIA,
                     currency=CurrencyTypes.GBP, float_freq_type=lfreq, float_dc_type=DayCountTypes.ACT_365F,
                     float_spread=spread, payment_lag=lag, bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
    swaps = [OIS(effective_dt=eff[i], term_dt_or_tenor=tenors[i],
                 fixed_leg_type=SwapTypes.PAY if pay[i] else SwapTypes.RECEIVE, fixed_coupon=float(coupon[i]),
                 fixed_freq_type=ffreq[i], fixed_dc_type=fdc[i], floating_index=CurveTypes.GBP_OIS_SONIA,
                 currency=CurrencyTypes.GBP, notional=float(notional[i]), payment_lag=int(lag[i]),
                 float_spread=float(spread[i]), float_freq_type=lfreq[i], float_dc_type=DayCountTypes.ACT_365F,
                 bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING) for i in range(n)]
    _same(compile_ois_terms(terms, vd), compile_ois(swaps, vd))


def test_scalars_broadcast_serial_dates_and_defaults():
    vd = F.README_VALUE_DT
    terms = OISTerms(effective_dt=int(vd.excel_dt()), tenor=["5Y", "5Y", "2Y"], coupon=0.04, notional=[1e6, 2e6, 3e6],
                     pay_fixed=True, fixed_freq_type=FrequencyTypes.ANNUAL, fixed_dc_type=DayCountTypes.ACT_360,
                     floating_index=CurveTypes.USD_OIS_SOFR, currency=CurrencyTypes.USD)
    swaps = [OIS(vd, t, SwapTypes.PAY, 0.04, FrequencyTypes.ANNUAL, DayCountTypes.ACT_360, CurveTypes.USD_OIS_SOFR,
                 CurrencyTypes.USD, notional=nn, float_freq_type=FrequencyTypes.ANNUAL,
                 float_dc_type=DayCountTypes.ACT_360) for t, nn in (("5Y", 1e6), ("5Y", 2e6), ("2Y", 3e6))]
    _same(compile_ois_terms(terms, vd), compile_ois(swaps, vd))
    with pytest.raises(LibError):
        compile_ois_terms(OISTerms(vd, ["5Y"], 0.04, [1e6, 2e6], True, FrequencyTypes.ANNUAL, DayCountTypes.ACT_360,
                                   CurveTypes.USD_OIS_SOFR, CurrencyTypes.USD), vd)


def test_large_batch_is_fast():
    """1e5 trades over 360 distinct schedules: seconds, where the object path needs minutes."""
    vd = F.README_VALUE_DT
    rng = np.random.default_rng(1)
    n = 100_000
    months = rng.integers(1, 361, n)
    names = {m: f"{m}M" for m in range(1, 361)}
    t0 = time.perf_counter()
    b = compile_ois_terms(OISTerms(vd, [names[int(m)] for m in months], rng.uniform(0.01, 0.07, n),
                                   np.full(n, 1e6), rng.random(n) < 0.5, FrequencyTypes.ANNUAL,
                                   DayCountTypes.ACT_365F, CurveTypes.GBP_OIS_SONIA, CurrencyTypes.GBP,
                                   float_freq_type=FrequencyTypes.ANNUAL, float_dc_type=DayCountTypes.ACT_365F,
                                   bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING), vd)
    assert b.n_trades == n and time.perf_counter() - t0 < 30.0


def test_array_route_is_the_template_route_bit_for_bit(monkeypatch):
    """Schedules on arrays (`utils.schedule_np`) against one `OIS` object per distinct schedule, the route they replace:
    distinct effective dates (seasoned, forward starting, month ends, a leap day), tenors in months and years, four
    frequencies, payment lags, two business-day rules, both calendars the arrays know, serial dates; a day count
    without a fixed denominator and FORWARD date generation on some trades, which take the template route inside the
    same call; all arrays bitwise equal and in the caller's order."""
    from adrates_amd.trades import compiler as C
    from adrates_amd.utils import CalendarTypes, DateGenRuleTypes
    from adrates_amd.utils.date import Date
    vd = F.README_VALUE_DT
    rng = np.random.default_rng(11)
    n = 3000
    days = rng.integers(-400, 200, n)
    eff = np.array([int(vd.excel_dt()) + int(d) for d in days], dtype=np.int64)
    eff[:3] = [int(Date(29, 2, 2024).excel_dt()), int(Date(31, 1, 2024).excel_dt()), int(Date(31, 8, 2023).excel_dt())]
    tenor_table = [f"{m}M" for m in range(18, 361, 7)] + ["2Y", "5Y", "10Y", "30Y"]
    freqs = [FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL, FrequencyTypes.QUARTERLY, FrequencyTypes.MONTHLY]
    pick = lambda table, p=None: [table[i] for i in rng.choice(len(table), size=n, p=p)]
    terms = OISTerms(effective_dt=eff, tenor=(rng.integers(0, len(tenor_table), n), tenor_table),
                     coupon=rng.uniform(0.01, 0.07, n), notional=np.round(rng.uniform(1e6, 5e7, n), -5),
                     pay_fixed=rng.random(n) < 0.5, fixed_freq_type=pick(freqs[:2]),
                     fixed_dc_type=pick([DayCountTypes.ACT_365F, DayCountTypes.ACT_360]),
                     floating_index=CurveTypes.GBP_OIS_SONIA, currency=CurrencyTypes.GBP, float_freq_type=pick(freqs),
                     float_dc_type=pick([DayCountTypes.ACT_365F, DayCountTypes.ACT_360, DayCountTypes.THIRTY_E_360], [0.6, 0.3, 0.1]),
                     float_spread=np.where(rng.random(n) < 0.3, 0.0025, 0.0), payment_lag=rng.integers(0, 4, n),
                     bd_type=pick([BusDayAdjustTypes.FOLLOWING, BusDayAdjustTypes.MODIFIED_FOLLOWING]),
                     cal_type=pick([CalendarTypes.WEEKEND, CalendarTypes.NONE], [0.8, 0.2]),
                     dg_type=pick([DateGenRuleTypes.BACKWARD, DateGenRuleTypes.FORWARD], [0.95, 0.05]))
    fast = compile_ois_terms(terms, vd)
    calls = []
    real = C._legs_by_arrays
    monkeypatch.setattr(C, "_legs_by_arrays", lambda *a, **k: calls.append(1) or real(*a, **k))
    compile_ois_terms(terms, vd)
    assert len(calls) == 4                                                 # two rules x two calendars
    monkeypatch.setattr(C, "_FIXED_DENOMINATOR", {})                       # forces the template route for every trade
    _same(fast, compile_ois_terms(terms, vd))


def test_unique_rows_is_numpy_unique_axis0():
    """The mixed-radix key of `compiler.unique_rows` orders rows like ``np.unique(axis=0)`` (same rows, same inverse),
    negative entries and single rows included; columns too wide for one int64 fall back to NumPy's."""
    from adrates_amd.trades.compiler import unique_rows
    rng = np.random.default_rng(0)
    for cols in ([rng.integers(40000, 46000, 5000), rng.integers(-2, 3, 5000), rng.integers(0, 5, 5000)],
                 [np.array([7]), np.array([-1])],
                 [rng.integers(-2 ** 40, 2 ** 40, 300), rng.integers(-2 ** 40, 2 ** 40, 300)]):
        rows, inverse = unique_rows(cols)
        want_rows, want_inverse = np.unique(np.stack(cols, axis=1), axis=0, return_inverse=True)
        assert np.array_equal(rows, want_rows) and np.array_equal(inverse, want_inverse.reshape(-1))
    rows, inverse = unique_rows([np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)])
    assert rows.shape == (0, 2) and inverse.shape == (0,)
