"""Array schedules (`utils.schedule_np`) against the object path they replace, date for date and bit for bit."""
import numpy as np
import pytest

from adrates_amd.market.position import xccy_engine as XE
from adrates_amd.utils import BusDayAdjustTypes, CalendarTypes, Date, DayCountTypes, FrequencyTypes
from adrates_amd.utils import schedule_np as S
from adrates_amd.utils.calendar import Calendar
from adrates_amd.utils.global_types import CurveTypes
from adrates_amd.utils.currency import CurrencyTypes
from adrates_amd.utils.schedule import Schedule

VALUE_DT = Date(30, 4, 2024)


def _serial(d):
    return int(d.excel_dt())


def test_calendar_arithmetic_matches_the_objects():
    rng = np.random.default_rng(1)
    serials = rng.integers(_serial(Date(1, 1, 1990)), _serial(Date(31, 12, 2080)), 3000)
    dates = [Date._from_serial(int(s)) for s in serials]
    y, m, d = S.ymd_from_serial(serials)
    assert [(x.y(), x.m(), x.d()) for x in dates] == list(zip(y.tolist(), m.tolist(), d.tolist()))
    cal = Calendar(CalendarTypes.WEEKEND)
    for bd in BusDayAdjustTypes:
        want = [_serial(cal.adjust(x, bd)) for x in dates]
        assert S.adjust(serials, bd).tolist() == want, bd
    lag = rng.integers(-3, 6, serials.size)
    want = [_serial(cal.add_business_days(x, int(k))) for x, k in zip(dates, lag)]
    assert S.add_business_days(serials, lag).tolist() == want
    # tenors, month ends and leap days included
    specials = [Date(29, 2, 2024), Date(31, 1, 2023), Date(30, 11, 2025), Date(31, 8, 2024), Date(28, 2, 2023)]
    table = ["1D", "3W", "1M", "18M", "11M", "1Y", "4Y", "30Y", "ON", "-2M", "-1Y", "0M"]
    count, unit = S.parse_tenors(table)
    for x in dates[:200] + specials:
        got = S.add_tenor(np.full(len(table), _serial(x)), count, unit)
        assert got.tolist() == [_serial(x.add_tenor(t)) for t in table], x


@pytest.mark.parametrize("bd", [BusDayAdjustTypes.FOLLOWING, BusDayAdjustTypes.MODIFIED_FOLLOWING, BusDayAdjustTypes.PRECEDING])
def test_backward_schedules_match_the_objects(bd):
    rng = np.random.default_rng(2)
    n = 1500
    eff = rng.integers(_serial(Date(1, 1, 2020)), _serial(Date(31, 12, 2026)), n)
    eff[:5] = [_serial(Date(29, 2, 2024)), _serial(Date(31, 1, 2024)), _serial(Date(31, 8, 2024)), _serial(Date(30, 4, 2024)), _serial(Date(31, 12, 2023))]
    months = rng.integers(1, 400, n)
    term = S.add_tenor(eff, months, np.full(n, 2))
    mpp = np.array([12, 6, 3, 1])[rng.integers(0, 4, n)]
    freq_of = {12: FrequencyTypes.ANNUAL, 6: FrequencyTypes.SEMI_ANNUAL, 3: FrequencyTypes.QUARTERLY, 1: FrequencyTypes.MONTHLY}
    off, dates, plain = S.backward_schedules(eff, term, mpp, bd)
    n_plain = 0
    for i in range(n):
        obj = Schedule(Date._from_serial(int(eff[i])), Date._from_serial(int(term[i])), freq_of[int(mpp[i])],
                       CalendarTypes.WEEKEND, bd)._adjusted_dts
        if plain[i]:
            n_plain += 1
            assert dates[off[i]:off[i + 1]].tolist() == [_serial(x) for x in obj], i
        else:                                  # the front-dropping quirk (schedule.py:256-266): left to the object path
            assert len(obj) < off[i + 1] - off[i]
    assert n_plain > 0.95 * n


def _terms(n, seed, foreign_dc=DayCountTypes.ACT_360):
    rng = np.random.default_rng(seed)
    back = rng.integers(0, 30, n)
    eff = np.array([_serial(VALUE_DT.add_months(-int(b)).add_days(int(k))) for b, k in zip(back, rng.integers(0, 28, n))])
    table = [f"{m}M" for m in range(1, 372)] + ["2Y", "10Y", "30Y"]
    codes = rng.integers(36, len(table), n)
    freqs = [FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL, FrequencyTypes.QUARTERLY]
    return XE.XccyTerms(effective_dt=eff, tenor=(codes, table), domestic_notional=rng.uniform(1e6, 5e7, n),
                        foreign_notional=rng.uniform(1e6, 5e7, n), domestic_spread=rng.uniform(0, 5e-4, n),
                        foreign_spread=rng.uniform(1e-3, 6e-3, n), domestic_freq_type=FrequencyTypes.ANNUAL,
                        foreign_freq_type=[freqs[i] for i in rng.integers(0, 3, n)],
                        domestic_dc_type=DayCountTypes.ACT_365F, foreign_dc_type=foreign_dc,
                        domestic_floating_index=CurveTypes.GBP_OIS_SONIA, foreign_floating_index=CurveTypes.USD_OIS_SOFR,
                        domestic_currency=CurrencyTypes.GBP, foreign_currency=CurrencyTypes.USD,
                        domestic_payment_lag=rng.integers(0, 3, n), foreign_payment_lag=rng.integers(0, 4, n))


def _same(a: XE.RawXccy, b: XE.RawXccy):
    import dataclasses
    for f in dataclasses.fields(XE.RawXccy):
        assert np.array_equal(getattr(a, f.name), getattr(b, f.name)), f.name


def test_raw_from_terms_is_the_template_route_bit_for_bit(monkeypatch):
    """The array route against one `XccyBasisSwap` object per distinct schedule (the route it replaces), all fields
    bitwise equal; swaps whose schedule hits the de-duplication quirk take the template route inside the same call."""
    terms = _terms(4000, 5)
    fast = XE.raw_from_terms(terms, VALUE_DT, DayCountTypes.ACT_365F)
    monkeypatch.setattr(XE, "_FIXED_DENOMINATOR", {})                   # forces the template route
    slow = XE.raw_from_terms(terms, VALUE_DT, DayCountTypes.ACT_365F)
    _same(fast, slow)


def test_raw_from_terms_mixes_routes_in_book_order(monkeypatch):
    """A day count without a fixed denominator on some swaps: those go by templates, the rest by arrays, and the
    book comes back in the caller's order."""
    n = 600
    terms = _terms(n, 6)
    rng = np.random.default_rng(7)
    terms.foreign_dc_type = [DayCountTypes.ACT_ACT_ISDA if x else DayCountTypes.ACT_360 for x in rng.random(n) < 0.3]
    mixed = XE.raw_from_terms(terms, VALUE_DT, DayCountTypes.ACT_365F)
    monkeypatch.setattr(XE, "_FIXED_DENOMINATOR", {})
    _same(mixed, XE.raw_from_terms(terms, VALUE_DT, DayCountTypes.ACT_365F))


def test_errors_match_the_object_path():
    """The array route raises what the objects raise (same `LibError` texts) and refuses what it cannot represent."""
    from adrates_amd.utils import LibError
    eff = np.array([_serial(Date(30, 4, 2024))])
    with pytest.raises(LibError, match="Effective date must be before termination date"):
        S.backward_schedules(eff, eff, np.array([12]), BusDayAdjustTypes.FOLLOWING)
    with pytest.raises(LibError, match="Unknown tenor type"):
        S.parse_tenors(["5Q"])
    with pytest.raises(LibError, match="Tenor must be a string"):
        S.parse_tenors([5])
    with pytest.raises(LibError, match="not supported"):
        S.ymd_from_serial(np.array([10]))                          # before 1-Mar-1900: the Lotus leap-day region
    with pytest.raises(LibError, match="not supported"):
        S.add_tenor(eff, np.array([400]), np.array([3]))           # 400 years on: beyond the month table
    # a zero-length tenor: the swap constructor's own check fires first on the object path, the schedule's here
    terms = _terms(3, 1)
    terms.tenor = (np.zeros(3, dtype=np.int64), ["0M"])
    with pytest.raises(LibError, match="Effective date must be before termination date"):
        XE.raw_from_terms(terms, VALUE_DT, DayCountTypes.ACT_365F)
    # coded columns are validated
    terms = _terms(3, 1)
    terms.foreign_freq_type = (np.array([0, 1, 2]), [FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL])
    with pytest.raises(LibError, match="coded column"):
        XE.raw_from_terms(terms, VALUE_DT, DayCountTypes.ACT_365F)
