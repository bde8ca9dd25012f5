"""The property assertions of the reference's tests/test_error_handling.py restated against this package's
date / day-count / schedule / calendar / interpolation utilities - the host code either side of the hot path
(SURVEY.md section 8(a) row D).  Each test names the reference test it follows."""
import pytest

from adrates_amd.market.curves.interpolator import interpolate
from adrates_amd.utils import BusDayAdjustTypes, DayCountTypes, FrequencyTypes, InterpTypes
from adrates_amd.utils.calendar import Calendar, CalendarTypes, DateGenRuleTypes
from adrates_amd.utils.date import Date
from adrates_amd.utils.day_count import DayCount
from adrates_amd.utils.error import LibError
from adrates_amd.utils.schedule import Schedule


@pytest.mark.parametrize("d, m, y", [(32, 1, 2023), (15, 13, 2023), (29, 2, 2023), (0, 1, 2023), (-1, 1, 2023)])
def test_invalid_dates_raise(d, m, y):
    """TestDateValidation: day 32, month 13, 29 Feb of a non-leap year, day 0, negative day (:28-59)."""
    with pytest.raises((ValueError, LibError, IndexError)):
        Date(d, m, y)


def test_leap_day_and_comparisons():
    """TestDateValidation (:45-79)."""
    dt = Date(29, 2, 2024)
    assert dt.d() == 29 and dt.m() == 2
    a, b, c = Date(15, 6, 2023), Date(16, 6, 2023), Date(15, 6, 2023)
    assert a < b and b > a and a != b and a == c


def test_day_count_edge_cases():
    """TestDayCountEdgeCases + TestNumericalStability (:83-126, 291-300)."""
    dc = DayCount(DayCountTypes.ACT_365F)
    dt = Date(15, 6, 2023)
    assert dc.year_frac(dt, dt)[0] == 0.0
    fwd, back = dc.year_frac(dt, Date(15, 12, 2023))[0], dc.year_frac(Date(15, 12, 2023), dt)[0]
    assert fwd > 0 and back < 0 and abs(fwd + back) < 1e-12
    yf, days, _ = dc.year_frac(Date(1, 1, 2000), Date(1, 1, 2100))
    assert 99.5 < yf < 100.5 and days > 36500
    assert dc.year_frac(Date(28, 2, 2024), Date(1, 3, 2024))[1] == 2
    yf, days, _ = dc.year_frac(dt, Date(16, 6, 2023))
    assert days == 1 and abs(yf - 1 / 365) < 1e-12


def test_schedule_edge_cases():
    """TestScheduleEdgeCases (:130-190)."""
    mk = lambda eff, term, freq: Schedule(effective_dt=eff, termination_dt=term, freq_type=freq,
                                          dg_type=DateGenRuleTypes.BACKWARD)
    assert len(mk(Date(15, 6, 2023), Date(15, 12, 2023), FrequencyTypes.SEMI_ANNUAL).schedule_dts()) >= 2
    assert len(mk(Date(15, 6, 2023), Date(15, 7, 2023), FrequencyTypes.MONTHLY).schedule_dts()) >= 2
    assert 50 <= len(mk(Date(15, 6, 2023), Date(15, 6, 2073), FrequencyTypes.ANNUAL).schedule_dts()) <= 52
    with pytest.raises(LibError):
        mk(Date(15, 6, 2023), Date(15, 6, 2022), FrequencyTypes.ANNUAL)


def test_weekend_calendar_and_adjustments():
    """TestCalendarEdgeCases (:244-287)."""
    cal = Calendar(CalendarTypes.WEEKEND)
    saturday, sunday, monday = Date(17, 6, 2023), Date(18, 6, 2023), Date(19, 6, 2023)
    assert not cal.is_business_day(saturday) and not cal.is_business_day(sunday) and cal.is_business_day(monday)
    following = cal.adjust(saturday, BusDayAdjustTypes.FOLLOWING)
    preceding = cal.adjust(saturday, BusDayAdjustTypes.PRECEDING)
    assert following.d() == 19 and cal.is_business_day(following)
    assert preceding.d() == 16 and cal.is_business_day(preceding)


def test_date_arithmetic_far_out():
    """TestNumericalStability.test_date_arithmetic_overflow_protection (:302-311)."""
    future = Date(15, 6, 2023).add_years(100)
    assert (future.y(), future.m(), future.d()) == (2123, 6, 15)


def test_interpolation_edge_cases():
    """TestInterpolatorEdgeCases / TestNumericalStability for the schemes on the path (:208-238, 313-324):
    monotone discount factors between monotone knots, extrapolation beyond the last knot stays in (0, 1) and below
    the last knot's value under LINEAR_ZERO_RATES, very close knots."""
    times, dfs = [1.0, 2.0, 5.0, 10.0], [0.98, 0.95, 0.88, 0.75]
    for method in (InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES):
        vals = [interpolate(t, times, dfs, method.value) for t in (1.0, 1.5, 2.0, 3.5, 5.0, 7.5, 10.0)]
        assert all(a >= b for a, b in zip(vals, vals[1:]))
    beyond = interpolate(15.0, times, dfs, InterpTypes.LINEAR_ZERO_RATES.value)
    assert 0.0 < beyond < 1.0 and beyond < dfs[-1]
    close = interpolate(1.0015, [1.0, 1.001, 1.002, 2.0], [0.98, 0.979, 0.978, 0.95], InterpTypes.LINEAR_ZERO_RATES.value)
    assert 0.977 < close < 0.98


def test_type_tolerance():
    """TestTypeValidation (:328-366): a float day is converted or refused, never silently mangled."""
    try:
        assert Date(15.5, 6, 2023).d() in (15, 16)
    except (TypeError, ValueError, LibError):
        pass
    assert DayCount(DayCountTypes.ACT_365F).year_frac(Date(15, 6, 2023), Date(15, 12, 2023))[0] > 0
