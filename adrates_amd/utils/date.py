"""Excel-serial calendar dates for schedule generation.

Host-side input producer for the OIS valuation path: the kernels only ever see
year fractions, but those fractions must equal the reference's to the last bit,
so the date arithmetic follows cavour/utils/date.py exactly:

* a date is identified by its Excel serial number, *including* Lotus' phantom
  29-Feb-1900 (cavour/utils/date.py:137-181);
* ``weekday = (serial + 5) % 7`` with Monday = 0 (cavour/utils/date.py:212-216);
* month arithmetic clamps the day to the target month's length
  (cavour/utils/date.py:597-653) and tenor arithmetic composes it the same way
  (cavour/utils/date.py:796-879).

Unlike the reference, which walks a pre-computed 31-slot-per-month table, this
implementation converts through proleptic Gregorian ordinals, so there is no
global year window to resize.
"""
from __future__ import annotations

import datetime as _dt
import math
from collections.abc import Iterable
from enum import Enum

from .error import LibError

_MONTH_LEN = (31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31)
_SHORT_MONTH = ("JAN", "FEB", "MAR", "APR", "MAY", "JUN",
                "JUL", "AUG", "SEP", "OCT", "NOV", "DEC")
_SHORT_DAY = ("MON", "TUE", "WED", "THU", "FRI", "SAT", "SUN")

# Excel serial 1 is 1-Jan-1900; serial 60 is the non-existent 29-Feb-1900, so
# real dates from 1-Mar-1900 on are one further from the epoch than they
# "should" be.
_ORD_31DEC1899 = _dt.date(1899, 12, 31).toordinal()
_ORD_1MAR1900 = _dt.date(1900, 3, 1).toordinal()


class DateFormatTypes(Enum):
    BLOOMBERG = 1
    US_SHORT = 2
    US_MEDIUM = 3
    US_LONG = 4
    US_LONGEST = 5
    UK_SHORT = 6
    UK_MEDIUM = 7
    UK_LONG = 8
    UK_LONGEST = 9
    DATETIME = 10


g_date_type_format = DateFormatTypes.UK_LONG


def set_date_format(format_type):
    """Select the global ``repr`` format (cavour/utils/date.py:63-66)."""
    global g_date_type_format
    g_date_type_format = format_type


def is_leap_year(y: int) -> bool:
    return (y % 4 == 0 and y % 100 != 0) or y % 400 == 0


def days_in_month(m: int, y: int) -> int:
    if m < 1 or m > 12:
        raise LibError("Month must be 1-12")
    return 29 if (m == 2 and is_leap_year(y)) else _MONTH_LEN[m - 1]


def _serial_from_dmy(d: int, m: int, y: int) -> int:
    o = _dt.date(y, m, d).toordinal()
    return o - _ORD_31DEC1899 + (1 if o >= _ORD_1MAR1900 else 0)


def _dmy_from_serial(serial: int):
    if serial == 60:
        raise LibError("Excel serial 60 (29-Feb-1900) is not a real date")
    o = serial + _ORD_31DEC1899 - (1 if serial > 60 else 0)
    dd = _dt.date.fromordinal(o)
    return dd.day, dd.month, dd.year


def _elementwise(method):
    """Let a binary Date method accept an iterable on the right-hand side and
    return the same container type (cavour/utils/date.py:221-229)."""
    def wrapper(self, other):
        if isinstance(other, Iterable):
            return type(other)(method(self, o) for o in other)
        return method(self, other)
    wrapper.__name__ = method.__name__
    return wrapper


class Date:
    """Day-month-year date with Excel serial arithmetic."""

    MON, TUE, WED, THU, FRI, SAT, SUN = range(7)

    __slots__ = ("_d", "_m", "_y", "_hh", "_mm", "_ss", "_excel_dt", "_weekday")

    def __init__(self, d, m, y, hh=0, mm=0, ss=0):
        if 1900 <= d < 2100 and 0 < y <= 31:
            raise LibError("Date arguments must now be in the order Date(dd, mm, yyyy)")
        if y < 1900:
            raise LibError("Year cannot be before 1900")
        if m < 1 or m > 12:
            raise LibError("Date: month must be 1-12")
        if d < 1:
            raise LibError("Date: Leap year. Day not valid.")
        if d > days_in_month(m, y):
            raise LibError("Date: Leap year. Day not valid." if is_leap_year(y)
                           else "Date: Not Leap year. Day not valid.")
        if not 0 <= hh <= 23:
            raise LibError("Hours must be in range 0-23")
        if not 0 <= mm <= 59:
            raise LibError("Minutes must be in range 0-59")
        if not 0 <= ss <= 59:
            raise LibError("Seconds must be in range 0-59")

        self._d, self._m, self._y = int(d), int(m), int(y)
        self._hh, self._mm, self._ss = hh, mm, ss
        serial = _serial_from_dmy(self._d, self._m, self._y)
        self._weekday = (serial + 5) % 7
        # intraday part, accumulated in the reference's order (date.py:323-327)
        frac = hh / 24.0
        frac += mm / 24.0 / 60.0
        frac += ss / 24.0 / 60.0 / 60.0
        self._excel_dt = serial + frac  # always a float, as in the reference

    # -- accessors -----------------------------------------------------------
    def d(self):
        return self._d

    def m(self):
        return self._m

    def y(self):
        return self._y

    def excel_dt(self):
        return self._excel_dt

    def weekday(self):
        return self._weekday

    # -- constructors --------------------------------------------------------
    @classmethod
    def from_string(cls, date_string, format_string):
        t = _dt.datetime.strptime(date_string, format_string)
        return cls(t.day, t.month, t.year)

    @classmethod
    def from_date(cls, date):
        if isinstance(date, _dt.date):
            return cls(date.day, date.month, date.year)
        import numpy as np
        if isinstance(date, np.datetime64):
            t = date.astype("datetime64[D]").astype(_dt.date)
            return cls(t.day, t.month, t.year)
        raise LibError("from_date needs a datetime.date or numpy.datetime64")

    @classmethod
    def _from_serial(cls, serial: int):
        d, m, y = _dmy_from_serial(serial)
        return cls(d, m, y)

    # -- comparisons / differences -------------------------------------------
    @_elementwise
    def __gt__(self, other):
        return self._excel_dt > other._excel_dt

    @_elementwise
    def __lt__(self, other):
        return self._excel_dt < other._excel_dt

    @_elementwise
    def __ge__(self, other):
        return self._excel_dt >= other._excel_dt

    @_elementwise
    def __le__(self, other):
        return self._excel_dt <= other._excel_dt

    @_elementwise
    def __sub__(self, other):
        return self._excel_dt - other._excel_dt

    @_elementwise
    def __rsub__(self, other):
        return self._excel_dt - other._excel_dt

    @_elementwise
    def __eq__(self, other):
        return self._excel_dt == other._excel_dt

    def __hash__(self):
        return hash(self._excel_dt)

    # -- calendar predicates -------------------------------------------------
    def is_weekend(self):
        return self._weekday in (Date.SAT, Date.SUN)

    def is_eom(self):
        return self._d == days_in_month(self._m, self._y)

    def eom(self):
        return Date(days_in_month(self._m, self._y), self._m, self._y)

    # -- arithmetic ----------------------------------------------------------
    def add_days(self, num_days: int = 1):
        """Calendar-day shift, forwards or backwards (cavour/utils/date.py:507-525).
        The phantom serial 60 is stepped over exactly as the reference's table
        walk does (it only counts slots holding a positive day counter)."""
        n = int(num_days)
        if n != num_days:
            raise LibError("Number of days must be a whole number")
        serial = _serial_from_dmy(self._d, self._m, self._y) + n
        return Date._from_serial(serial)

    def add_hours(self, hours):
        if hours < 0:
            raise LibError("Number of hours must be positive")
        total = self._hh + hours
        moved = self.add_days(int(total / 24))
        return Date(moved._d, moved._m, moved._y, total % 24, self._mm, self._ss)

    def add_weekdays(self, num_days: int):
        """Shift by weekdays, skipping Saturdays/Sundays only
        (cavour/utils/date.py:529-593, "new logic" branch)."""
        if not isinstance(num_days, int):
            raise LibError("Num days must be an integer")
        step = 1 if num_days > 0 else -1
        left = abs(num_days)
        end = self
        while left > 0:
            end = end.add_days(step)
            if not end.is_weekend():
                left -= 1
        return end

    def _shift_months(self, months: int):
        idx = self._y * 12 + (self._m - 1) + months
        y, m0 = divmod(idx, 12)
        m = m0 + 1
        return Date(min(self._d, days_in_month(m, y)), m, y)

    def add_months(self, mm):
        """Shift by whole months, clamping the day to the target month's
        length; a list in gives a list out (cavour/utils/date.py:597-653)."""
        scalar = isinstance(mm, (int, float))
        out = []
        for v in ([mm] if scalar else mm):
            if int(v) != v:
                raise LibError("Must only pass integers or float integers.")
            out.append(self._shift_months(int(v)))
        return out[0] if scalar else out

    def add_years(self, yy):
        """Shift by (possibly fractional) years; the fractional part becomes
        days at 365.242/12 per month (cavour/utils/date.py:657-694)."""
        scalar = isinstance(yy, (int, float))
        out = []
        for v in ([yy] if scalar else yy):
            whole = int(v * 12.0)
            extra = int((v * 12.0 - whole) * (365.242 / 12.0))
            out.append(self.add_months(whole).add_days(extra))
        return out[0] if scalar else out

    def add_tenor(self, tenor):
        """Shift by a tenor string such as "1D", "2W", "18M", "10Y", "ON", "TN".
        No business-day adjustment is applied (cavour/utils/date.py:796-879).

        Month tenors step one month at a time (each step clamps the day) and
        finally restore the original day-of-month where the landing month
        allows; year tenors step twelve months at a time and do not restore it
        (29-Feb + "1Y" = 28-Feb, and stays on the 28th afterwards).
        """
        is_list = isinstance(tenor, list)
        if is_list:
            if not all(isinstance(t, str) for t in tenor):
                raise LibError("Tenor must be a string e.g. '5Y'")
            tenors = tenor
        elif isinstance(tenor, str):
            tenors = [tenor]
        else:
            raise LibError("Tenor must be a string e.g. '5Y'")

        out = []
        for raw in tenors:
            ts = raw.upper()
            if ts in ("ON", "TN"):
                unit, n = "D", 1
            else:
                unit = ts[-1]
                if unit not in "DWMY":
                    raise LibError("Unknown tenor type in " + raw)
                n = int(ts[:-1])
            sign = int(math.copysign(1, n)) if n != 0 else 1
            cur = Date(self._d, self._m, self._y)
            if unit == "D":
                cur = cur.add_days(n)
            elif unit == "W":
                cur = cur.add_days(7 * n)
            elif unit == "M":
                for _ in range(abs(n)):
                    cur = cur._shift_months(sign)
                cur = Date(min(self._d, days_in_month(cur._m, cur._y)), cur._m, cur._y)
            else:
                for _ in range(abs(n)):
                    cur = cur._shift_months(12 * sign)
            out.append(cur)
        return out if is_list else out[0]

    # -- IMM helpers (kept because they are cheap; not used by the OIS path) --
    def third_wednesday_of_month(self, m: int, y: int):
        for d in range(15, 22):
            if Date(d, m, y).weekday() == Date.WED:
                return d
        raise LibError("Third Wednesday not found")

    def next_imm_date(self):
        """Next quarterly IMM date strictly after this date
        (cavour/utils/date.py:759-792)."""
        y, m, d = self._y, self._m, self._d
        q_month = ((m - 1) // 3 + 1) * 3
        if m == q_month and d >= self.third_wednesday_of_month(m, y):
            q_month += 3
        if q_month > 12:
            q_month -= 12
            y += 1
        return Date(self.third_wednesday_of_month(q_month, y), q_month, y)

    # -- conversions / printing ----------------------------------------------
    def datetime(self):
        return _dt.date(self._y, self._m, self._d)

    def str(self):
        return f"{self._d:02d}{_SHORT_MONTH[self._m - 1]}{self._y}"

    def __repr__(self):
        dd = f"{self._d:02d}"
        mm = f"{self._m:02d}"
        mon = _SHORT_MONTH[self._m - 1]
        yyyy = str(self._y)
        yy = yyyy[2:]
        dow = _SHORT_DAY[self._weekday]
        f = g_date_type_format
        if f == DateFormatTypes.UK_LONGEST:
            return f"{dow} {dd} {mon} {yyyy}"
        if f == DateFormatTypes.UK_LONG:
            return f"{dd}-{mon}-{yyyy}"
        if f == DateFormatTypes.UK_MEDIUM:
            return f"{dd}/{mm}/{yyyy}"
        if f == DateFormatTypes.UK_SHORT:
            return f"{dd}/{mm}/{yy}"
        if f == DateFormatTypes.US_LONGEST:
            return f"{dow} {mon} {dd} {yyyy}"
        if f == DateFormatTypes.US_LONG:
            return f"{mon}-{dd}-{yyyy}"
        if f == DateFormatTypes.US_MEDIUM:
            return f"{mm}-{dd}-{yyyy}"
        if f == DateFormatTypes.US_SHORT:
            return f"{mm}-{dd}-{yy}"
        if f == DateFormatTypes.BLOOMBERG:
            return f"{mm}/{dd}/{yy}"
        if f == DateFormatTypes.DATETIME:
            return f"{dd}/{mm}/{yyyy} {self._hh:02d}:{self._mm:02d}:{self._ss:02d}"
        raise LibError("Unknown date format")

    def _print(self):
        print(self)


def datediff(d1: Date, d2: Date) -> int:
    """Whole days from d1 to d2 (cavour/utils/date.py:1042-1046)."""
    return int(d2.excel_dt() - d1.excel_dt())


def from_datetime(dt) -> Date:
    return Date(dt.day, dt.month, dt.year)


def daily_working_day_schedule(start_dt: Date, end_dt: Date):
    out = [start_dt]
    cur = start_dt
    while cur < end_dt:
        cur = cur.add_weekdays(1)
        out.append(cur)
    return out


def date_range(start_dt: Date, end_dt: Date, tenor: str = "1D"):
    """Dates from start (inclusive) to end (inclusive) in steps of ``tenor``
    (cavour/utils/date.py:1075-1093)."""
    if start_dt > end_dt:
        return []
    out = []
    cur = start_dt
    while cur < end_dt:
        out.append(cur)
        cur = cur.add_tenor(tenor)
    out.append(end_dt)
    return out
