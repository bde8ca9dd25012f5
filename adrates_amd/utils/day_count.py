"""Day-count conventions: year fraction between two dates.

Mirrors `DayCount.year_frac` / `days_in_year` of cavour/utils/day_count.py
(:122-331, :334-370).  Every convention returns ``(fraction, numerator,
denominator)``; the OIS path uses ACT/365F, ACT/360 and (as the OIS float-leg
default, cavour/trades/rates/ois.py:113) 30E/360.
"""
from enum import Enum

from .date import Date, datediff, is_leap_year
from .error import LibError
from .frequency import FrequencyTypes, annual_frequency
from .global_vars import gDaysInYear


class DayCountTypes(Enum):
    ZERO = 0
    THIRTY_360_BOND = 1
    THIRTY_E_360 = 2
    THIRTY_E_360_ISDA = 3
    THIRTY_E_PLUS_360 = 4
    ACT_ACT_ISDA = 5
    ACT_ACT_ICMA = 6
    ACT_365F = 7
    ACT_360 = 8
    ACT_365L = 9
    SIMPLE = 10


def is_last_day_of_feb(dt: Date):
    """True on the last day of February; ``None`` (falsy) for other February
    days, exactly like cavour/utils/day_count.py:64-74."""
    if dt.m() == 2:
        if dt.d() == (29 if is_leap_year(dt.y()) else 28):
            return True
        return None
    return False


_THIRTY_360 = (DayCountTypes.THIRTY_360_BOND, DayCountTypes.THIRTY_E_360,
               DayCountTypes.THIRTY_E_360_ISDA, DayCountTypes.THIRTY_E_PLUS_360)


class DayCount:
    def __init__(self, dccType: DayCountTypes):
        if dccType not in DayCountTypes:
            raise LibError("Need to pass FinDayCountType")
        self._type = dccType

    def year_frac(self, dt1: Date, dt2: Date, dt3: Date = None,
                  freq_type: FrequencyTypes = FrequencyTypes.ANNUAL,
                  isTerminationDate: bool = False):
        """Year fraction from ``dt1`` to ``dt2``; ``dt3``/``freq_type`` are only
        needed by the bond-style conventions."""
        t = self._type

        if t in _THIRTY_360:
            d1, m1, y1 = dt1.d(), dt1.m(), dt1.y()
            d2, m2, y2 = dt2.d(), dt2.m(), dt2.y()
            if d1 == 31:
                d1 = 30
            if t == DayCountTypes.THIRTY_360_BOND:
                if d2 == 31 and d1 == 30:
                    d2 = 30
            elif t == DayCountTypes.THIRTY_E_360:
                if d2 == 31:
                    d2 = 30
            elif t == DayCountTypes.THIRTY_E_360_ISDA:
                if is_last_day_of_feb(dt1) is True:
                    d1 = 30
                if d2 == 31:
                    d2 = 30
                if is_last_day_of_feb(dt2) is True and isTerminationDate is False:
                    d2 = 30
            else:  # THIRTY_E_PLUS_360: a 31st rolls to the 1st of the next month
                if d2 == 31:
                    m2 += 1
                    d2 = 1
            num = 360 * (y2 - y1) + 30 * (m2 - m1) + (d2 - d1)
            return num / 360, num, 360

        if t in (DayCountTypes.ACT_ACT_ISDA, DayCountTypes.ZERO):
            y1, y2 = dt1.y(), dt2.y()
            den1 = 366 if is_leap_year(y1) else 365
            den2 = 366 if is_leap_year(y2) else 365
            if y1 == y2:
                num = dt2 - dt1
                return num / den1, num, den1
            days1 = datediff(dt1, Date(1, 1, y1 + 1))
            days2 = datediff(Date(1, 1, y2), dt2)
            frac = days1 / den1 + days2 / den2 + (y2 - y1 - 1.0)
            return frac, days1 + days2, den1 + den2

        if t == DayCountTypes.ACT_ACT_ICMA:
            freq = annual_frequency(freq_type)
            if dt3 is None or freq is None:
                raise LibError("ACT_ACT_ICMA requires three dates and a freq")
            num = dt2 - dt1
            den = freq * (dt3 - dt1)
            return num / den, num, den

        if t == DayCountTypes.ACT_365F:
            num = dt2 - dt1
            return num / 365, num, 365

        if t == DayCountTypes.ACT_360:
            num = dt2 - dt1
            return num / 360, num, 360

        if t == DayCountTypes.ACT_365L:
            freq = annual_frequency(freq_type)
            y1 = dt1.y()
            y3 = dt2.y() if dt3 is None else dt3.y()
            num = dt2 - dt1
            den = 365
            if is_leap_year(y1):
                feb29 = Date(29, 2, y1)
            elif is_leap_year(y3):
                feb29 = Date(29, 2, y3)
            else:
                feb29 = Date(1, 1, 1900)
            if freq == 1:
                if feb29 > dt1 and feb29 <= dt3:
                    den = 366
            elif is_leap_year(y3):
                den = 366
            return num / den, num, den

        if t == DayCountTypes.SIMPLE:
            num = dt2 - dt1
            return num / gDaysInYear, num, gDaysInYear

        raise LibError(str(t) + " is not one of DayCountTypes")

    def days_in_year(self):
        """Fixed denominator of the convention, where it has one
        (cavour/utils/day_count.py:334-370)."""
        t = self._type
        if t in _THIRTY_360 or t == DayCountTypes.ACT_360:
            return 360
        if t is DayCountTypes.ACT_365F:
            return 365
        if t is DayCountTypes.SIMPLE:
            return gDaysInYear
        if t in (DayCountTypes.ACT_ACT_ISDA, DayCountTypes.ZERO):
            raise LibError("ACT/ACT (ISDA or ZERO) requires the actual dates to compute days in year")
        if t is DayCountTypes.ACT_365L:
            raise LibError("ACT/365L depends on whether the period spans a leap day")
        if t is DayCountTypes.ACT_ACT_ICMA:
            raise LibError("ACT/ACT ICMA needs the full coupon-period dates and frequency")
        raise LibError(f"No fixed days-in-year defined for convention {t}")

    def __repr__(self):
        return str(self._type)
