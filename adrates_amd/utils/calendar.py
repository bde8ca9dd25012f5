"""Business-day calendars and adjustment rules used when rolling schedule dates.

Mirrors the subset of cavour/utils/calendar.py the OIS path reaches
(`Calendar.adjust` :139-217, `is_business_day` :257-274, `add_business_days`
:221-253) for the NONE and WEEKEND calendars.  The national holiday calendars
of the reference are outside the hot-path scope;
asking for one raises ``LibError`` instead of silently treating it as WEEKEND.
"""
from enum import Enum

from .date import Date
from .error import LibError


class BusDayAdjustTypes(Enum):
    NONE = 1
    FOLLOWING = 2
    MODIFIED_FOLLOWING = 3
    PRECEDING = 4
    MODIFIED_PRECEDING = 5


class CalendarTypes(Enum):
    NONE = 1
    WEEKEND = 2
    AUSTRALIA = 3
    CANADA = 4
    FRANCE = 5
    GERMANY = 6
    ITALY = 7
    JAPAN = 8
    NEW_ZEALAND = 9
    NORWAY = 10
    SWEDEN = 11
    SWITZERLAND = 12
    TARGET = 13
    UNITED_STATES = 14
    UNITED_KINGDOM = 15
    INTERSECTION = 16


class DateGenRuleTypes(Enum):
    FORWARD = 1
    BACKWARD = 2


_SUPPORTED = (CalendarTypes.NONE, CalendarTypes.WEEKEND, CalendarTypes.INTERSECTION)


class Calendar:
    """Decides which dates are business days and rolls dates that are not."""

    def __init__(self, cal_type: CalendarTypes, constituent_calendars=None):
        if cal_type not in CalendarTypes:
            raise LibError("Need to pass FinCalendarType and not " + str(cal_type))
        if cal_type not in _SUPPORTED:
            raise LibError(f"Calendar {cal_type.name} is outside the OIS hot-path scope "
                           "(only NONE, WEEKEND and INTERSECTION of those are built)")
        self._cal_type = cal_type
        self._constituent_calendars = constituent_calendars or []

    # ------------------------------------------------------------------ rolls
    def _roll(self, dt: Date, step: int) -> Date:
        while not self.is_business_day(dt):
            dt = dt.add_days(step)
        return dt

    def adjust(self, dt: Date, bd_type: BusDayAdjustTypes) -> Date:
        """Roll ``dt`` to a business day under the given convention
        (cavour/utils/calendar.py:139-217).  The MODIFIED variants roll the
        other way from the *original* date when the first roll leaves the month."""
        if type(bd_type) != BusDayAdjustTypes:
            raise LibError("Invalid type passed. Need Finbd_type")
        if self._cal_type == CalendarTypes.NONE or bd_type == BusDayAdjustTypes.NONE:
            return dt
        if bd_type == BusDayAdjustTypes.FOLLOWING:
            return self._roll(dt, +1)
        if bd_type == BusDayAdjustTypes.PRECEDING:
            return self._roll(dt, -1)
        if bd_type in (BusDayAdjustTypes.MODIFIED_FOLLOWING,
                       BusDayAdjustTypes.MODIFIED_PRECEDING):
            step = +1 if bd_type == BusDayAdjustTypes.MODIFIED_FOLLOWING else -1
            rolled = self._roll(dt, step)
            if rolled.m() != dt.m():
                rolled = self._roll(Date(dt.d(), dt.m(), dt.y()), -step)
            return rolled
        raise LibError("Unknown adjustment convention" + str(bd_type))

    def add_business_days(self, start_dt: Date, num_days: int) -> Date:
        """Move ``num_days`` business days forwards (or backwards if negative)
        (cavour/utils/calendar.py:221-253)."""
        if not isinstance(num_days, int):
            raise LibError("Num days must be an integer")
        step = 1 if num_days >= 0 else -1
        left = abs(num_days)
        cur = Date(start_dt.d(), start_dt.m(), start_dt.y())
        while left > 0:
            cur = cur.add_days(step)
            if self.is_business_day(cur):
                left -= 1
        return cur

    # ------------------------------------------------------------- predicates
    def is_business_day(self, dt: Date) -> bool:
        if self._cal_type == CalendarTypes.INTERSECTION:
            return all(c.is_business_day(dt) for c in self._constituent_calendars)
        # Saturdays and Sundays are never business days - this holds for the
        # NONE calendar too in the reference (calendar.py:266-269), although
        # ``adjust`` returns early for NONE and never asks.
        if dt.is_weekend():
            return False
        return not self.is_holiday(dt)

    def is_holiday(self, dt: Date) -> bool:
        if self._cal_type == CalendarTypes.INTERSECTION:
            return any(c.is_holiday(dt) for c in self._constituent_calendars)
        if self._cal_type == CalendarTypes.NONE:
            return False
        if self._cal_type == CalendarTypes.WEEKEND:
            return dt.is_weekend()
        raise LibError("Unknown calendar")

    def __repr__(self):
        return self._cal_type.name
