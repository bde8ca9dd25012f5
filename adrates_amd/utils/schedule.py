"""ISDA-style coupon date schedules (mirrors cavour/utils/schedule.py:53-270).

A schedule holds the previous coupon date followed by every flow date up to
the termination date.  BACKWARD generation steps back from the termination
date in whole periods (a short stub lands at the front); FORWARD steps from
the effective date.  Interior dates are business-day adjusted; the termination
date only when ``adjust_termination_dt`` is set (the default).
"""
from .calendar import BusDayAdjustTypes, Calendar, CalendarTypes, DateGenRuleTypes
from .date import Date
from .error import LibError
from .frequency import FrequencyTypes, annual_frequency
from .helpers import check_argument_types, label_to_string


class Schedule:
    def __init__(self,
                 effective_dt: Date,
                 termination_dt: Date,
                 freq_type: FrequencyTypes = FrequencyTypes.ANNUAL,
                 cal_type: CalendarTypes = CalendarTypes.WEEKEND,
                 bd_type: BusDayAdjustTypes = BusDayAdjustTypes.FOLLOWING,
                 dg_type: DateGenRuleTypes = DateGenRuleTypes.BACKWARD,
                 adjust_termination_dt: bool = True,
                 end_of_month: bool = False,
                 first_dt=None,
                 next_to_last_dt=None):
        check_argument_types(self.__init__, locals())

        if effective_dt >= termination_dt:
            raise LibError("Effective date must be before termination date.")

        self._effective_dt = effective_dt
        self._termination_dt = termination_dt

        # Long-stub controls are accepted and validated but, as in the
        # reference (schedule.py:115-133), have no effect on the dates.
        if first_dt is None:
            self._first_dt = effective_dt
        elif effective_dt < first_dt < termination_dt:
            self._first_dt = first_dt
            print("FIRST DATE NOT IMPLEMENTED")
        else:
            raise LibError("First date must be after effective date and before termination date")

        if next_to_last_dt is None:
            self._next_to_last_dt = termination_dt
        elif effective_dt < next_to_last_dt < termination_dt:
            self._next_to_last_dt = next_to_last_dt
            print("NEXT TO LAST DATE NOT IMPLEMENTED")
        else:
            raise LibError("Next to last date must be after effective date and before termination date")

        self._freq_type = freq_type
        self._cal_type = cal_type
        self._bd_type = bd_type
        self._dg_type = dg_type
        self._adjust_termination_dt = adjust_termination_dt
        self._end_of_month = end_of_month is True
        self._adjusted_dts = None
        self.generate()

    def schedule_dts(self):
        if self._adjusted_dts is None:
            self.generate()
        return self._adjusted_dts

    def generate(self):
        """Build ``_adjusted_dts`` (cavour/utils/schedule.py:163-270)."""
        calendar = Calendar(self._cal_type)
        months_per_period = int(12 / annual_frequency(self._freq_type))
        adjusted = []

        if self._dg_type == DateGenRuleTypes.BACKWARD:
            # unadjusted dates, latest first; the loop's last value is the
            # first date at or before the effective date (the PCD)
            rolled_back = []
            cursor = self._termination_dt
            k = 0
            while cursor > self._effective_dt:
                rolled_back.append(cursor)
                k += 1
                cursor = self._termination_dt.add_months(-months_per_period * k)
                if self._end_of_month:
                    cursor = cursor.eom()
            rolled_back.append(cursor)

            adjusted.append(rolled_back[-1])                      # PCD, never adjusted
            for unadj in reversed(rolled_back[1:-1]):             # interior flows
                adjusted.append(calendar.adjust(unadj, self._bd_type))
            adjusted.append(self._termination_dt)

        elif self._dg_type == DateGenRuleTypes.FORWARD:
            # The reference's forward walk (schedule.py:210-232) records the
            # effective date twice before stepping; the duplicate is dropped
            # again by the monotonicity pass below.
            walked = [self._effective_dt]
            cursor = self._effective_dt
            k = 1
            while cursor < self._termination_dt:
                walked.append(cursor)
                cursor = self._effective_dt.add_months(months_per_period * k)
                k += 1
            for unadj in walked[1:]:
                adjusted.append(calendar.adjust(unadj, self._bd_type))
            adjusted.append(self._termination_dt)

        if adjusted[0] < self._effective_dt:
            adjusted[0] = self._effective_dt

        if self._adjust_termination_dt is True:
            self._termination_dt = calendar.adjust(self._termination_dt, self._bd_type)
            adjusted[-1] = self._termination_dt

        if len(adjusted) < 2:
            raise LibError("Schedule has two dates only.")

        # Equal neighbours drop the *front* element (sic, schedule.py:256-266:
        # the reference pops index 0 while iterating over a copy); decreasing
        # dates are an error.
        self._adjusted_dts = adjusted
        prev = adjusted[0]
        for dt in list(adjusted[1:]):
            if dt == prev:
                self._adjusted_dts.pop(0)
            if dt < prev:
                raise LibError("Dates are not monotonic")
            prev = dt

        return self._adjusted_dts

    def __repr__(self):
        s = label_to_string("OBJECT TYPE", type(self).__name__)
        s += label_to_string("EFFECTIVE DATE", self._effective_dt)
        s += label_to_string("END DATE", self._termination_dt)
        s += label_to_string("FREQUENCY", self._freq_type)
        s += label_to_string("CALENDAR", self._cal_type)
        s += label_to_string("BUSDAYRULE", self._bd_type)
        s += label_to_string("DATEGENRULE", self._dg_type)
        s += label_to_string("ADJUST TERM DATE", self._adjust_termination_dt)
        s += label_to_string("END OF MONTH", self._end_of_month, "")
        return s

    def _print(self):
        print(self)
