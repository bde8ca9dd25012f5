"""Date, calendar, schedule, day-count utilities and the shared enums."""
from .calendar import BusDayAdjustTypes, Calendar, CalendarTypes, DateGenRuleTypes
from .currency import CurrencyTypes
from .date import Date, DateFormatTypes, datediff, set_date_format
from .day_count import DayCount, DayCountTypes
from .error import LibError
from .frequency import FrequencyTypes, annual_frequency
from .global_types import (CollateralType, CurveTypes, InstrumentTypes, InterpTypes,
                           RequestTypes, SwapTypes)
from .global_vars import ONE_MILLION, g_small, gDaysInYear
from .helpers import check_argument_types, times_from_dates, to_tenor
from .schedule import Schedule
