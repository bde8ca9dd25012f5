"""Small host-side helpers of the valuation path.

* `times_from_dates`  - cavour/utils/helpers.py:154-197
* `to_tenor`          - cavour/utils/helpers.py:201-242 (pillar labels of the ladders)
* `check_argument_types` - cavour/utils/helpers.py:618-636 ("Argument Type Error")
"""
import math
from typing import List, Union

import numpy as np

from .date import Date
from .day_count import DayCount, DayCountTypes
from .error import LibError
from .global_vars import gDaysInYear


def times_from_dates(dt, value_dt: Date, day_count_type: DayCountTypes = None):
    """Year fraction(s) from ``value_dt`` to a date or list of dates.

    With no day count the plain day difference over 365 is used.  A single
    ``Date`` gives a float, a list gives a numpy array."""
    if not isinstance(value_dt, Date):
        raise LibError("Valuation date is not a Date")
    counter = None if day_count_type is None else DayCount(day_count_type)

    def one(d):
        if counter is None:
            return (d - value_dt) / gDaysInYear
        return counter.year_frac(value_dt, d)[0]

    if isinstance(dt, Date):
        return one(dt)
    if isinstance(dt, list) and isinstance(dt[0], Date):
        return np.array([one(d) for d in dt])
    if isinstance(dt, np.ndarray):
        raise LibError("You passed an ndarray instead of dates.")
    raise LibError("Discount factor must take dates.")


def _tenor_label(val: float) -> str:
    if val < 1 / 12:
        return f"{math.ceil(val * 365 / 7)}W"
    if val < 1:
        return f"{max(int(round(val * 12)), 1)}M"
    years = int(math.floor(val))
    months = int(round((val - years) * 12))
    if months == 12:
        years, months = years + 1, 0
    return f"{years}Y" if months == 0 else f"{years}Y{months}M"


def to_tenor(x: Union[float, List[float]]) -> Union[str, List[str]]:
    """Label a year fraction: weeks (rounded up) below one month, months below
    one year, otherwise years plus leftover months.  Note the reference's
    quirks, which result consumers rely on: 1D and 1W both become "1W" and a
    31-day 1M pillar becomes "5W" (SURVEY.md section 8(a) row K)."""
    if isinstance(x, list):
        return [_tenor_label(v) for v in x]
    return _tenor_label(x)


def _usable_type(t):
    """Turn an annotation into something `isinstance` accepts
    (cavour/utils/helpers.py:508-527)."""
    origin = getattr(t, "__origin__", None)
    if origin is not None:
        if origin is list:
            return (list, np.ndarray)
        if origin is dict:
            return dict
        if origin is Union:
            return tuple(_usable_type(a) for a in t.__args__)
        return t
    if t is float:
        return (int, float, np.float64)
    if isinstance(t, tuple):
        return tuple(_usable_type(a) for a in t)
    return t


def _flatten(tp):
    if isinstance(tp, tuple):
        out = []
        for a in tp:
            out.extend(_flatten(a))
        return tuple(out)
    return (tp,)


def check_argument_types(func, values):
    """Raise ``LibError("Argument Type Error")`` when an annotated argument of
    ``func`` was given a value of another type."""
    for name, annotation in getattr(func, "__annotations__", {}).items():
        if name not in values:
            continue
        allowed = _flatten(_usable_type(annotation))
        try:
            ok = isinstance(values[name], allowed)
        except TypeError:
            continue  # annotation is not a runtime-checkable type
        if not ok:
            print("ERROR with function arguments for", func.__name__)
            print("Please check inputs for argument >>", name, "<<")
            print("You have input an argument", values[name], "of type", type(values[name]))
            print("The allowed types are", allowed)
            raise LibError("Argument Type Error")


def label_to_string(label, value, separator="\n", list_format=False):
    """"LABEL: value" line used by the ``__repr__`` of trades and schedules."""
    if list_format and isinstance(value, (list, tuple, np.ndarray)):
        return "".join(f"{label}: {v}\n" for v in value)[:-1] + separator if len(value) else f"{label}: {separator}"
    return f"{label}: {value}{separator}"
