"""Coupon schedules of many legs at once, on NumPy arrays of Excel serials.

The array counterpart of `Date` / `Calendar` / `Schedule` / `SwapFloatLeg.generate_payment_dts` for the conventions
books are made of: BACKWARD date generation without end-of-month rolling, the WEEKEND (or NONE) calendar, any
business-day rule, payment lags in business days.  Every function reproduces the object path date for date
(cavour/utils/date.py:597-653, 796-879; calendar.py:139-253; schedule.py:163-270; swap_float_leg.py:130-186), which
`tests/test_schedule_np.py` checks on random terms; legs the arrays cannot express (a schedule whose neighbouring
dates coincide - the reference then drops the FRONT date, schedule.py:256-266) are reported in a mask and left to the
object path by the callers (`xccy_engine.raw_from_terms`).
"""
import numpy as np

from .calendar import BusDayAdjustTypes
from .error import LibError

_EPOCH_SERIAL = 25569                      # Excel serial of 1970-01-01 (serials after the phantom 29-Feb-1900)
_MIN_SERIAL = 61                           # 1-Mar-1900: below it the Lotus off-by-one applies (date.py:137-181)


_FIRST_YEAR, _LAST_YEAR = 1900, 2300
# Excel serial of the first day of every month from Jan of _FIRST_YEAR to Jan of _LAST_YEAR + 1, built once
_MONTH_START = (np.arange((_FIRST_YEAR - 1970) * 12, (_LAST_YEAR + 1 - 1970) * 12 + 1).astype("datetime64[M]")
                .astype("datetime64[D]").astype(np.int64) + _EPOCH_SERIAL)
_MONTH_BASE = _FIRST_YEAR * 12
_MONTH_LENGTH = np.diff(_MONTH_START)


def ymd_from_serial(serial):
    """(year, month, day) arrays of Excel serials (>= 1-Mar-1900)."""
    s = np.asarray(serial, dtype=np.int64)
    if s.size and (s.min() < _MIN_SERIAL or s.max() >= _MONTH_START[-1]):
        raise LibError("schedule_np: dates before 1-Mar-1900 or after %d are not supported" % _LAST_YEAR)
    at = np.searchsorted(_MONTH_START, s, side="right") - 1
    y, m0 = np.divmod(at + _MONTH_BASE, 12)
    return y, m0 + 1, s - _MONTH_START[at] + 1


def _month_start_serial(month_index):
    """Excel serial of the first day of month ``month_index`` (months since 0000-01)."""
    at = month_index - _MONTH_BASE
    if at.size and (at.min() < 2 or at.max() >= _MONTH_START.shape[0]):
        raise LibError("schedule_np: dates before 1-Mar-1900 or after %d are not supported" % _LAST_YEAR)
    return _MONTH_START[at]


def days_in_month(month_index):
    return _month_start_serial(month_index + 1) - _month_start_serial(month_index)


def serial_of(month_index, day):
    """Serial of day ``min(day, month length)`` in month ``month_index`` (months since 0000-01): the clamping of
    `Date._shift_months`."""
    at = month_index - _MONTH_BASE
    if at.size and (at.min() < 2 or at.max() >= _MONTH_LENGTH.shape[0]):
        raise LibError("schedule_np: dates before 1-Mar-1900 or after %d are not supported" % _LAST_YEAR)
    return _MONTH_START[at] + np.minimum(day, _MONTH_LENGTH[at]) - 1


def weekday(serial):
    return (serial + 5) % 7                               # Monday = 0 (date.py:212-216)


def _roll(serial, step):
    wd = weekday(serial)
    if step > 0:
        return serial + np.where(wd == 5, 2, np.where(wd == 6, 1, 0))
    return serial - np.where(wd == 5, 1, np.where(wd == 6, 2, 0))


def adjust(serial, bd_type, weekend_calendar=True):
    """`Calendar(WEEKEND).adjust` on arrays (calendar.py:139-217); the NONE calendar leaves dates alone."""
    if not weekend_calendar or bd_type == BusDayAdjustTypes.NONE:
        return serial
    if bd_type == BusDayAdjustTypes.FOLLOWING:
        return _roll(serial, +1)
    if bd_type == BusDayAdjustTypes.PRECEDING:
        return _roll(serial, -1)
    if bd_type in (BusDayAdjustTypes.MODIFIED_FOLLOWING, BusDayAdjustTypes.MODIFIED_PRECEDING):
        step = +1 if bd_type == BusDayAdjustTypes.MODIFIED_FOLLOWING else -1
        rolled = _roll(serial, step)
        month = lambda s: np.searchsorted(_MONTH_START, s, side="right")
        return np.where(month(rolled) != month(serial), _roll(serial, -step), rolled)
    raise LibError("Unknown adjustment convention" + str(bd_type))


def add_business_days(serial, num_days):
    """`Calendar.add_business_days` on arrays (calendar.py:221-253): ``num_days`` weekdays on from each date."""
    num_days = np.asarray(num_days, dtype=np.int64)
    if not num_days.any():
        return serial
    out = serial.copy()
    left = np.abs(num_days)
    step = np.where(num_days >= 0, 1, -1)
    while True:
        go = left > 0
        if not go.any():
            return out
        nxt = out + step
        wd = weekday(nxt)                       # (the NONE calendar skips weekends here too, calendar.py:266-269)
        nxt = nxt + np.where(step > 0, np.where(wd == 5, 2, np.where(wd == 6, 1, 0)),
                             -np.where(wd == 5, 1, np.where(wd == 6, 2, 0)))
        out = np.where(go, nxt, out)
        left = left - go


def add_tenor(serial, count, unit):
    """`Date.add_tenor` for arrays of tenors ``count`` x ``unit`` (unit: array of 'D', 'W', 'M', 'Y' codes 0-3)."""
    s = np.asarray(serial, dtype=np.int64)
    count = np.asarray(count, dtype=np.int64)
    unit = np.asarray(unit, dtype=np.int64)
    y, m, d = ymd_from_serial(s)
    idx = y * 12 + (m - 1)
    out = np.where(unit == 0, s + count, s + 7 * count)
    monthly = unit == 2
    yearly = unit == 3
    if monthly.any():
        out[monthly] = serial_of(idx[monthly] + count[monthly], d[monthly])         # day restored where the month allows
    if yearly.any():
        # twelve months at a time, clamping at every step and never restoring: 29-Feb lands on, and stays on, the 28th
        d_y = np.where((m == 2) & (d == 29) & (count != 0), 28, d)
        out[yearly] = serial_of(idx[yearly] + 12 * count[yearly], d_y[yearly])
    return out


_UNITS = {"D": 0, "W": 1, "M": 2, "Y": 3}


def parse_tenors(table):
    """(count, unit code) of each tenor string of a table (the parsing of `Date.add_tenor`)."""
    counts, units = [], []
    for raw in table:
        if not isinstance(raw, str):
            raise LibError("Tenor must be a string e.g. '5Y'")
        ts = raw.upper()
        if ts in ("ON", "TN"):
            counts.append(1); units.append(0)
            continue
        if ts[-1] not in _UNITS:
            raise LibError("Unknown tenor type in " + raw)
        counts.append(int(ts[:-1])); units.append(_UNITS[ts[-1]])
    return np.asarray(counts, dtype=np.int64), np.asarray(units, dtype=np.int64)


def backward_schedules(effective, termination, months_per_period, bd_type, weekend_calendar=True):
    """`Schedule(effective, termination, freq, cal, bd, BACKWARD)._adjusted_dts` of many legs.

    Returns ``(off, dates, plain)``: CSR offsets and serials of every leg's schedule (previous coupon date - the
    effective date - first, adjusted termination date last) and a mask of the legs whose dates are strictly
    increasing; for the others (``plain`` false) the reference's de-duplication quirk applies and the entries here
    are not its schedule."""
    eff = np.asarray(effective, dtype=np.int64)
    term = np.asarray(termination, dtype=np.int64)
    mpp = np.asarray(months_per_period, dtype=np.int64)
    if (eff >= term).any():
        raise LibError("Effective date must be before termination date.")
    ey, em, ed = ymd_from_serial(eff)
    ty, tm, td = ymd_from_serial(term)
    e_idx, t_idx = ey * 12 + em - 1, ty * 12 + tm - 1
    gap = t_idx - e_idx
    # number of unadjusted dates termination - k periods that lie after the effective date
    whole = gap // mpp
    k_same = whole                                                         # the k whose month is the effective month, if any
    lands = (gap % mpp == 0)
    same_month_later = lands & (np.minimum(td, days_in_month(t_idx - k_same * mpp)) > ed)
    n_flows = np.where(lands, whole + same_month_later, whole + 1)
    n_dates = n_flows + 1
    off = np.concatenate(([0], np.cumsum(n_dates))).astype(np.int64)
    j = np.arange(off[-1], dtype=np.int64) - np.repeat(off[:-1], n_dates)       # 0 = previous coupon date
    month = np.repeat(t_idx - n_flows * mpp, n_dates) + j * np.repeat(mpp, n_dates)
    dates = adjust(serial_of(month, np.repeat(td, n_dates)), bd_type, weekend_calendar)
    dates[off[:-1]] = eff                                   # never adjusted; interior dates and the termination date are
    increasing = np.empty(off[-1], dtype=bool)
    increasing[0] = True
    np.greater(dates[1:], dates[:-1], out=increasing[1:])
    increasing[off[:-1]] = True
    return off, dates, np.logical_and.reduceat(increasing, off[:-1])


def leg_times(effective, termination, months_per_period, payment_lag, bd_type, weekend_calendar, denominator,
              value_serial, payment_denominator=None):
    """`leg_times_np` on the library's host threads (`adr_leg_counts_host` / `adr_leg_times_host`, csrc/book_host.cpp):
    the same arrays, bit for bit (tests/test_book_native.py), without the whole-book temporaries - 100 000 swaps x 2 legs
    in a few milliseconds instead of ~0.1 s."""
    from .. import _native
    eff = np.asarray(effective, dtype=np.int64)
    if eff.size and (eff >= np.asarray(termination, dtype=np.int64)).any():
        raise LibError("Effective date must be before termination date.")
    return _native.leg_times_host(eff, termination, months_per_period, payment_lag, bd_type.value, weekend_calendar,
                                  denominator, value_serial, payment_denominator)


def leg_times_np(effective, termination, months_per_period, payment_lag, bd_type, weekend_calendar, denominator,
                 value_serial, payment_denominator=None):
    """The arrays `SwapFixedLeg.generate_payments` / `SwapFloatLeg.generate_payment_dts` produce, for many legs on a day
    count with a fixed denominator (ACT/365F, ACT/360, SIMPLE): CSR offsets over the coupons, payment / accrual start /
    accrual end times as year fractions from ``value_serial`` (``payment_denominator``: another day count's denominator
    for the payment times, e.g. the discounting curve's), accrual fractions, and the mask of legs whose schedule is
    plain (see `backward_schedules`).  ``denominator``: per leg."""
    off, dts, plain = backward_schedules(effective, termination, months_per_period, bd_type, weekend_calendar)
    is_start = np.ones(off[-1], dtype=bool)
    is_start[off[1:] - 1] = False
    is_end = np.ones(off[-1], dtype=bool)
    is_end[off[:-1]] = False
    start, end = dts[is_start], dts[is_end]
    lens = off[1:] - off[:-1] - 1
    pay = add_business_days(end, np.repeat(np.asarray(payment_lag, dtype=np.int64), lens))
    d = np.repeat(np.asarray(denominator), lens)
    dp = d if payment_denominator is None else payment_denominator
    coupons = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    return (coupons, (pay - value_serial) / dp, (start - value_serial) / d, (end - value_serial) / d, (end - start) / d, plain)
