"""Discount curve queries on a node set ``(_times, _dfs)`` - the non-AD API of the reference's curves.

Mirror of cavour/market/curves/discount_curve.py: `df` :300-315, `df_ad` / `_df_ad` / `_linear_forward_interp`
:317-415, `_df` :418-436, `zero_rate` / `cc_rate` :186-222, `swap_rate` :226-296, `fwd` / `_fwd` :452-493,
`bump` :497-516, `fwd_rate` :520-560.  Host-side convenience: the valuation
engine does not use these node sets, it bootstraps its own knot grid.

Quirks kept on purpose: `df()` converts dates to times with ACT/ACT ISDA unless told otherwise, whatever
the curve's own day count; `df_ad` always uses linear interpolation of the segments' forward rates,
regardless of the curve's `interp_type`.  PCHIP / cubic interpolators are out of scope
and raise.
"""
from __future__ import annotations

import numpy as np

from ...utils.date import Date
from ...utils.day_count import DayCount, DayCountTypes
from ...utils.error import LibError
from ...utils.frequency import FrequencyTypes, annual_frequency
from ...utils.global_types import InterpTypes
from ...utils.global_vars import gDaysInYear, g_small
from ...utils.helpers import times_from_dates
from ...utils.schedule import Schedule
from .interpolator import interpolate

_DIRECT = (InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_ZERO_RATES, InterpTypes.LINEAR_FWD_RATES)


class DiscountCurve:
    def __init__(self, value_dt: Date, df_dts: list, df_values, interp_type: InterpTypes = InterpTypes.FLAT_FWD_RATES):
        """Curve from year offsets ``df_dts`` and discount factors (discount_curve.py:40-91)."""
        if len(df_dts) < 1:
            raise LibError("Times has zero length")
        if len(df_dts) != len(df_values):
            raise LibError("Times and Values are not the same")
        times, dfs = [0.0], [1.0]
        dates = value_dt.add_years(list(df_dts))
        start = 0
        if dates[0] == value_dt:
            dfs[0] = df_values[0]
            start = 1
        for i in range(start, len(df_dts)):
            times.append((dates[i] - value_dt) / gDaysInYear)
            dfs.append(df_values[i])
        self._times = np.array(times, dtype=np.float64)
        if np.any(np.diff(self._times) <= 0.0):
            raise LibError("Times are not sorted in increasing order")
        self._df_dts = df_dts
        self._value_dt = value_dt
        self._dfs = np.array(dfs, dtype=np.float64)
        self._interp_type = interp_type
        self._freq_type = FrequencyTypes.CONTINUOUS
        self._dc_type = DayCountTypes.ACT_ACT_ISDA

    @property
    def value_dt(self):
        return self._value_dt

    # ------------------------------------------------------------------ discount factors
    def df(self, dt, day_count=DayCountTypes.ACT_ACT_ISDA):
        times = times_from_dates(dt, self._value_dt, day_count)
        dfs = self._df(times)
        return dfs if isinstance(dfs, float) else np.array(dfs)

    def _df(self, t):
        if self._interp_type not in _DIRECT:
            raise LibError(f"{self._interp_type} needs the spline interpolators, which are out of scope")
        if isinstance(t, (int, np.integer)):
            t = float(t)
        out = interpolate(t, np.asarray(self._times), np.asarray(self._dfs), self._interp_type.value)
        return float(out) if isinstance(t, (float, np.float64)) else out

    def df_ad(self, dt, day_count=DayCountTypes.ACT_ACT_ISDA):
        """Discount factor(s) at TIME(s) ``dt`` in years (the argument is not converted; :317-340)."""
        return self._df_ad(dt)

    def _df_ad(self, t):
        return self._linear_forward_interp(t, self._times, self._dfs)

    @staticmethod
    def _linear_forward_interp(t, times, dfs):
        times = np.asarray(times, dtype=np.float64)
        dfs = np.asarray(dfs, dtype=np.float64)
        fwd_rates = -np.log(dfs[1:] / dfs[:-1]) / (times[1:] - times[:-1])
        fwd = _interp_like_jax(t, times[:-1], fwd_rates)
        i0 = np.searchsorted(times, t, side="right") - 1
        out = dfs[i0] * np.exp(-fwd * (t - times[i0]))
        return float(out) if np.ndim(out) == 0 else out

    # ------------------------------------------------------------------ rates
    def _zero_to_df(self, value_dt, rates, times, freq_type, dc_type):
        if isinstance(times, float):
            times = np.array([times])
        t = np.maximum(times, g_small)
        f = annual_frequency(freq_type)
        if freq_type == FrequencyTypes.CONTINUOUS:
            return np.exp(-rates * t)
        if freq_type == FrequencyTypes.SIMPLE:
            return 1.0 / (1.0 + rates * t)
        if freq_type in (FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL, FrequencyTypes.QUARTERLY,
                         FrequencyTypes.MONTHLY):
            return 1.0 / np.power(1.0 + rates / f, f * t)
        raise LibError("Unknown Frequency type")

    def _df_to_zero(self, dfs, maturity_dts, freq_type, dc_type):
        f = annual_frequency(freq_type)
        date_list = [maturity_dts] if isinstance(maturity_dts, Date) else maturity_dts
        df_list = [dfs] if isinstance(dfs, float) else dfs
        if len(date_list) != len(df_list):
            raise LibError("Date list and df list do not have same length")
        times = times_from_dates(date_list, self._value_dt, dc_type)
        out = []
        for df, t in zip(df_list, np.atleast_1d(times)):
            t = max(t, g_small)
            if freq_type == FrequencyTypes.CONTINUOUS:
                out.append(-np.log(df) / t)
            elif freq_type == FrequencyTypes.SIMPLE:
                out.append((1.0 / df - 1.0) / t)
            else:
                out.append((np.power(df, -1.0 / (t * f)) - 1.0) * f)
        return np.array(out)

    def zero_rate(self, dts, freq_type: FrequencyTypes = FrequencyTypes.CONTINUOUS,
                  dc_type: DayCountTypes = DayCountTypes.ACT_360):
        if not isinstance(freq_type, FrequencyTypes):
            raise LibError("Invalid Frequency type.")
        if not isinstance(dc_type, DayCountTypes):
            raise LibError("Invalid Day Count type.")
        zero_rates = self._df_to_zero(self.df(dts), dts, freq_type, dc_type)
        return zero_rates[0] if isinstance(dts, Date) else np.array(zero_rates)

    def cc_rate(self, dts, dc_type: DayCountTypes = DayCountTypes.SIMPLE):
        return self.zero_rate(dts, FrequencyTypes.CONTINUOUS, dc_type)

    def swap_rate(self, effective_dt: Date, maturity_dt, freq_type=FrequencyTypes.ANNUAL,
                  dc_type: DayCountTypes = DayCountTypes.THIRTY_E_360):
        """Par rate of an unadjusted-schedule swap (:226-296); always returns an array, as the reference."""
        if effective_dt < self._value_dt:
            raise LibError("Swap starts before the curve valuation date.")
        if not isinstance(freq_type, FrequencyTypes):
            raise LibError("Invalid Frequency type.")
        if freq_type == FrequencyTypes.SIMPLE:
            raise LibError("Cannot calculate par rate with simple yield freq.")
        if freq_type == FrequencyTypes.CONTINUOUS:
            raise LibError("Cannot calculate par rate with continuous freq.")
        maturity_dts = [maturity_dt] if isinstance(maturity_dt, Date) else maturity_dt
        par_rates = []
        for mat in maturity_dts:
            if mat <= effective_dt:
                raise LibError("Maturity date is before the swap start date.")
            flow_dts = Schedule(effective_dt, mat, freq_type).generate()
            flow_dts[0] = effective_dt
            counter = DayCount(dc_type)
            prev_dt, pv01, df = flow_dts[0], 0.0, 1.0
            for next_dt in flow_dts[1:]:
                df = self.df(next_dt)
                pv01 += counter.year_frac(prev_dt, next_dt)[0] * df
                prev_dt = next_dt
            par_rates.append(0.0 if abs(pv01) < g_small else (self.df(effective_dt) - df) / pv01)
        return np.array(par_rates)

    def fwd(self, dts):
        """Continuously compounded one-day forward rate(s) (:452-476)."""
        single = isinstance(dts, Date)
        plus_one = [dts.add_days(1)] if single else [d.add_days(1) for d in dts]
        df1, df2 = self.df(dts), self.df(plus_one)
        fwd = np.log(df1 / df2) / (1.0 * (1.0 / gDaysInYear))
        return fwd[0] if single else np.array(fwd)

    def _fwd(self, times):
        dt = 1e-6
        times = np.maximum(times, dt)
        return np.log(self._df(times - dt) / self._df(times + dt)) / (2.0 * dt)

    def bump(self, bump_size: float):
        """A curve whose continuously compounded rates are shifted by ``bump_size`` (:497-516)."""
        times = np.asarray(self._times, dtype=np.float64).tolist()
        values = np.asarray(self._dfs, dtype=np.float64) * np.exp(-bump_size * np.asarray(times))
        return DiscountCurve(self._value_dt, times, values, self._interp_type)

    def fwd_rate(self, start_dt, date_or_tenor, dc_type: DayCountTypes = DayCountTypes.ACT_360):
        if isinstance(start_dt, Date):
            start_dts = [start_dt]
        elif isinstance(start_dt, list):
            start_dts = start_dt
        else:
            raise LibError("Start date and end date must be same types.")
        counter = DayCount(dc_type)
        out = []
        for i, dt1 in enumerate(start_dts):
            if isinstance(date_or_tenor, str):
                dt2 = dt1.add_tenor(date_or_tenor)
            elif isinstance(date_or_tenor, Date):
                dt2 = date_or_tenor
            else:
                dt2 = date_or_tenor[i]
            out.append((self.df(dt1) / self.df(dt2) - 1.0) / counter.year_frac(dt1, dt2)[0])
        return out[0] if isinstance(start_dt, Date) else np.array(out)


def _interp_like_jax(x, xp, fp):
    """`jax.numpy.interp` (the reference's interpolation primitive): clamp outside ``xp``, and inside
    ``fp[i-1] + (x - xp[i-1]) / (xp[i] - xp[i-1]) * (fp[i] - fp[i-1])`` with ``i = searchsorted(xp, x, 'right')``
    clipped to ``[1, len - 1]`` - the same piecewise-linear function as numpy.interp with its own rounding."""
    xp = np.asarray(xp, dtype=np.float64)
    fp = np.asarray(fp, dtype=np.float64)
    x_arr = np.asarray(x, dtype=np.float64)
    if xp.size == 1:
        return np.full_like(x_arr, fp[0]) if x_arr.ndim else float(fp[0])
    i = np.clip(np.searchsorted(xp, x_arr, side="right"), 1, xp.size - 1)
    dx = xp[i] - xp[i - 1]
    delta = x_arr - xp[i - 1]
    with np.errstate(divide="ignore", invalid="ignore"):
        f = np.where(np.abs(dx) <= np.finfo(np.float64).eps, fp[i], fp[i - 1] + (delta / dx) * (fp[i] - fp[i - 1]))
    f = np.where(x_arr < xp[0], fp[0], np.where(x_arr > xp[-1], fp[-1], f))
    return f if x_arr.ndim else float(f)
