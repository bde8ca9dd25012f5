"""OIS curve object: the per-curve inputs the valuation engine consumes.

The engine never uses a pre-built discount curve; it re-bootstraps its own knot
grid from three lists held here (SURVEY.md section 8(a) row A,
cavour/trades/rates/ois_curve.py:113-154):

* ``swap_rates[i]``  - fixed coupon of calibration swap i (decimal),
* ``swap_times[i]``  - (last adjusted fixed date - value date) / days-in-year of
  the float leg's day count,
* ``year_fracs[i]``  - the fixed-leg accrual fractions of swap i.

The reference's `OISCurve` additionally runs a second, recursive bootstrap on
its own de-duplicated node set for the non-AD `df()` API; that is a "next" row
 and is not built here.
"""
from ...utils.date import Date
from ...utils.day_count import DayCount
from ...utils.global_types import InterpTypes
from ...utils.helpers import check_argument_types


class OISCurve:
    def __init__(self,
                 value_dt: Date,
                 ois_swaps: list,
                 interp_type: InterpTypes = InterpTypes.FLAT_FWD_RATES,
                 check_refit: bool = False):
        check_argument_types(self.__init__, locals())
        self._value_dt = value_dt
        self._used_swaps = ois_swaps
        self._interp_type = interp_type
        self._check_refit = check_refit
        self._prepare_curve_builder_inputs()

    def _prepare_curve_builder_inputs(self):
        self._dc_type = self._used_swaps[0]._float_leg._dc_type
        days_in_year = DayCount(self._dc_type).days_in_year()
        self.swap_rates = []
        self.swap_times = []
        self.year_fracs = []
        for swap in self._used_swaps:
            last_fixed_dt = swap._adjusted_fixed_dts[-1]
            self.swap_times.append((last_fixed_dt - self._value_dt) / days_in_year)
            self.swap_rates.append(swap._fixed_coupon)
            self.year_fracs.append(swap._fixed_leg._year_fracs)
        return self.swap_rates

    def __repr__(self):
        return (f"OISCurve(value_dt={self._value_dt}, pillars={len(self.swap_rates)}, "
                f"interp={self._interp_type.name})")
