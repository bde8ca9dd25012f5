"""OIS curve object: the per-curve inputs the valuation engine consumes, and the curve's own node set.

Engine inputs (cavour/trades/rates/ois_curve.py:113-154) - the engine never
uses a pre-built discount curve, it re-bootstraps its own knot grid from three lists held here:

* ``swap_rates[i]``  - fixed coupon of calibration swap i (decimal),
* ``swap_times[i]``  - (last adjusted fixed date - value date) / days-in-year of the float leg's day count,
* ``year_fracs[i]``  - the fixed-leg accrual fractions of swap i.

Own node set (ois_curve.py:156-212): a second, different bootstrap that
backs the non-AD queries `df()` / `df_ad()` / `zero_rate()` ... inherited from `DiscountCurve`.  Its nodes
are the pillar maturities plus, for a swap whose previous coupon date is not yet a node, that coupon date
- valued with the log-linearly interpolated par rate of the neighbouring pillars - recursively.  Nodes are
keyed by ``round(time, 2)`` like the reference; node times of intermediate coupons are running sums of the
swap's accrual fractions, pillar times are date differences over the day count's year.
"""
import numpy as np

from ...market.curves.discount_curve import DiscountCurve, _interp_like_jax
from ...utils.date import Date
from ...utils.day_count import DayCount
from ...utils.error import LibError
from ...utils.global_types import InterpTypes
from ...utils.helpers import check_argument_types


class OISCurve(DiscountCurve):
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
        self._build_curve_ad(self.swap_rates)

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

    def _build_curve_ad(self, swap_rates):
        """Bootstrap the curve's own nodes (ois_curve.py:156-212).

        The reference recurses from each pillar back through its missing coupon dates; here the missing
        dates of a pillar are collected first and then valued oldest first - the same nodes in the same
        order with the same arithmetic, without the recursion."""
        pillar_times = np.array(self.swap_times, dtype=np.float64)
        log_rates = np.log(np.array(swap_rates, dtype=np.float64))
        times, dfs, repr_dfs = [0.0], [1.0], [1.0]
        pv01_at = {}                                     # round(node time, 2) -> PV01 up to that node

        def add_node(t_mat, rate, acc, pv01_prev):
            df = (1.0 - rate * pv01_prev) / (acc * rate + 1)
            times.append(t_mat)
            dfs.append(df)
            pv01_at[round(t_mat, 2)] = pv01_prev + acc * df
            return df

        for i, fracs in enumerate(self.year_fracs):
            rate = swap_rates[i]
            if len(fracs) == 1:
                acc = fracs[0]
                df = 1 / (acc * rate + 1.0)
                times.append(self.swap_times[i])
                dfs.append(df)
                pv01_at[round(self.swap_times[i], 2)] = acc * df
                repr_dfs.append(df)
                continue
            # coupon dates of this swap, newest first, until one is already a node
            missing = []
            step = 0
            while True:
                before = sum(fracs[:-1 - step])
                if round(before, 2) in pv01_at:
                    break
                if step + 1 >= len(fracs):
                    # the reference would index past the start of the accrual list here
                    raise LibError("OISCurve: a calibration swap's first coupon date is not a node of an earlier swap")
                step += 1
                missing.append((before, step))
            for before, s in reversed(missing):
                earlier = sum(fracs[:-1 - s])
                # intermediate node: par rate interpolated log-linearly in the pillar rates
                r_mid = float(np.exp(_interp_like_jax(before, pillar_times, log_rates)))
                add_node(before, r_mid, fracs[-1 - s], pv01_at[round(earlier, 2)])
            before = sum(fracs[:-1])
            df = add_node(self.swap_times[i], rate, fracs[-1], pv01_at[round(before, 2)])
            repr_dfs.append(df)

        self._times = np.array(times, dtype=np.float64)
        self._dfs = np.array(dfs, dtype=np.float64)
        self._repr_dfs = np.array(repr_dfs, dtype=np.float64)
        return self._times, self._dfs

    def _check_refits(self, swap_tol):
        """Each calibration swap must reprice to zero off the curve's own nodes (ois_curve.py:344-358)."""
        for swap in self._used_swaps:
            v = swap.value(swap._effective_dt, self, None) / swap._notional
            if abs(v) > swap_tol:
                raise LibError(f"Swap with maturity {swap._maturity_dt} not repriced. Difference is {abs(v)}")

    def __repr__(self):
        return (f"OISCurve(value_dt={self._value_dt}, pillars={len(self.swap_rates)}, "
                f"interp={self._interp_type.name})")
