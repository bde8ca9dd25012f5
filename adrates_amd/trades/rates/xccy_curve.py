"""Cross-currency discount curve: foreign cash flows under domestic collateral, bootstrapped from basis swaps.

Host-side curve of SURVEY.md section 8(f) row 1 (cavour/trades/rates/xccy_curve.py).  The reference has two
builders that it documents as producing the same nodes - a loop with temporary curve objects (`_build_curve`,
:200-527) and a `lax.scan` it can differentiate (`_build_curve_ad`, :529-703, `_prepare_ad_inputs` :707-937,
`_run_jax_bootstrap_impl` :954-1206); `Model.build_xccy_curve` uses the second.  This module implements that
recurrence once, for both values of `use_ad`:

* the points are the foreign-leg payment dates of all calibration swaps on or after the value date - including
  the notional exchange at the effective date, which the reference finds in the leg's schedule because
  `SwapFloatLeg.value()` has inserted it there - sorted by (time, swap);
* between pillars the basis is flat: ``D_x(t) = D_x(t_prev) * D_f(t) / D_f(t_prev) * exp(-b * (t - t_prev))`` with
  ``b`` the spread of the point's own swap and ``t_prev`` the previous point of ANY swap;
* at a swap's maturity the par condition ``PV_dom + S * (PV_known + CF_last * D_x) = 0`` (foreign leg paid,
  ``S = spot_fx``) gives the node;
* nodes = the points after the value date, first occurrence per ``round(t, 4)``, behind ``(0, 1)``.

Where the reference calls `jacrev` / `jacfwd(jacrev)` / `jacrev(jacfwd)` on the scan, the recurrence is
evaluated on second-order forward-mode numbers (market/curves/jets.py), which yields the same four tensors:
``_jac_basis [K, P_b]``, ``_hess_basis [K, P_b, P_b]`` (w.r.t. the pillar spreads), ``_jac_foreign_curve_dfs
[K, K_f]`` and ``_mixed_hess_foreign_basis [K, P_b, K_f]`` (w.r.t. the foreign OIS curve's own node DFs).  Two
quirks of the reference's differentiated function are kept: w.r.t. the foreign node DFs only the discount
ratio ``D_f(t) / D_f(t_prev)`` is differentiated - the coupons' forward rates are computed from the undisturbed
grid (:637-668 override `df_foreign_ois` only) - and that ratio is re-evaluated by log-linear interpolation
at ACT/365 payment times, whereas the values use ``foreign_curve.df(date)`` in the foreign curve's day count.
"""
from __future__ import annotations

import numpy as np

from ...market.curves.discount_curve import DiscountCurve, _interp_like_jax
from ...market.curves.jets import Jet
from ...utils.date import Date
from ...utils.day_count import DayCountTypes
from ...utils.error import LibError
from ...utils.global_types import InterpTypes
from ...utils.global_vars import gDaysInYear
from ...utils.helpers import check_argument_types, times_from_dates


class XccyCurve(DiscountCurve):
    def __init__(self,
                 value_dt: Date,
                 basis_swaps: list,
                 domestic_curve: DiscountCurve,
                 foreign_curve: DiscountCurve,
                 spot_fx: float,
                 interp_type: InterpTypes = InterpTypes.FLAT_FWD_RATES,
                 check_refit: bool = False,
                 use_ad: bool = False):
        check_argument_types(self.__init__, locals())
        self._value_dt = value_dt
        self._domestic_curve = domestic_curve
        self._foreign_curve = foreign_curve
        self._spot_fx = spot_fx
        self._interp_type = interp_type
        self._check_refit = check_refit
        self._use_ad = use_ad
        self._used_swaps = sorted(basis_swaps, key=lambda s: s._maturity_dt)
        self._prepare_curve_builder_inputs()
        self._build_curve_ad()
        if check_refit:
            self._check_refits(1e-5)

    def _prepare_curve_builder_inputs(self):
        self._dc_type = DayCountTypes.ACT_365F
        self.basis_spreads = [s._foreign_spread for s in self._used_swaps]
        self.swap_times = [(s._maturity_dt - self._value_dt) / gDaysInYear for s in self._used_swaps]

    # ------------------------------------------------------------------ inputs of the recurrence
    def _payment_points(self):
        """One record per foreign-leg payment on or after the value date (xccy_curve.py:707-800)."""
        fc = self._foreign_curve
        points = []
        for s_idx, swap in enumerate(self._used_swaps):
            leg = swap._foreign_leg
            n = len(leg._payment_dts)
            pay = list(leg._payment_dts)
            start, end, fracs = list(leg._start_accrued_dts), list(leg._end_accrued_dts), list(leg._year_fracs)
            notionals = list(leg._notional_array) if len(leg._notional_array) else [leg._notional] * n
            if leg._notional_exchange and leg._effective_dt >= self._value_dt and pay[0] != leg._effective_dt:
                # what SwapFloatLeg.value() leaves behind in the reference (swap_float_leg.py:301-318)
                pay.insert(0, leg._effective_dt); start.insert(0, leg._effective_dt); end.insert(0, leg._effective_dt)
                fracs.insert(0, 0.0); notionals.insert(0, leg._notional)
            for j, dt in enumerate(pay):
                if not dt >= self._value_dt:
                    continue
                t = (dt - self._value_dt) / 365.0
                is_exchange = abs(fracs[j]) < 1e-10
                points.append(dict(
                    time=t, key=round(t, 4), swap=s_idx, basis=swap._foreign_spread,
                    is_maturity=dt == swap._maturity_dt, at_value_dt=dt == self._value_dt,
                    year_frac=fracs[j], notional=notionals[j], is_exchange=is_exchange,
                    is_last=(dt == swap._maturity_dt) and leg._notional_exchange,
                    spread_sens=0.0 if is_exchange else fracs[j] * notionals[j],
                    t_start=times_from_dates(start[j], self._value_dt, fc._dc_type),
                    t_end=times_from_dates(end[j], self._value_dt, fc._dc_type),
                    df_ois=fc.df(dt, fc._dc_type)))
        points.sort(key=lambda p: (p["time"], p["swap"]))
        return points

    def _bootstrap(self, points, pv_domestic, basis, df_ois, n):
        """The recurrence (xccy_curve.py:1002-1170) on floats (``n == 0``) or jets with ``n`` inputs.

        ``basis[s]`` and ``df_ois[i]`` are floats or jets; returns the discount factor of every point."""
        fc = self._foreign_curve
        f_times = np.asarray(fc._times, dtype=np.float64)
        log_f = np.log(np.asarray(fc._dfs, dtype=np.float64))
        one = (lambda x: x) if n == 0 else (lambda x: Jet.const(x, n))
        exp = (lambda x: float(np.exp(x))) if n == 0 else (lambda x: x.exp())
        S = self._spot_fx
        dfs = []
        pv_known = {}      # per swap: sum of the PV contributions of its earlier points
        cf_mat = {}
        prev = -1          # index of the previous point after the value date (any swap)
        for i, p in enumerate(points):
            b = basis[p["swap"]]
            # coupon: forward rate off the foreign OIS nodes, log-linear (flat forwards)
            if p["is_exchange"]:
                base_cf = p["notional"] if p["is_last"] else -p["notional"]
            else:
                df_s = float(np.exp(_interp_like_jax(p["t_start"], f_times, log_f)))
                df_e = float(np.exp(_interp_like_jax(p["t_end"], f_times, log_f)))
                yf = p["year_frac"]
                fwd = (df_s / df_e - 1.0) / max(yf, 1e-10) if yf > 1e-10 else 0.0
                base_cf = fwd * yf * p["notional"] + (p["notional"] if p["is_last"] else 0.0)
            cashflow = base_cf + b * p["spread_sens"]
            # flat forward basis from the previous point
            if prev < 0:
                df_mid = df_ois[i] * exp(-b * p["time"])
            else:
                df_mid = dfs[prev] * (df_ois[i] / df_ois[prev]) * exp(-b * (p["time"] - points[prev]["time"]))
            if p["at_value_dt"]:
                contrib = cashflow * 1.0
            elif not p["is_maturity"]:
                contrib = cashflow * df_mid
            else:
                contrib = one(0.0) if n else 0.0
            cf_here = cashflow if p["is_maturity"] else (one(0.0) if n else 0.0)
            known = pv_known.get(p["swap"], one(0.0) if n else 0.0) + contrib
            cf_last = cf_mat.get(p["swap"], one(0.0) if n else 0.0) + cf_here
            pv_known[p["swap"]], cf_mat[p["swap"]] = known, cf_last
            df_final = df_mid
            if p["is_maturity"]:
                # par condition with the foreign leg paid: PV_dom + S * (-(known) - cf_last * D) = 0
                denominator = S * (-1.0 * cf_last)
                den_v = denominator.v if n else denominator
                if abs(den_v) > 1e-12:
                    df_final = (-(pv_domestic[p["swap"]] + S * (-1.0 * known))) / denominator
            dfs.append(df_final)
            if not p["at_value_dt"]:
                prev = i
        return dfs

    def _build_curve_ad(self):
        points = self._payment_points()
        if not points:
            raise LibError("XccyCurve: no calibration payments on or after the value date")
        self._points = points
        pv_dom = [s._domestic_leg.value(self._value_dt, self._domestic_curve, self._domestic_curve)
                  for s in self._used_swaps]
        n_b = len(self._used_swaps)
        # node selection: points after the value date, first occurrence of each rounded time
        node_of, seen = [], set()
        for i, p in enumerate(points):
            if p["at_value_dt"] or p["key"] in seen:
                continue
            seen.add(p["key"])
            node_of.append(i)
        self._node_points = node_of

        values = self._bootstrap(points, pv_dom, list(self.basis_spreads), [p["df_ois"] for p in points], 0)
        self._times = np.array([0.0] + [points[i]["time"] for i in node_of], dtype=np.float64)
        self._dfs = np.array([1.0] + [values[i] for i in node_of], dtype=np.float64)
        self._repr_dfs = self._dfs

        # d / d(pillar spreads), values as above
        jb = [Jet.variable(b, s, n_b) for s, b in enumerate(self.basis_spreads)]
        out = self._bootstrap(points, pv_dom, jb, [Jet.const(p["df_ois"], n_b) for p in points], n_b)
        K = len(node_of) + 1
        self._jac_basis = np.zeros((K, n_b))
        self._hess_basis = np.zeros((K, n_b, n_b))
        for k, i in enumerate(node_of):
            self._jac_basis[k + 1] = out[i].g
            self._hess_basis[k + 1] = out[i].h

        # d / d(foreign curve node DFs) and the mixed second derivative: the discount ratio only, re-evaluated by
        # log-linear interpolation of the foreign nodes at the ACT/365 payment times (xccy_curve.py:637-668)
        fc = self._foreign_curve
        f_times = np.asarray(fc._times, dtype=np.float64)
        f_dfs = np.asarray(fc._dfs, dtype=np.float64)
        n_f = f_dfs.shape[0]
        n = n_b + n_f
        log_nodes = []
        for m in range(n_f):
            g = np.zeros(n); g[n_b + m] = 1.0 / f_dfs[m]
            h = np.zeros((n, n)); h[n_b + m, n_b + m] = -1.0 / (f_dfs[m] * f_dfs[m])
            log_nodes.append(Jet(float(np.log(f_dfs[m])), g, h))

        def df_at(t):
            if t <= f_times[0]:
                return log_nodes[0].exp()
            if t >= f_times[-1]:
                return log_nodes[-1].exp()
            i = int(np.clip(np.searchsorted(f_times, t, side="right"), 1, n_f - 1))
            w = (t - f_times[i - 1]) / (f_times[i] - f_times[i - 1])
            return (log_nodes[i - 1] + (log_nodes[i] - log_nodes[i - 1]) * w).exp()

        jb = [Jet.variable(b, s, n) for s, b in enumerate(self.basis_spreads)]
        out = self._bootstrap(points, pv_dom, jb, [df_at(p["time"]) for p in points], n)
        self._jac_foreign_curve_dfs = np.zeros((K, n_f))
        self._mixed_hess_foreign_basis = np.zeros((K, n_b, n_f))
        for k, i in enumerate(node_of):
            self._jac_foreign_curve_dfs[k + 1] = out[i].g[n_b:]
            self._mixed_hess_foreign_basis[k + 1] = out[i].h[:n_b, n_b:]
        return self._times, self._dfs

    # ------------------------------------------------------------------ queries
    def df(self, dt, day_count=None):
        """Always ACT/365F, whatever ``day_count`` says (xccy_curve.py:1210-1234)."""
        times = times_from_dates(dt, self._value_dt, DayCountTypes.ACT_365F)
        dfs = self._df(times)
        return dfs if isinstance(dfs, float) else np.array(dfs)

    def par_residuals(self):
        """``(PV_dom + spot_fx * PV_foreign) / N_dom`` of every calibration swap off the finished curve - the
        par condition the bootstrap solves."""
        out = []
        for swap in self._used_swaps:
            dom = swap._domestic_leg.value(self._value_dt, self._domestic_curve, self._domestic_curve)
            frn = swap._foreign_leg.value(self._value_dt, self, self._foreign_curve)
            out.append((dom + self._spot_fx * frn) / swap._domestic_notional)
        return out

    def _check_refits(self, swap_tol):
        """The reference's check (xccy_curve.py:1238-1272): every calibration swap valued through
        `XccyBasisSwap.value` with this curve and `spot_fx`.  Note that `value` converts with ``/ spot_fx`` while
        the bootstrap's par condition multiplies by it, so this only passes when the two agree (spot_fx = 1) -
        `Model.build_xccy_curve` never asks for it."""
        for swap in self._used_swaps:
            v = swap.value(self._value_dt, self._domestic_curve, self._foreign_curve, xccy_discount_curve=self,
                           spot_fx=self._spot_fx) / swap._domestic_notional
            if abs(v) > swap_tol:
                raise LibError(f"XCCY swap with maturity {swap._maturity_dt} not repriced. Difference is {abs(v)}")

    def __repr__(self):
        return (f"XccyCurve(value_dt={self._value_dt}, pillars={len(self._used_swaps)}, nodes={len(self._times)}, "
                f"spot_fx={self._spot_fx})")
