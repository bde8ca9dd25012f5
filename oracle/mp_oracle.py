"""Third, independent evaluation of the OIS path (and, below, of the cross-currency path) in 60-digit arithmetic.
TEST INFRASTRUCTURE ONLY - imported by tests/ alone.

Purpose: `oracle/cavour_oracle.py` (torch.func autodiff), `oracle/port.c` (analytic partials in knot-DF space) and the
HIP kernels (log-space closed forms) all DIFFERENTIATE something.  A shared mistake on the differentiation side - a
wrong chain rule, a dropped cross term - could make the three agree with each other and still be wrong.  This module
differentiates nothing: the PV is restated as a plain function of the par-rate vector in `mpmath` (60 significant
digits) and its first and second derivatives w.r.t. the par rates are taken by central differences with a step of
1e-20 - at 60 digits the truncation error (~h^2) and the cancellation error (~1e-60 / h^2) are both far below 1e-15,
so the differences ARE the derivatives for every purpose of a float64 comparison.

What is restated (value level only), all paths relative to /root/reference:
  bootstrap of the knot DFs      cavour/market/position/engine.py:2337-2354  (scan body; the knot grid itself - which
                                 knots exist, their order and `prev_idx` - is index logic, taken from
                                 `cavour_oracle.expand_points`)
  simple_interpolate             cavour/market/curves/interpolator_ad.py:186-249 (snap within 1e-10 to the first
                                 nearest knot, else interpolate at t + 1e-12; jnp.interp = searchsorted(right),
                                 clip to [1, K-1], flat outside)
  fixed / float leg PV           cavour/market/position/engine.py:2414-2448, 2639-2728
  units                          delta x 1e-4, gamma x 1e-8 (engine.py:2554, 2566)

The discrete decisions (bracketing knots, snap or not, masks) depend on times only, never on the rates, so they are
made once in float64 exactly as the reference makes them; everything that depends on the rates runs in mpmath.
"""
from __future__ import annotations

import numpy as np
from mpmath import mp, mpf

from . import cavour_oracle as O

mp.dps = 60
_STEP = mpf(10) ** -20


def _lookup_plan(times, t, method):
    """Float64 side of `simple_interpolate` for one time: ('snap', k) or ('interp', i_lo, i_hi, w) or
    ('flat', k) - the same decisions as `cavour_oracle.simple_interpolate`, made on the knot times alone."""
    x = np.asarray(times, dtype=np.float64)
    K = len(x)
    dist = np.abs(t - x)
    k = int(np.argmin(dist))
    if dist[k] < 1e-10:
        return ("snap", k)
    tau = t + 1e-12
    if tau < x[0]:
        return ("flat", 0)
    if tau > x[-1]:
        return ("flat", K - 1)
    i = int(np.clip(np.searchsorted(x, tau, side="right"), 1, K - 1))
    dx = x[i] - x[i - 1]
    if abs(dx) <= np.spacing(np.finfo(np.float64).eps):
        return ("interp", i - 1, i, 0.0)
    return ("interp", i - 1, i, (tau - x[i - 1]) / dx)


def _df(plan, t, x, d, method):
    """Discount factor for one planned lookup; ``d`` are mpf knot DFs."""
    kind = plan[0]
    if kind == "snap":
        return d[plan[1]]
    tt = mpf(float(t))
    if kind == "flat":
        k = plan[1]
        if method == O.LINEAR_ZERO_RATES:
            return mp.exp(-(-mp.log(d[k]) / mpf(max(float(x[k]), 1e-15))) * tt)
        return d[k]                                   # FLAT_FWD keeps the last log-DF, LINEAR_FWD the last DF
    _, a, b, w = plan
    w = mpf(float(w))
    if method == O.LINEAR_ZERO_RATES:
        za = -mp.log(d[a]) / mpf(max(float(x[a]), 1e-15))
        zb = -mp.log(d[b]) / mpf(max(float(x[b]), 1e-15))
        return mp.exp(-(za + w * (zb - za)) * tt)
    if method == O.FLAT_FWD_RATES:
        la, lb = -mp.log(d[a]), -mp.log(d[b])
        return mp.exp(-(la + w * (lb - la)))
    if method == O.LINEAR_FWD_RATES:
        return d[a] + w * (d[b] - d[a])
    raise ValueError("Invalid interpolation scheme.")


class MpTrade:
    """PV of one OIS as a function of the par rates, in mpmath."""

    def __init__(self, swap_rates, swap_times, year_fracs, method, fixed, floating):
        self.times, self.acc, self.rate_idx, self.prev_idx, _ = O.expand_points(swap_rates, swap_times, year_fracs)
        self.rates0 = [mpf(float(r)) for r in swap_rates]
        self.method = method
        self.fixed, self.floating = fixed, floating
        t = self.times
        self.plan0 = _lookup_plan(t, 0.0, method)
        self.plan_fix = [_lookup_plan(t, float(u), method) for u in fixed["payment_times"]]
        self.plan_tp = [_lookup_plan(t, float(u), method) for u in floating["payment_times"]]
        self.plan_ts = [_lookup_plan(t, float(u), method) for u in floating["start_times"]]
        self.plan_te = [_lookup_plan(t, float(u), method) for u in floating["end_times"]]

    def knot_dfs(self, rates):
        K = len(self.acc)
        pv01 = [mpf(0)] * K
        dfs = [mpf(0)] * K
        for i in range(K):
            r = rates[int(self.rate_idx[i])]
            a = mpf(float(self.acc[i]))
            if self.prev_idx[i] < 0:
                prev = mpf(0)
                d = 1 / (1 + r * a)
            else:
                prev = pv01[int(self.prev_idx[i])]
                d = (1 - r * prev) / (1 + r * a)
            pv01[i] = prev + a * d
            dfs[i] = d
        return dfs

    def pv(self, rates):
        d = self.knot_dfs(rates)
        x, m = self.times, self.method
        fx, fl = self.fixed, self.floating
        d0 = _df(self.plan0, 0.0, x, d, m)
        total = mpf(0)
        # fixed leg: payments strictly after the value time (engine.py:2432)
        acc = mpf(0)
        for j, tp in enumerate(fx["payment_times"]):
            if tp > 0.0:
                acc += mpf(float(fx["payments"][j])) * _df(self.plan_fix[j], tp, x, d, m) / d0
        if len(fx["payment_times"]) and fx["payment_times"][-1] > 0.0 and fx.get("principal", 0.0) != 0.0:
            acc += mpf(float(fx["principal"])) * _df(self.plan_fix[-1], fx["payment_times"][-1], x, d, m) / d0
        total += mpf(float(fx["leg_sign"])) * acc
        # float leg: payments at or after the value time (engine.py:2700)
        acc = mpf(0)
        N, s = mpf(float(fl["notional"])), mpf(float(fl["spread"]))
        for j, tp in enumerate(fl["payment_times"]):
            al = float(fl["pay_alphas"][j])
            if al > 0:
                fwd = (_df(self.plan_ts[j], fl["start_times"][j], x, d, m)
                       / _df(self.plan_te[j], fl["end_times"][j], x, d, m) - 1) / mpf(al)
            else:
                fwd = mpf(0)
            if tp >= 0.0:
                acc += (fwd + s) * mpf(al) * N * _df(self.plan_tp[j], tp, x, d, m) / d0
        total += mpf(float(fl["leg_sign"])) * acc
        return total

    def _bumped(self, shifts):
        r = list(self.rates0)
        for p, k in shifts:
            r[p] = r[p] + k * _STEP
        return self.pv(r)

    def value(self):
        return float(self.pv(self.rates0))

    def delta(self, pillars=None):
        """[P] per bp; ``pillars``: which entries to compute (others are returned as NaN)."""
        P = len(self.rates0)
        out = np.full(P, np.nan)
        for p in (range(P) if pillars is None else pillars):
            out[p] = float((self._bumped([(p, 1)]) - self._bumped([(p, -1)])) / (2 * _STEP)) * 1e-4
        return out

    def gamma(self, pairs):
        """{(p, q): d2PV/dr_p dr_q per bp^2} for the requested pairs."""
        v0 = self.pv(self.rates0)
        out = {}
        for p, q in pairs:
            if p == q:
                g = (self._bumped([(p, 1)]) - 2 * v0 + self._bumped([(p, -1)])) / _STEP ** 2
            else:
                g = (self._bumped([(p, 1), (q, 1)]) - self._bumped([(p, 1), (q, -1)])
                     - self._bumped([(p, -1), (q, 1)]) + self._bumped([(p, -1), (q, -1)])) / (4 * _STEP ** 2)
            out[(p, q)] = float(g) * 1e-8
        return out


# --------------------------------------------------------------------------------------------------------------
# Cross-currency basis swaps: the same idea for Engine._compute_xccy (cavour/market/position/engine.py:1411-1988) and
# the XCCY curve's bootstrap (cavour/trades/rates/xccy_curve.py:954-1206, `_run_jax_bootstrap_impl`).  The PV is a
# plain mpmath function of (domestic par rates, foreign par rates, basis spreads): the two OIS engine grids are
# bootstrapped from their par rates, the XCCY knot DFs come from the scan over the calibration swaps' payment points
# (everything the scan takes from the foreign curve's OWN nodes - discount factors at the payment dates, forwards -
# is held at its float value, as the reference's `from_basis` does), the legs are `_float_leg_jax` with a separate
# index curve and notional exchanges.  Ladders by central differences; nothing is differentiated symbolically.
# --------------------------------------------------------------------------------------------------------------
def _interp_clamped(x, xp, fp):
    """jnp.interp (linear, flat outside) of mpf ordinates ``fp`` at a float abscissa."""
    xp = np.asarray(xp, dtype=np.float64)
    if x < xp[0]:
        return fp[0]
    if x > xp[-1]:
        return fp[-1]
    i = int(np.clip(np.searchsorted(xp, x, side="right"), 1, len(xp) - 1))
    w = mpf(float((x - xp[i - 1]) / (xp[i] - xp[i - 1])))
    return fp[i - 1] + w * (fp[i] - fp[i - 1])


class MpXccy:
    """PV of one basis swap as a function of the three quote vectors, in mpmath."""

    def __init__(self, swap, value_dt, dom_curve, for_curve, xccy_curve, calib_swaps, times_from_dates):
        from . import xccy_oracle as XO
        self.tfd = times_from_dates
        self.dom = (dom_curve._interp_type.value,) + O.expand_points(dom_curve.swap_rates, dom_curve.swap_times, dom_curve.year_fracs)[:4]
        self.frn = (for_curve._interp_type.value,) + O.expand_points(for_curve.swap_rates, for_curve.swap_times, for_curve.year_fracs)[:4]
        self.r_dom = [mpf(float(r)) for r in dom_curve.swap_rates]
        self.r_for = [mpf(float(r)) for r in for_curve.swap_rates]
        self.spreads = [mpf(float(s._foreign_spread)) for s in calib_swaps]
        self.spot = mpf(float(xccy_curve._spot_fx))
        self.x_method = xccy_curve._interp_type.value
        # the scan's fixed inputs (floats): payment points, domestic leg values, foreign own-node DFs
        self.pts = XO.payment_points(value_dt, calib_swaps, for_curve, times_from_dates)
        self.pv_dom = [float(s._domestic_leg.value(value_dt, dom_curve, dom_curve)) for s in calib_swaps]
        self.f_times = np.asarray(for_curve._times, dtype=np.float64)
        self.f_logdfs = [mp.log(mpf(float(d))) for d in for_curve._dfs]
        nodes, seen = [], set()
        for i, p in enumerate(self.pts):
            if p["at_val"] or round(p["time"], 4) in seen:
                continue
            seen.add(round(p["time"], 4)); nodes.append(i)
        self.nodes = nodes
        self.x_times = np.array([0.0] + [self.pts[i]["time"] for i in nodes])
        # the trade
        dl, fl = swap._domestic_leg, swap._foreign_leg
        receive = type(dl._leg_type).RECEIVE
        T = lambda dts, dc: [float(times_from_dates(d, value_dt, dc)) for d in dts]
        ddc, fdc, xdc = dl._dc_type, fl._dc_type, xccy_curve._dc_type
        self.dom_leg = dict(tp=T(dl._payment_dts, ddc), ts=T(dl._start_accrued_dts, ddc), te=T(dl._end_accrued_dts, ddc),
                            al=[float(a) for a in dl._year_fracs], spread=float(dl._spread), N=float(dl._notional),
                            sign=1.0 if dl._leg_type == receive else -1.0, exch=bool(dl._notional_exchange),
                            t_eff=float(times_from_dates(swap._effective_dt, value_dt, ddc)),
                            t_mat=float(times_from_dates(swap._maturity_dt, value_dt, ddc)))
        self.for_leg = dict(tp=T(fl._payment_dts, xdc), ts=T(fl._start_accrued_dts, fdc), te=T(fl._end_accrued_dts, fdc),
                            al=[float(a) for a in fl._year_fracs], spread=float(fl._spread), N=float(fl._notional),
                            sign=1.0 if fl._leg_type == receive else -1.0, exch=bool(fl._notional_exchange),
                            t_eff=float(times_from_dates(swap._effective_dt, value_dt, xdc)),
                            t_mat=float(times_from_dates(swap._maturity_dt, value_dt, xdc)))

    # ---- curves
    @staticmethod
    def _grid_dfs(grid, rates):
        _, times, acc, rate_idx, prev_idx = grid
        K = len(acc)
        pv01, dfs = [mpf(0)] * K, [mpf(0)] * K
        for i in range(K):
            r, a = rates[int(rate_idx[i])], mpf(float(acc[i]))
            prev = mpf(0) if prev_idx[i] < 0 else pv01[int(prev_idx[i])]
            d = (1 - r * prev) / (1 + r * a) if prev_idx[i] >= 0 else 1 / (1 + r * a)
            pv01[i] = prev + a * d
            dfs[i] = d
        return dfs

    def xccy_dfs(self, spreads):
        """The scan (xccy_curve.py:954-1206) at value level; returns the knot DFs [1, nodes...]."""
        pts, out, pv_c, cf_c, prev = self.pts, [], [], [], -1
        for i, p in enumerate(pts):
            basis = spreads[p["swap"]]
            df_s = mp.exp(_interp_clamped(p["ts"], self.f_times, self.f_logdfs))
            df_e = mp.exp(_interp_clamped(p["te"], self.f_times, self.f_logdfs))
            fwd = (df_s / df_e - 1) / mpf(max(p["yf"], 1e-10)) if p["yf"] > 1e-10 else mpf(0)
            interest = fwd * mpf(p["yf"]) * mpf(p["N"]) + (mpf(p["N"]) if p["last"] else 0)
            base = (mpf(p["N"]) if p["last"] else -mpf(p["N"])) if p["exch"] else interest
            cashflow = base + basis * mpf(p["sens"])
            df_o = mpf(float(p["df_ois"]))
            if prev < 0:
                df_mid = df_o * mp.exp(-basis * mpf(p["time"]))
            else:
                df_mid = out[prev] * (df_o / mpf(float(pts[prev]["df_ois"]))) * mp.exp(-basis * mpf(p["time"] - pts[prev]["time"]))
            known = (not p["is_mat"]) and (not p["at_val"])
            total = cashflow * df_mid if known else (cashflow if p["at_val"] else mpf(0))
            cf_here = cashflow if p["is_mat"] else mpf(0)
            same = [j for j in range(i) if pts[j]["swap"] == p["swap"]]
            pv_known = sum((pv_c[j] for j in same), mpf(0)) + total
            cf_mat = sum((cf_c[j] for j in same), mpf(0)) + cf_here
            pv_c.append(total); cf_c.append(cf_here)
            numerator = -(mpf(self.pv_dom[p["swap"]]) + self.spot * (-pv_known))
            denominator = self.spot * (-cf_mat)
            out.append(numerator / denominator if (p["is_mat"] and abs(denominator) > 1e-12) else df_mid)
            if not p["at_val"]:
                prev = i
        return [mpf(1)] + [out[i] for i in self.nodes]

    # ---- legs
    @staticmethod
    def _lookup(times, dfs, method, t):
        return _df(_lookup_plan(times, float(t), method), float(t), times, dfs, method)

    def _leg(self, leg, disc, idx):
        """`_float_leg_jax` (engine.py:2639-2728): ``disc`` / ``idx`` = (times, dfs, method) of the two curves."""
        d0 = self._lookup(*disc, 0.0)
        pv = mpf(0)
        N, s = mpf(leg["N"]), mpf(leg["spread"])
        for j, tp in enumerate(leg["tp"]):
            al = leg["al"][j]
            fwd = (self._lookup(*idx, leg["ts"][j]) / self._lookup(*idx, leg["te"][j]) - 1) / mpf(al) if al > 0 else mpf(0)
            if tp >= 0.0:
                pv += (fwd + s) * mpf(al) * N * self._lookup(*disc, tp) / d0
        if leg["exch"]:
            if leg["t_eff"] >= 0.0:
                pv -= N * self._lookup(*disc, leg["t_eff"]) / d0
            if leg["t_mat"] >= 0.0:
                pv += N * self._lookup(*disc, leg["t_mat"]) / d0
        return mpf(leg["sign"]) * pv

    def pv(self, r_dom, r_for, spreads):
        dom = (self.dom[1], self._grid_dfs(self.dom, r_dom), self.dom[0])
        frn = (self.frn[1], self._grid_dfs(self.frn, r_for), self.frn[0])
        xcy = (self.x_times, self.xccy_dfs(spreads), self.x_method)
        return self._leg(self.dom_leg, dom, dom) + self._leg(self.for_leg, xcy, frn) / self.spot

    # ---- ladders by central differences
    def _args(self, which, shifts):
        a = {"dom": list(self.r_dom), "for": list(self.r_for), "basis": list(self.spreads)}
        for w, p, k in shifts:
            a[w][p] = a[w][p] + k * _STEP
        return a["dom"], a["for"], a["basis"]

    def value(self):
        return float(self.pv(self.r_dom, self.r_for, self.spreads))

    def delta(self, which, pillars):
        """{pillar: dPV/dq per bp} for the quote vector ``which`` in {'dom', 'for', 'basis'}."""
        out = {}
        for p in pillars:
            up, dn = self.pv(*self._args(which, [(which, p, 1)])), self.pv(*self._args(which, [(which, p, -1)]))
            out[p] = float((up - dn) / (2 * _STEP)) * 1e-4
        return out

    def second(self, a, p, b, q):
        """d2 PV / d a_p d b_q per bp^2 (a, b in {'dom', 'for', 'basis'}; the same vector and pillar: the pure second)."""
        if (a, p) == (b, q):
            v0 = self.pv(self.r_dom, self.r_for, self.spreads)
            g = (self.pv(*self._args(a, [(a, p, 1)])) - 2 * v0 + self.pv(*self._args(a, [(a, p, -1)]))) / _STEP ** 2
        else:
            f = lambda s1, s2: self.pv(*self._args(a, [(a, p, s1), (b, q, s2)]))
            g = (f(1, 1) - f(1, -1) - f(-1, 1) + f(-1, -1)) / (4 * _STEP ** 2)
        return float(g) * 1e-8
