"""Third, independent evaluation of the OIS path in 60-digit arithmetic.  TEST INFRASTRUCTURE ONLY.

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
