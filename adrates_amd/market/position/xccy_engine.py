"""Cross-currency basis swaps through the OIS kernels.

Mirrors `Engine._compute_xccy` (cavour/market/position/engine.py:1411-1988) for `XccyBasisSwap`: PV in domestic
currency, three delta ladders and three gamma matrices - domestic OIS pillars, foreign OIS pillars, basis
pillars.  The reference differentiates two `_float_leg_jax` calls (domestic leg on its own curve; foreign leg
discounted on the XCCY curve with forwards off the foreign OIS curve) w.r.t. three knot-DF vectors and chains
them with the curves' Jacobians / Hessians.  Every piece is a sum of terms ``c * exp(sum beta * ln d_k)`` over
ONE curve's knots once the other curve is held fixed - which is exactly how the reference defines its ladders
(the XCCY curve is fixed when the foreign OIS rates move, :1702-1712) - so the existing kernels do all of it:

1. domestic leg: a float leg plus two "fixed" flows (-N at the effective time, +N at maturity) on the
   domestic curve's tables;
2. foreign-rate ladders: one single-coupon trade per foreign coupon on the foreign OIS tables, notional
   ``N * D_x(tp_j)``, paid at time 0 (where D = 1), accrual from ``ts_j`` to ``te_j``: its Greeks are those of
   ``N D_x(tp_j) D_f(ts_j) / D_f(te_j)``;
3. foreign PV and basis ladders: fixed flows ``N (fwd_j + s) alpha_j`` at ``tp_j`` plus the two exchanges, on
   tables uploaded from the XCCY curve's ``(_times, _dfs, _jac_basis, _hess_basis)``.

Flows dated exactly at the value time follow the reference's masks: coupons and exchanges count (``>=``), and
since their discount factor is 1 they are added to the PV on the host (the kernels' fixed-flow mask is ``>``).

The reference's cross-gamma block (:1895-1960) is not reproduced: it contracts a tensor indexed by the foreign
curve's own nodes with the Jacobian of the engine's knot grid, two different sizes.
"""
from __future__ import annotations

import numpy as np

from ... import _native
from ...requests.results import AnalyticsResult, Delta, Gamma, Risk, Valuation
from ...trades.compiler import TradeBatch
from ...utils.error import LibError
from ...utils.global_types import CurveTypes, InterpTypes, RequestTypes, SwapTypes
from ...utils.helpers import times_from_dates, to_tenor


def knot_df(times, dfs, t, method: int):
    """`InterpolatorAd.simple_interpolate` (cavour/market/curves/interpolator_ad.py:186-249) for values:
    snap to a knot within 1e-10 (first knot on ties), else interpolate at t + 1e-12, flat outside the grid."""
    x = np.asarray(times, dtype=np.float64)
    d = np.asarray(dfs, dtype=np.float64)
    tt = np.atleast_1d(np.asarray(t, dtype=np.float64))
    dist = np.abs(tt[:, None] - x[None, :])
    k = np.argmin(dist, axis=1)
    snapped = dist[np.arange(tt.size), k] < 1e-10
    tau = tt + 1e-12
    if method == InterpTypes.LINEAR_ZERO_RATES.value:
        out = np.exp(-np.interp(tau, x, -np.log(d) / np.maximum(x, 1e-15)) * tt)
    elif method == InterpTypes.FLAT_FWD_RATES.value:
        out = np.exp(-np.interp(tau, x, -np.log(d)))
    else:
        raise LibError("Invalid interpolation scheme.")
    out = np.where(snapped, d[k], out)
    return float(out[0]) if np.ndim(t) == 0 else out


def _times(dts, value_dt, dc):
    return np.array([times_from_dates(d, value_dt, dc) for d in dts], dtype=np.float64)


def _batch(n, fix_tp, fix_pay, fix_counts, flt, flt_counts, notional, spread, fix_sign, flt_sign):
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    off = lambda c: np.concatenate(([0], np.cumsum(c))).astype(np.int64)
    z = np.zeros(0)
    tp, ts, te, al = flt if flt is not None else (z, z, z, z)
    return TradeBatch(off(fix_counts), off(flt_counts), f64(fix_tp), f64(fix_pay), f64(tp), f64(ts), f64(te), f64(al),
                      f64(notional), f64(spread), f64(fix_sign), f64(flt_sign))


def compute_xccy(engine, derivative, reqs):
    model = engine.model
    dom_model = getattr(model.curves, derivative._domestic_floating_index.name)
    for_model = getattr(model.curves, derivative._foreign_floating_index.name)
    name = f"{derivative._foreign_currency.name}_{derivative._domestic_currency.name}_BASIS"
    try:
        xccy = getattr(model.curves, name)
    except AttributeError:
        raise LibError(f"XCCY curve {name} not found in model.")
    if getattr(xccy, "_jac_basis", None) is None:
        raise LibError("the XCCY curve carries no basis Jacobian (build it with use_ad=True)")
    spot = xccy._spot_fx
    value_dt = model.value_dt
    want_delta = bool(reqs & {RequestTypes.DELTA, RequestTypes.GAMMA})
    want_gamma = RequestTypes.GAMMA in reqs

    dom_cur, for_cur = engine._device_curve(dom_model), engine._device_curve(for_model)
    ctx = dom_cur["ctx"]
    x_dev = getattr(xccy, "_adr_device_curve", None)
    if x_dev is None:
        x_dev = _native.DeviceCurve(ctx, xccy._interp_type.value, np.asarray(xccy._times), np.asarray(xccy._dfs),
                                    np.asarray(xccy._jac_basis), np.asarray(xccy._hess_basis))
        xccy._adr_device_curve = x_dev

    dl, fl = derivative._domestic_leg, derivative._foreign_leg
    sign = lambda leg: 1.0 if leg._leg_type == SwapTypes.RECEIVE else -1.0
    kw = dict(want_value=True, want_delta=want_delta, want_gamma=want_gamma)
    pv_const = 0.0          # flows dated exactly at the value time (discount factor 1, no sensitivity)

    # ---- 1. domestic leg on the domestic curve
    ddc = dl._dc_type
    t_eff, t_mat = times_from_dates(derivative._effective_dt, value_dt, ddc), times_from_dates(derivative._maturity_dt, value_dt, ddc)
    fix_tp, fix_pay = [], []
    if dl._notional_exchange:
        for t, amount in ((t_eff, -dl._notional), (t_mat, dl._notional)):
            if t > 0.0:
                fix_tp.append(t); fix_pay.append(amount)
            elif t == 0.0:
                pv_const += sign(dl) * amount
    b = _batch(1, fix_tp, fix_pay, [len(fix_tp)],
               (_times(dl._payment_dts, value_dt, ddc), _times(dl._start_accrued_dts, value_dt, ddc),
                _times(dl._end_accrued_dts, value_dt, ddc), np.asarray(dl._year_fracs)), [len(dl._payment_dts)],
               [dl._notional], [dl._spread], [sign(dl)], [sign(dl)])
    dom = _price(ctx, dom_cur["dev"], b, kw)

    # ---- 2. foreign coupons: forwards off the foreign OIS grid, discount factors off the XCCY knots
    fdc, xdc = fl._dc_type, xccy._dc_type
    tp_x = _times(fl._payment_dts, value_dt, xdc)
    ts_f, te_f = _times(fl._start_accrued_dts, value_dt, fdc), _times(fl._end_accrued_dts, value_dt, fdc)
    alpha = np.asarray(fl._year_fracs, dtype=np.float64)
    live = tp_x >= 0.0
    x_times, x_dfs, x_method = np.asarray(xccy._times), np.asarray(xccy._dfs), xccy._interp_type.value
    f_host = for_cur["host"]
    f_method = for_model._interp_type.value
    c = knot_df(x_times, x_dfs, tp_x, x_method) / knot_df(x_times, x_dfs, 0.0, x_method)   # relative to the value time
    fwd = np.where(alpha > 0, (knot_df(f_host.times, f_host.dfs, ts_f, f_method)
                               / knot_df(f_host.times, f_host.dfs, te_f, f_method) - 1.0) / np.where(alpha > 0, alpha, 1.0), 0.0)
    delta_for = gamma_for = None
    if want_delta:
        idx = np.flatnonzero(live & (alpha > 0))
        m = idx.size
        zeros = np.zeros(m)
        b = _batch(m, [], [], [0] * m, (zeros, ts_f[idx], te_f[idx], alpha[idx]), [1] * m,
                   fl._notional * c[idx], zeros, np.ones(m), np.full(m, sign(fl)))
        out = _price(ctx, for_cur["dev"], b, dict(want_value=False, want_delta=True, want_gamma=want_gamma))
        delta_for = out["delta"].sum(0) / spot
        gamma_for = out["gamma"].sum(0) / spot if want_gamma else None

    # ---- 3. foreign PV and basis ladders: the coupons as fixed flows on the XCCY tables
    amounts = (fwd + fl._spread) * alpha * fl._notional
    fix_tp, fix_pay = [], []
    for t, amount in zip(tp_x[live], amounts[live]):
        if t > 0.0:
            fix_tp.append(t); fix_pay.append(amount)
        else:
            pv_const += sign(fl) * amount / spot
    if fl._notional_exchange:
        for t, amount in ((times_from_dates(derivative._effective_dt, value_dt, xdc), -fl._notional),
                          (times_from_dates(derivative._maturity_dt, value_dt, xdc), fl._notional)):
            if t > 0.0:
                fix_tp.append(t); fix_pay.append(amount)
            elif t == 0.0:
                pv_const += sign(fl) * amount / spot
    b = _batch(1, fix_tp, fix_pay, [len(fix_tp)], None, [0], [fl._notional], [0.0], [sign(fl)], [sign(fl)])
    frn = _price(ctx, x_dev, b, kw)

    ccy = derivative._domestic_currency
    value = delta = gamma = None
    if RequestTypes.VALUE in reqs:
        value = Valuation(amount=float(dom["pv"][0] + frn["pv"][0] / spot + pv_const), currency=ccy)
    dom_tenors, for_tenors = to_tenor(list(dom_model.swap_times)), to_tenor(list(for_model.swap_times))
    basis_tenors = to_tenor(list(xccy.swap_times))
    if RequestTypes.DELTA in reqs:
        delta = Risk([Delta(np.array(dom["delta"][0]), dom_tenors, ccy, derivative._domestic_floating_index),
                      Delta(np.array(delta_for), for_tenors, ccy, derivative._foreign_floating_index),
                      Delta(np.array(frn["delta"][0] / spot), basis_tenors, ccy, CurveTypes.USD_GBP_BASIS)])
    if want_gamma:
        gamma = Risk([Gamma(np.array(dom["gamma"][0]), dom_tenors, ccy, derivative._domestic_floating_index),
                      Gamma(np.array(gamma_for), for_tenors, ccy, derivative._foreign_floating_index),
                      Gamma(np.array(frn["gamma"][0] / spot), basis_tenors, ccy, CurveTypes.USD_GBP_BASIS)])
    return AnalyticsResult(value=value, risk=delta, gamma=gamma)


def _price(ctx, dev_curve, batch, kw):
    trades = _native.DeviceTrades(ctx, batch)
    try:
        return _native.price(ctx, dev_curve, trades, **kw)
    finally:
        trades.close()
