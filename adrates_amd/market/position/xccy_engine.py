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
2. foreign-rate ladders: the foreign coupons on the foreign OIS tables with notional ``N * D_x(tp_j)``, paid at
   time 0 (where D = 1), accrual from ``ts_j`` to ``te_j``: the Greeks of ``sum_j N D_x(tp_j) D_f(ts_j) / D_f(te_j)``;
3. foreign PV and basis ladders: fixed flows ``N (fwd_j + s) alpha_j`` at ``tp_j`` plus the two exchanges, on
   tables uploaded from the XCCY curve's ``(_times, _dfs, _jac_basis, _hess_basis)``.

Piece 2 is one trade per swap whose coupons carry a per-coupon notional multiplier
(`adr_trades_upload_weighted`), so a book of swaps costs three launches.

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
    elif method == InterpTypes.LINEAR_FWD_RATES.value:
        out = np.interp(tau, x, d)
    else:
        raise LibError("Invalid interpolation scheme.")
    out = np.where(snapped, d[k], out)
    return float(out[0]) if np.ndim(t) == 0 else out


def _times(dts, value_dt, dc):
    return np.array([times_from_dates(d, value_dt, dc) for d in dts], dtype=np.float64)


class _Csr:
    """Flat cash-flow arrays of a batch under construction."""

    def __init__(self, fields):
        self.cols = {f: [] for f in fields}
        self.off = [0]

    def add(self, **cols):
        n = None
        for k, v in cols.items():
            self.cols[k].append(np.asarray(v, dtype=np.float64).reshape(-1))
            n = self.cols[k][-1].size
        self.off.append(self.off[-1] + (n or 0))

    def col(self, k):
        return np.concatenate(self.cols[k]) if self.cols[k] else np.zeros(0)

    def offsets(self):
        return np.asarray(self.off, dtype=np.int64)


def _sign(leg):
    return 1.0 if leg._leg_type == SwapTypes.RECEIVE else -1.0


def compile_xccy(swaps, value_dt, xccy, for_times, for_dfs, for_method):
    """The three trade batches of the assembly above, plus the per-swap PV of the flows dated at the value time.

    ``for_times, for_dfs``: the foreign OIS curve's engine grid (the reference projects the foreign forwards off
    `_cached_curve`'s knots, engine.py:1640-1668).  Returns ``(domestic, foreign_rates, foreign_flows, pv_const,
    spot)``; ``foreign_rates`` carries `flt_weight`."""
    spot = xccy._spot_fx
    x_times, x_dfs, x_method = np.asarray(xccy._times), np.asarray(xccy._dfs), xccy._interp_type.value
    xdc = xccy._dc_type
    n = len(swaps)
    pv_const = np.zeros(n)
    dom_fix, dom_flt = _Csr(("tp", "pay")), _Csr(("tp", "ts", "te", "al"))
    for_flt = _Csr(("tp_x", "ts", "te", "al"))
    exch = _Csr(("tp", "pay"))
    for i, swap in enumerate(swaps):
        dl, fl = swap._domestic_leg, swap._foreign_leg
        ddc, fdc = dl._dc_type, fl._dc_type
        tp, pay = [], []
        if dl._notional_exchange:
            for t, amount in ((times_from_dates(swap._effective_dt, value_dt, ddc), -dl._notional),
                              (times_from_dates(swap._maturity_dt, value_dt, ddc), dl._notional)):
                if t > 0.0:
                    tp.append(t); pay.append(amount)
                elif t == 0.0:
                    pv_const[i] += _sign(dl) * amount
        dom_fix.add(tp=tp, pay=pay)
        dom_flt.add(tp=_times(dl._payment_dts, value_dt, ddc), ts=_times(dl._start_accrued_dts, value_dt, ddc),
                    te=_times(dl._end_accrued_dts, value_dt, ddc), al=dl._year_fracs)
        for_flt.add(tp_x=_times(fl._payment_dts, value_dt, xdc), ts=_times(fl._start_accrued_dts, value_dt, fdc),
                    te=_times(fl._end_accrued_dts, value_dt, fdc), al=fl._year_fracs)
        tp, pay = [], []
        if fl._notional_exchange:
            for t, amount in ((times_from_dates(swap._effective_dt, value_dt, xdc), -fl._notional),
                              (times_from_dates(swap._maturity_dt, value_dt, xdc), fl._notional)):
                if t > 0.0:
                    tp.append(t); pay.append(amount)
                elif t == 0.0:
                    pv_const[i] += _sign(fl) * amount / spot
        exch.add(tp=tp, pay=pay)

    f64 = lambda v: np.array(v, dtype=np.float64)
    dom_n, dom_s = f64([s._domestic_leg._notional for s in swaps]), f64([_sign(s._domestic_leg) for s in swaps])
    for_n, for_s = f64([s._foreign_leg._notional for s in swaps]), f64([_sign(s._foreign_leg) for s in swaps])
    domestic = TradeBatch(dom_fix.offsets(), dom_flt.offsets(), dom_fix.col("tp"), dom_fix.col("pay"),
                          dom_flt.col("tp"), dom_flt.col("ts"), dom_flt.col("te"), dom_flt.col("al"), dom_n,
                          f64([s._domestic_leg._spread for s in swaps]), dom_s, dom_s)

    # foreign coupons, all swaps at once: forwards off the foreign OIS grid, discount factors off the XCCY knots
    off = for_flt.offsets()
    owner = np.repeat(np.arange(n), np.diff(off))
    tp_x, ts, te, al = for_flt.col("tp_x"), for_flt.col("ts"), for_flt.col("te"), for_flt.col("al")
    c = knot_df(x_times, x_dfs, tp_x, x_method) / knot_df(x_times, x_dfs, 0.0, x_method)   # relative to the value time
    accrues = al > 0
    fwd = np.where(accrues, (knot_df(for_times, for_dfs, ts, for_method) / knot_df(for_times, for_dfs, te, for_method)
                             - 1.0) / np.where(accrues, al, 1.0), 0.0)
    live = tp_x >= 0.0
    keep = live & accrues
    kept_off = np.concatenate(([0], np.cumsum(np.bincount(owner[keep], minlength=n)))).astype(np.int64)
    zeros_n, none = np.zeros(n), np.zeros(0)
    foreign_rates = TradeBatch(np.zeros(n + 1, dtype=np.int64), kept_off, none, none, np.zeros(int(keep.sum())),
                               ts[keep], te[keep], al[keep], for_n, zeros_n, for_s, for_s, flt_weight=c[keep])

    spreads = f64([s._foreign_leg._spread for s in swaps])
    amounts = (fwd + spreads[owner]) * al * for_n[owner]
    at_value_time = live & (tp_x == 0.0)
    np.add.at(pv_const, owner[at_value_time], (for_s[owner] * amounts)[at_value_time] / spot)
    later = live & (tp_x > 0.0)
    # per swap: its later coupons, then its exchanges
    cnt = np.bincount(owner[later], minlength=n) + np.diff(exch.offsets())
    fix_off = np.concatenate(([0], np.cumsum(cnt))).astype(np.int64)
    flow_tp, flow_pay = np.empty(int(fix_off[-1])), np.empty(int(fix_off[-1]))
    c_owner = owner[later]                                  # sorted: the coupons are laid out swap by swap
    c_first = np.concatenate(([0], np.cumsum(np.bincount(c_owner, minlength=n))))[:-1]
    coupon_pos = fix_off[:-1][c_owner] + (np.arange(c_owner.size) - c_first[c_owner])     # from the swap's first slot
    flow_tp[coupon_pos], flow_pay[coupon_pos] = tp_x[later], amounts[later]
    e_off = exch.offsets()
    e_owner = np.repeat(np.arange(n), np.diff(e_off))
    exch_pos = fix_off[1:][e_owner] - (e_off[1:][e_owner] - np.arange(e_off[-1]))         # back from its last slot
    flow_tp[exch_pos], flow_pay[exch_pos] = exch.col("tp"), exch.col("pay")
    foreign_flows = TradeBatch(fix_off, np.zeros(n + 1, dtype=np.int64), flow_tp, flow_pay, none, none, none, none,
                               for_n, zeros_n, for_s, for_s)
    return domestic, foreign_rates, foreign_flows, pv_const, spot


def _xccy_device_curve(ctx, xccy):
    """The XCCY curve's tables as an `adr_curve` (cached on the curve object).  The fast kernel's packed layout
    wants an even pillar count (16-byte gamma stores), so an odd basis ladder gets one all-zero pillar appended
    here and dropped again from the results (`_trim`)."""
    times, dfs = np.asarray(xccy._times, dtype=np.float64), np.asarray(xccy._dfs, dtype=np.float64)
    # valid for exactly these knots, this scheme and this context
    key = (times.tobytes(), dfs.tobytes(), xccy._interp_type.value)
    hit = getattr(xccy, "_adr_device_curve", None)
    if hit is not None and hit[0] == key and hit[1] is ctx:
        return hit[2]
    jac = getattr(xccy, "_jac_basis", None)
    hess = getattr(xccy, "_hess_basis", None) if jac is not None else None
    jac = np.zeros((times.size, 2)) if jac is None else np.asarray(jac, dtype=np.float64)
    hess = None if hess is None else np.asarray(hess, dtype=np.float64)
    if jac.shape[1] % 2:
        jac = np.pad(jac, ((0, 0), (0, 1)))
        hess = None if hess is None else np.pad(hess, ((0, 0), (0, 1), (0, 1)))
    dev = _native.DeviceCurve(ctx, xccy._interp_type.value, times, dfs, jac, hess)
    xccy._adr_device_curve = (key, ctx, dev)
    return dev


def _trim(a, kind, P):
    """Drop the padding pillar of `_xccy_device_curve` from a delta ([..., P']) or gamma ([..., P', P']) array."""
    a = np.asarray(a)
    return a[..., :P] if kind == "delta" else a[..., :P, :P]


def _curves(engine, swaps):
    model = engine.model
    first = swaps[0]
    for s in swaps:
        if (s._domestic_floating_index, s._foreign_floating_index, s._domestic_currency, s._foreign_currency) != \
           (first._domestic_floating_index, first._foreign_floating_index, first._domestic_currency, first._foreign_currency):
            raise LibError("a batch of cross-currency swaps must share its currencies and floating indices")
    dom_model = getattr(model.curves, first._domestic_floating_index.name)
    for_model = getattr(model.curves, first._foreign_floating_index.name)
    name = f"{first._foreign_currency.name}_{first._domestic_currency.name}_BASIS"
    try:
        xccy = getattr(model.curves, name)
    except AttributeError:
        raise LibError(f"XCCY curve {name} not found in model.")
    if getattr(xccy, "_jac_basis", None) is None:
        raise LibError("the XCCY curve carries no basis Jacobian (build it with use_ad=True)")
    dom_cur, for_cur = engine._device_curve(dom_model), engine._device_curve(for_model)
    x_dev = _xccy_device_curve(dom_cur["ctx"], xccy)
    return dom_model, for_model, xccy, dom_cur, for_cur, x_dev


def price_xccy_batch(engine, swaps, reqs, per_trade=True, aggregate=False):
    """VALUE / DELTA / GAMMA of a book of cross-currency basis swaps on one currency pair: three launches.

    Returns a dict: ``pv [n]``, ``delta_dom [n, P_d]``, ``delta_for [n, P_f]``, ``delta_basis [n, P_b]`` and the
    three ``gamma_*`` (per request, when ``per_trade``), ``agg_*`` sums over the book (when ``aggregate``);
    everything in domestic currency, per bp / bp^2."""
    reqs = set(reqs)
    swaps = list(swaps)
    if not swaps:
        raise LibError("price_xccy_batch needs at least one swap (the book's currency pair names its curves)")
    dom_model, for_model, xccy, dom_cur, for_cur, x_dev = _curves(engine, swaps)
    ctx = dom_cur["ctx"]
    f_host = for_cur["host"]
    domestic, foreign_rates, foreign_flows, pv_const, spot = compile_xccy(
        swaps, engine.model.value_dt, xccy, f_host.times, f_host.dfs, for_model._interp_type.value)
    want_value = RequestTypes.VALUE in reqs
    want_gamma = RequestTypes.GAMMA in reqs
    want_delta = want_gamma or RequestTypes.DELTA in reqs
    kw = dict(want_delta=want_delta, want_gamma=want_gamma, per_trade=per_trade, aggregate=aggregate)
    dom = _price(ctx, dom_cur["dev"], domestic, dict(kw, want_value=want_value))
    frn = _price(ctx, x_dev, foreign_flows, dict(kw, want_value=want_value))
    rates = _price(ctx, for_cur["dev"], foreign_rates, dict(kw, want_value=False)) if want_delta else {}
    out = {}
    for pre in (("",) if per_trade else ()) + (("agg_",) if aggregate else ()):
        if want_value:
            const = pv_const if pre == "" else float(pv_const.sum())
            out[pre + "pv"] = dom[pre + "pv"] + frn[pre + "pv"] / spot + const
        for kind in (("delta",) if want_delta else ()) + (("gamma",) if want_gamma else ()):
            out[f"{pre}{kind}_dom"] = np.asarray(dom[pre + kind])
            out[f"{pre}{kind}_for"] = np.asarray(rates[pre + kind]) / spot
            out[f"{pre}{kind}_basis"] = _trim(frn[pre + kind], kind, len(xccy.swap_times)) / spot
    out["tenors"] = (to_tenor(list(dom_model.swap_times)), to_tenor(list(for_model.swap_times)),
                     to_tenor(list(xccy.swap_times)))
    return out


def compute_xccy(engine, derivative, reqs):
    if RequestTypes.CASHFLOWS in reqs:
        # the reference's block for this request (engine.py:1970-1986) ends in a NameError (`risk_ccy` is never
        # assigned in _compute_xccy), so there is no behaviour to mirror
        raise NotImplementedError("CASHFLOWS is not available for cross-currency swaps")
    res = price_xccy_batch(engine, [derivative], reqs)
    ccy = derivative._domestic_currency
    curves = (derivative._domestic_floating_index, derivative._foreign_floating_index, CurveTypes.USD_GBP_BASIS)
    value = delta = gamma = None
    if RequestTypes.VALUE in reqs:
        value = Valuation(amount=float(res["pv"][0]), currency=ccy)
    if RequestTypes.DELTA in reqs:
        delta = Risk([Delta(np.array(res[k][0]), t, ccy, c)
                      for k, t, c in zip(("delta_dom", "delta_for", "delta_basis"), res["tenors"], curves)])
    if RequestTypes.GAMMA in reqs:
        gamma = Risk([Gamma(np.array(res[k][0]), t, ccy, c)
                      for k, t, c in zip(("gamma_dom", "gamma_for", "gamma_basis"), res["tenors"], curves)])
    return AnalyticsResult(value=value, risk=delta, gamma=gamma)


def compute_ois_xccy_collateral(engine, derivative, reqs, collateral_ccy):
    """An OIS collateralised in another currency (`Engine._compute_ois_xccy_collateral`, engine.py:217-503):
    both legs discounted on the ``{swap ccy}_{collateral ccy}_XCCY`` curve, forwards off the swap's own OIS curve,
    PV and ladders converted to the collateral currency.  Pieces 2 and 3 of the assembly above with the OIS's
    float leg in the foreign leg's role and its fixed coupons among the flows; every time in the fixed leg's day
    count (:264-283).  Like the reference: no GAMMA (:489-494), CASHFLOWS is an empty table (:496-501)."""
    model = engine.model
    if RequestTypes.GAMMA in reqs:
        raise NotImplementedError("GAMMA not yet supported for OIS with cross-currency collateral. "
                                  "Only VALUE and DELTA are currently implemented.")
    ois_model = getattr(model.curves, derivative._floating_index.name)
    name = f"{derivative._currency.name}_{collateral_ccy.name}_XCCY"
    try:
        xccy = getattr(model.curves, name)
    except AttributeError:
        raise LibError(f"XCCY curve {name} not found in model. Required for cross-currency collateral valuation.")
    spot = xccy._spot_fx
    value_dt = model.value_dt
    fx, fl = derivative._fixed_leg, derivative._float_leg
    dc = fx._dc_type
    x_times, x_dfs, x_method = np.asarray(xccy._times), np.asarray(xccy._dfs), xccy._interp_type.value
    ois_cur = engine._device_curve(ois_model)
    ctx, o_host = ois_cur["ctx"], ois_cur["host"]
    o_method = ois_model._interp_type.value

    tp, ts, te = (_times(d, value_dt, dc) for d in (fl._payment_dts, fl._start_accrued_dts, fl._end_accrued_dts))
    al = np.asarray(fl._year_fracs, dtype=np.float64)
    accrues, live = al > 0, tp >= 0.0
    fwd = np.where(accrues, (knot_df(o_host.times, o_host.dfs, ts, o_method) / knot_df(o_host.times, o_host.dfs, te, o_method)
                             - 1.0) / np.where(accrues, al, 1.0), 0.0)
    amounts = _sign(fl) * (fwd + fl._spread) * al * fl._notional
    fixed_tp = _times(fx._payment_dts, value_dt, dc)
    fixed_pay = _sign(fx) * fx._cpn * np.asarray(fx._year_fracs, dtype=np.float64) * fx._notional
    if fx._principal != 0.0 or fl._principal != 0.0:
        raise LibError("principal payments are not supported on this path")     # every OIS has principal 0 (ois.py:149)
    later, fixed_later = live & (tp > 0.0), fixed_tp > 0.0
    pv_const = float(amounts[live & (tp == 0.0)].sum())          # paid at the value time: discount factor 1
    none, one, zero = np.zeros(0), np.ones(1), np.zeros(1)
    flow_tp, flow_pay = np.concatenate([tp[later], fixed_tp[fixed_later]]), np.concatenate([amounts[later], fixed_pay[fixed_later]])
    flows = TradeBatch(np.array([0, flow_tp.size]), np.array([0, 0]), flow_tp, flow_pay, none, none, none, none,
                       one, zero, one, one)
    want_delta = RequestTypes.DELTA in reqs
    has_jac = getattr(xccy, "_jac_basis", None) is not None
    x_dev = _xccy_device_curve(ctx, xccy)
    on_x = _price(ctx, x_dev, flows, dict(want_value=True, want_delta=want_delta and has_jac, want_gamma=False))

    value = delta = cashflows = None
    if RequestTypes.VALUE in reqs:
        value = Valuation(amount=float(on_x["pv"][0] + pv_const) / spot, currency=collateral_ccy)
    if want_delta:
        keep = live & accrues
        weight = knot_df(x_times, x_dfs, tp[keep], x_method) / knot_df(x_times, x_dfs, 0.0, x_method)
        m = int(keep.sum())
        rates = TradeBatch(np.array([0, 0]), np.array([0, m]), none, none, np.zeros(m), ts[keep], te[keep], al[keep],
                           np.array([fl._notional]), zero, one, np.array([_sign(fl)]), flt_weight=weight)
        on_o = _price(ctx, ois_cur["dev"], rates, dict(want_value=False, want_delta=True, want_gamma=False))
        ladders = [Delta(np.array(on_o["delta"][0]) / spot, to_tenor(list(ois_model.swap_times)), collateral_ccy,
                         derivative._floating_index)]
        if has_jac:
            ladders.append(Delta(_trim(on_x["delta"][0], "delta", len(xccy.swap_times)) / spot,
                                 to_tenor(list(xccy.swap_times)), collateral_ccy,
                                 CurveTypes.USD_GBP_BASIS))
        delta = Risk(ladders)
    if RequestTypes.CASHFLOWS in reqs:
        from ...requests.results import Cashflows
        cashflows = Cashflows([], derivative._currency)
    return AnalyticsResult(value=value, risk=delta, gamma=None, cashflows=cashflows)


def _price(ctx, dev_curve, batch, kw):
    trades = _native.DeviceTrades(ctx, batch)
    try:
        return _native.price(ctx, dev_curve, trades, **kw)
    finally:
        trades.close()
