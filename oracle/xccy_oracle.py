"""TEST INFRASTRUCTURE - CPU restatement of the reference's XCCY curve bootstrap with torch autodiff.

Only tests may import this module.  It follows the reference's differentiable builder,
cavour/trades/rates/xccy_curve.py: `_prepare_ad_inputs` (:707-937) for the payment points,
`_run_jax_bootstrap_impl` (:954-1206) for the scan, `_build_curve_ad` (:529-703) for the four derivative
tensors - with `torch.func.jacrev / jacfwd` where the reference uses the JAX transforms of the same names.
The product (adrates_amd/trades/rates/xccy_curve.py) evaluates the recurrence on forward-mode jets instead.

Parity status: unpinned - the reference cannot be imported here and its tests for this curve
(tests/test_xccy_curve.py) assert properties only.  What narrows it: the scan, the three ladders, the gammas and the
cross term are re-evaluated without any differentiation in 60-digit arithmetic (oracle/mp_oracle.py::MpXccy, central
differences w.r.t. the quotes) and agree to 1e-11 (tests/test_mp_third_evaluation.py).
"""
import torch
from torch.func import jacfwd, jacrev

_F64 = torch.float64


def _interp(x, xp, fp):
    """jnp.interp (clamped, linear) for tensors: xp static [n], fp tensor [n], x static scalar/1-D."""
    x = torch.as_tensor(x, dtype=_F64)
    i = torch.clamp(torch.searchsorted(xp, x, right=True), 1, xp.numel() - 1)
    w = (x - xp[i - 1]) / (xp[i] - xp[i - 1])
    f = fp[i - 1] + w * (fp[i] - fp[i - 1])
    f = torch.where(x < xp[0], fp[0], f)
    return torch.where(x > xp[-1], fp[-1], f)


def payment_points(curve_value_dt, swaps, foreign_curve, times_from_dates):
    """`_prepare_ad_inputs`: one point per foreign payment date >= value date, sorted by (time, swap).  The
    reference reads the leg AFTER `value()` has inserted the effective-date notional exchange into it."""
    pts = []
    for s, swap in enumerate(swaps):
        leg = swap._foreign_leg
        pay = list(leg._payment_dts); start = list(leg._start_accrued_dts); end = list(leg._end_accrued_dts)
        yf = list(leg._year_fracs); N = leg._notional
        if leg._notional_exchange and leg._effective_dt >= curve_value_dt:
            pay = [leg._effective_dt] + pay; start = [leg._effective_dt] + start; end = [leg._effective_dt] + end
            yf = [0.0] + yf
        for j, dt in enumerate(pay):
            if dt >= curve_value_dt:
                exch = abs(yf[j]) < 1e-10
                pts.append(dict(time=(dt - curve_value_dt) / 365.0, swap=s, is_mat=(dt == swap._maturity_dt),
                                at_val=(dt == curve_value_dt), yf=yf[j], N=N, exch=exch,
                                last=(dt == swap._maturity_dt) and leg._notional_exchange,
                                sens=0.0 if exch else yf[j] * N,
                                ts=times_from_dates(start[j], curve_value_dt, foreign_curve._dc_type),
                                te=times_from_dates(end[j], curve_value_dt, foreign_curve._dc_type),
                                df_ois=foreign_curve.df(dt, foreign_curve._dc_type)))
    pts.sort(key=lambda p: (p["time"], p["swap"]))
    return pts


def scan(pts, pv_dom, pillar_spreads, df_ois, spot_fx, f_times, f_dfs):
    """`_run_jax_bootstrap_impl`: returns the DF of every point as a tensor [n_points]."""
    f_times = torch.as_tensor(f_times, dtype=_F64)
    log_f = torch.log(torch.as_tensor(f_dfs, dtype=_F64))
    n = len(pts)
    out = []
    pv_contrib, cf_contrib = [], []
    prev = -1
    for i, p in enumerate(pts):
        basis = pillar_spreads[p["swap"]]
        df_s = torch.exp(_interp(p["ts"], f_times, log_f)); df_e = torch.exp(_interp(p["te"], f_times, log_f))
        fwd = (df_s / df_e - 1.0) / max(p["yf"], 1e-10) if p["yf"] > 1e-10 else torch.zeros((), dtype=_F64)
        interest = fwd * p["yf"] * p["N"] + (p["N"] if p["last"] else 0.0)
        base = torch.as_tensor(p["N"] if p["last"] else -p["N"], dtype=_F64) if p["exch"] else interest
        cashflow = base + basis * p["sens"]
        if prev < 0:
            df_mid = df_ois[i] * torch.exp(-basis * p["time"])
        else:
            df_mid = out[prev] * (df_ois[i] / df_ois[prev]) * torch.exp(-basis * (p["time"] - pts[prev]["time"]))
        is_known = (not p["is_mat"]) and (not p["at_val"])
        total = cashflow * df_mid if is_known else (cashflow * 1.0 if p["at_val"] else torch.zeros((), dtype=_F64))
        cf_here = cashflow if p["is_mat"] else torch.zeros((), dtype=_F64)
        same = [j for j in range(i) if pts[j]["swap"] == p["swap"]]
        pv_known = sum([pv_contrib[j] for j in same], torch.zeros((), dtype=_F64)) + total
        cf_mat = sum([cf_contrib[j] for j in same], torch.zeros((), dtype=_F64)) + cf_here
        pv_contrib.append(total); cf_contrib.append(cf_here)
        numerator = -(pv_dom[p["swap"]] + spot_fx * (pv_known * -1.0))
        denominator = spot_fx * (cf_mat * -1.0)
        if p["is_mat"]:
            safe = torch.where(torch.abs(denominator) > 1e-12, denominator, torch.ones((), dtype=_F64))
            out.append(torch.where(torch.abs(denominator) > 1e-12, numerator / safe, df_mid))
        else:
            out.append(df_mid)
        if not p["at_val"]:
            prev = i
    return torch.stack(out)


def build(curve_value_dt, swaps, domestic_curve, foreign_curve, spot_fx, times_from_dates):
    """times, dfs and the four derivative tensors as `_build_curve_ad` stores them."""
    pts = payment_points(curve_value_dt, swaps, foreign_curve, times_from_dates)
    pv_dom = [s._domestic_leg.value(curve_value_dt, domestic_curve, domestic_curve) for s in swaps]
    spreads = torch.tensor([s._foreign_spread for s in swaps], dtype=_F64)
    f_times = torch.tensor(list(foreign_curve._times), dtype=_F64)
    f_dfs = torch.tensor(list(foreign_curve._dfs), dtype=_F64)
    df_ois_values = torch.tensor([p["df_ois"] for p in pts], dtype=_F64)
    pay_times = torch.tensor([p["time"] for p in pts], dtype=_F64)
    nodes, seen = [], set()
    for i, p in enumerate(pts):
        if p["at_val"] or round(p["time"], 4) in seen:
            continue
        seen.add(round(p["time"], 4)); nodes.append(i)
    idx = torch.tensor(nodes)

    def from_basis(b):
        return torch.cat([torch.ones(1, dtype=_F64), scan(pts, pv_dom, b, df_ois_values, spot_fx, f_times, f_dfs)[idx]])

    def from_basis_and_foreign(b, f):
        df_ois = torch.exp(_interp(pay_times, f_times, torch.log(f)))       # re-interpolated, ACT/365 times
        return torch.cat([torch.ones(1, dtype=_F64), scan(pts, pv_dom, b, df_ois, spot_fx, f_times, f_dfs)[idx]])

    dfs = from_basis(spreads)
    return dict(times=torch.cat([torch.zeros(1, dtype=_F64), pay_times[idx]]).numpy(), dfs=dfs.numpy(),
                jac_basis=jacrev(from_basis)(spreads).numpy(),
                hess_basis=jacfwd(jacrev(from_basis))(spreads).numpy(),
                jac_foreign=jacrev(from_basis_and_foreign, argnums=1)(spreads, f_dfs).numpy(),
                mixed=jacrev(jacfwd(from_basis_and_foreign, argnums=1), argnums=0)(spreads, f_dfs)
                .permute(0, 2, 1).numpy())


# ------------------------------------------------------------------------------------------------------------
# Engine._compute_xccy (cavour/market/position/engine.py:1411-1988): VALUE, the three delta ladders and the
# three gamma matrices of a cross-currency basis swap.  The cross-gamma block (:1895-1960) is not restated: it
# contracts a tensor indexed by the foreign curve's OWN nodes with the Jacobian of the engine's knot grid,
# whose sizes differ.
# ------------------------------------------------------------------------------------------------------------
import numpy as np
from torch.func import grad, hessian

from . import cavour_oracle as O


def float_leg_dual(disc_dfs, disc_times, disc_method, idx_dfs, idx_times, idx_method, payment_times, start_times,
                   end_times, alphas, spread, notional, leg_sign, exchange, t_eff, t_mat, value_time=0.0):
    """`_float_leg_jax` (engine.py:2639-2728) with a separate index curve and notional exchanges."""
    payment_times = np.asarray(payment_times, dtype=np.float64)
    a_np = np.asarray(alphas, dtype=np.float64)
    a = torch.as_tensor(a_np)
    df_val = O.simple_interpolate(value_time, disc_times, disc_dfs, disc_method)
    df_s = O.simple_interpolate(np.asarray(start_times, dtype=np.float64), idx_times, idx_dfs, idx_method)
    df_e = O.simple_interpolate(np.asarray(end_times, dtype=np.float64), idx_times, idx_dfs, idx_method)
    pos = torch.as_tensor(a_np > 0)
    fwd = torch.where(pos, (df_s / df_e - 1.0) / torch.where(pos, a, torch.ones_like(a)), torch.zeros_like(a))
    cf = (fwd + spread) * a * notional
    df_rel = O.simple_interpolate(payment_times, disc_times, disc_dfs, disc_method) / df_val
    valid = torch.as_tensor(payment_times >= value_time)
    pv = torch.where(valid, cf * df_rel, torch.zeros_like(df_rel)).sum()
    if exchange:
        if t_eff >= value_time:
            pv = pv - notional * O.simple_interpolate(t_eff, disc_times, disc_dfs, disc_method) / df_val
        if t_mat >= value_time:
            pv = pv + notional * O.simple_interpolate(t_mat, disc_times, disc_dfs, disc_method) / df_val
    return leg_sign * pv


def _chain(pv_fn, dfs, jac, hess):
    d = torch.as_tensor(np.asarray(dfs), dtype=_F64)
    J = torch.as_tensor(np.asarray(jac), dtype=_F64)
    g = grad(pv_fn)(d)
    H = hessian(pv_fn)(d)
    gamma = J.T @ H @ J + torch.sum(g[:, None, None] * torch.as_tensor(np.asarray(hess), dtype=_F64), dim=0)
    return float(pv_fn(d)), (g @ J).numpy(), gamma.numpy()


def xccy_analytics(swap, value_dt, dom_cache, dom_method, for_cache, for_method, xccy_curve, times_from_dates):
    """dict(value, delta_dom, delta_for, delta_basis, gamma_dom, gamma_for, gamma_basis) in domestic currency,
    per bp / bp^2 (engine.py:1578, 1680-1733, 1769-1880)."""
    spot = xccy_curve._spot_fx
    dl, fl = swap._domestic_leg, swap._foreign_leg
    receive = type(dl._leg_type).RECEIVE
    ddc, fdc, xdc = dl._dc_type, fl._dc_type, xccy_curve._dc_type
    T = lambda dts, dc: np.array([times_from_dates(d, value_dt, dc) for d in dts])
    x_times, x_dfs, x_method = np.asarray(xccy_curve._times), np.asarray(xccy_curve._dfs), xccy_curve._interp_type.value

    def pv_dom(d):
        return float_leg_dual(d, dom_cache["times"], dom_method, d, dom_cache["times"], dom_method,
                              T(dl._payment_dts, ddc), T(dl._start_accrued_dts, ddc), T(dl._end_accrued_dts, ddc),
                              dl._year_fracs, dl._spread, dl._notional, 1.0 if dl._leg_type == receive else -1.0,
                              dl._notional_exchange, times_from_dates(swap._effective_dt, value_dt, ddc),
                              times_from_dates(swap._maturity_dt, value_dt, ddc))

    def pv_for(x_d, f_d):
        return float_leg_dual(x_d, x_times, x_method, f_d, for_cache["times"], for_method,
                              T(fl._payment_dts, xdc), T(fl._start_accrued_dts, fdc), T(fl._end_accrued_dts, fdc),
                              fl._year_fracs, fl._spread, fl._notional, 1.0 if fl._leg_type == receive else -1.0,
                              fl._notional_exchange, times_from_dates(swap._effective_dt, value_dt, xdc),
                              times_from_dates(swap._maturity_dt, value_dt, xdc))

    f_fixed = torch.as_tensor(np.asarray(for_cache["dfs"]), dtype=_F64)
    x_fixed = torch.as_tensor(x_dfs, dtype=_F64)
    v_dom, d_dom, g_dom = _chain(pv_dom, dom_cache["dfs"], dom_cache["jac"], dom_cache["hess"])
    v_for, d_for, g_for = _chain(lambda f: pv_for(x_fixed, f), for_cache["dfs"], for_cache["jac"], for_cache["hess"])
    _, d_bas, g_bas = _chain(lambda x: pv_for(x, f_fixed), x_dfs, xccy_curve._jac_basis, xccy_curve._hess_basis)
    # mixed second derivative foreign OIS rates x basis spreads with both curves' knot DFs as the only channels
    # (the XCCY curve's own dependence on the foreign curve through its bootstrap held fixed - the part
    # engine.py:1895-1960 computes instead): J_for^T (d2 PV / d d_f d d_x) J_basis, [P_for, P_basis]
    from torch.func import jacrev as _jacrev
    mixed = _jacrev(grad(pv_for, argnums=1), argnums=0)(x_fixed, f_fixed)                 # [K_for, K_x]
    J_f = torch.as_tensor(np.asarray(for_cache["jac"]), dtype=_F64)
    J_b = torch.as_tensor(np.asarray(xccy_curve._jac_basis), dtype=_F64)
    cross = (J_f.T @ mixed @ J_b).numpy()
    return dict(cross_for_basis=cross * 1e-8 / spot, value=v_dom + v_for / spot, delta_dom=d_dom * 1e-4, delta_for=d_for * 1e-4 / spot,
                delta_basis=d_bas * 1e-4 / spot, gamma_dom=g_dom * 1e-8, gamma_for=g_for * 1e-8 / spot,
                gamma_basis=g_bas * 1e-8 / spot)


def ois_xccy_collateral_analytics(swap, value_dt, ois_cache, ois_method, xccy_curve, times_from_dates):
    """`Engine._compute_ois_xccy_collateral` (engine.py:217-503): an OIS discounted on an XCCY curve, forwards
    off its own OIS curve; PV and the two delta ladders in collateral currency.  All times in the fixed leg's
    day count (:264-283); the reference has no gamma for this path (:489-494)."""
    spot = xccy_curve._spot_fx
    fx, fl = swap._fixed_leg, swap._float_leg
    receive = type(fx._leg_type).RECEIVE
    dc = fx._dc_type
    T = lambda dts: np.array([times_from_dates(d, value_dt, dc) for d in dts])
    x_times, x_dfs, x_method = np.asarray(xccy_curve._times), np.asarray(xccy_curve._dfs), xccy_curve._interp_type.value
    fixed_payments = fx._cpn * np.asarray(fx._year_fracs, dtype=np.float64) * fx._notional
    fixed_tp, tp, ts, te = T(fx._payment_dts), T(fl._payment_dts), T(fl._start_accrued_dts), T(fl._end_accrued_dts)

    def pv(x_d, o_d):
        fixed = O.price_fixed_leg(x_d, x_times, x_method, fixed_tp, fixed_payments, fx._principal,
                                  1.0 if fx._leg_type == receive else -1.0)
        floating = float_leg_dual(x_d, x_times, x_method, o_d, ois_cache["times"], ois_method, tp, ts, te, fl._year_fracs,
                                  fl._spread, fl._notional, 1.0 if fl._leg_type == receive else -1.0, False, 0.0, 0.0)
        return fixed + floating

    x_t = torch.as_tensor(x_dfs, dtype=_F64)
    o_t = torch.as_tensor(np.asarray(ois_cache["dfs"]), dtype=_F64)
    g_o = grad(lambda o: pv(x_t, o))(o_t)
    g_x = grad(lambda x: pv(x, o_t))(x_t)
    J_o = torch.as_tensor(np.asarray(ois_cache["jac"]), dtype=_F64)
    J_x = torch.as_tensor(np.asarray(xccy_curve._jac_basis), dtype=_F64)
    return dict(value=float(pv(x_t, o_t)) / spot, delta_ois=(g_o @ J_o).numpy() * 1e-4 / spot,
                delta_basis=(g_x @ J_x).numpy() * 1e-4 / spot)
