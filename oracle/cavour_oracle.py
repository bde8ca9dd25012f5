"""CPU oracle for the OIS PV / delta / gamma path.  TEST INFRASTRUCTURE ONLY.

This module restates, function by function, the algorithm of the reference's
valuation engine so that the HIP path can be checked against it.  It is imported
only by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` - never by the product package ``adrates_amd``.

Parity status: **parity unpinned in the absolute sense.**  The reference
(Python + JAX) cannot be imported in the build container (no jax / numba / xbbg,
no network - SURVEY.md section 8(c)) and its tests hold no golden numbers for
this path.  The oracle is therefore pinned by (i) the notebook outputs of the
reference (`notebooks/intro.ipynb` cells 36-44: 1W swap PV, delta ladder, total
gamma), (ii) the reference tests' properties (calibration swaps reprice through
the engine to <= 1e-5, AD delta vs bump-and-reprice, gamma symmetry, pay/receive
antisymmetry) - see tests/test_oracle_pins.py.

Where the reference differentiates with `jax.grad / jax.hessian / jax.jacrev`,
the oracle differentiates the same restated function with `torch.func` in
float64, i.e. it performs the same *kind* of computation (AD through the scan,
dense knot-space Hessian, chain rule through the curve Jacobian/Hessian) and
shares no closed-form derivative code with the product.

Reference map (all paths relative to /root/reference):
  expand_points / bootstrap      cavour/market/position/engine.py:2246-2360
  cached_curve                   cavour/market/position/engine.py:2362-2412
  simple_interpolate             cavour/market/curves/interpolator_ad.py:186-249
                                 (+ jax.numpy.interp: searchsorted side='right',
                                  clip to [1, K-1], constant outside the range)
  price_fixed_leg                cavour/market/position/engine.py:2414-2448
  float_leg                      cavour/market/position/engine.py:2639-2728
  leg_analytics                  cavour/market/position/engine.py:2498-2576, 2808-2934
  ois_analytics                  cavour/market/position/engine.py:153-215
"""
from __future__ import annotations

import numpy as np
import torch
from torch.func import grad, hessian, jacrev

FLAT_FWD_RATES = 1
LINEAR_FWD_RATES = 2
LINEAR_ZERO_RATES = 4

_F64 = torch.float64


# --------------------------------------------------------------------------- knots
def expand_points(swap_rates, swap_times, year_fracs):
    """Knot grid of the engine's bootstrap (engine.py:2283-2334).

    Returns numpy arrays ``times[K]`` (exact cumulative accruals, sorted, with
    duplicates kept), ``acc[K]``, ``rate_idx[K]`` (pillar whose par rate the
    point uses; the t=0 point borrows pillar 0), ``prev_idx[K]`` (first sorted
    point whose 2-decimal key equals the point's previous-coupon key, -1 if
    none) and the list of key collisions (distinct maturities > 1e-6 apart that
    share a rounded key) so that affected curves can be flagged.
    """
    points = [dict(maturity=0.0, key=0.0, acc=0.0, prev_key=None, swap=0)]
    for i, fracs in enumerate(year_fracs):
        cumsum = 0.0
        for j, frac in enumerate(fracs):
            frac = float(frac)
            prev_cum = cumsum
            cumsum += frac
            points.append(dict(maturity=cumsum, key=round(cumsum, 2), acc=frac,
                               prev_key=round(prev_cum, 2) if j > 0 else None, swap=i))
    ordered = sorted(points, key=lambda p: p["maturity"])  # stable: ties keep swap order

    first_with_key = {}
    collisions = []
    for idx, p in enumerate(ordered):
        if p["key"] not in first_with_key:
            first_with_key[p["key"]] = idx
        else:
            other = ordered[first_with_key[p["key"]]]
            if abs(other["maturity"] - p["maturity"]) > 1e-6:
                collisions.append((p["key"], other["maturity"], p["maturity"]))
    prev_idx = [(-1 if p["prev_key"] is None else first_with_key.get(p["prev_key"], -1))
                for p in ordered]
    return (np.array([p["maturity"] for p in ordered], dtype=np.float64),
            np.array([p["acc"] for p in ordered], dtype=np.float64),
            np.array([p["swap"] for p in ordered], dtype=np.int64),
            np.array(prev_idx, dtype=np.int64),
            collisions)


def _bootstrap_dfs(rates, acc, rate_idx, prev_idx):
    """The `lax.scan` body of engine.py:2337-2354 as a Python loop over torch
    scalars; ``pv01`` starts as zeros exactly like the scan's carry."""
    K = len(acc)
    zero = rates.new_zeros(())
    pv01 = [zero] * K
    dfs = []
    for i in range(K):
        r = rates[int(rate_idx[i])]
        a = float(acc[i])
        if prev_idx[i] < 0:
            prev = zero
            d = 1.0 / (1.0 + r * a)
        else:
            prev = pv01[int(prev_idx[i])]
            d = (1.0 - r * prev) / (1.0 + r * a)
        pv01[i] = prev + a * d
        dfs.append(d)
    return torch.stack(dfs)


def cached_curve(swap_rates, swap_times, year_fracs, derivatives=True):
    """times, dfs, d(dfs)/d(rates) and d2(dfs)/d(rates)2 - the cache dict of
    engine.py:2362-2412 (the `times[0] > 1e-7` prepend never triggers because
    the grid already starts at t = 0).  ``derivatives=False`` skips the two AD
    passes (bump-and-reprice checks only need values)."""
    times, acc, rate_idx, prev_idx, collisions = expand_points(swap_rates, swap_times, year_fracs)
    rates = torch.tensor([float(r) for r in swap_rates], dtype=_F64)

    def f(r):
        return _bootstrap_dfs(r, acc, rate_idx, prev_idx)

    dfs = f(rates)
    assert times[0] <= 1e-7
    out = dict(times=times, dfs=dfs.numpy().copy(), acc=acc, rate_idx=rate_idx, prev_idx=prev_idx,
               collisions=collisions)
    if derivatives:
        out["jac"] = jacrev(f)(rates).numpy().copy()
        out["hess"] = hessian(f)(rates).numpy().copy()
    return out


# ------------------------------------------------------------------- interpolation
def _interp(tau, xp, fp):
    """jax.numpy.interp(tau, xp, fp) for constant abscissae ``xp`` (numpy) and
    differentiable ordinates ``fp`` (torch)."""
    K = len(xp)
    i = np.clip(np.searchsorted(xp, tau, side="right"), 1, K - 1)
    dx = xp[i] - xp[i - 1]
    delta = tau - xp[i - 1]
    eps = np.spacing(np.finfo(np.float64).eps)
    dx0 = np.abs(dx) <= eps
    w = torch.as_tensor(delta / np.where(dx0, 1.0, dx), dtype=_F64)
    lo = fp[torch.as_tensor(i - 1)]
    hi = fp[torch.as_tensor(i)]
    f = torch.where(torch.as_tensor(dx0), lo, lo + w * (hi - lo))
    f = torch.where(torch.as_tensor(tau < xp[0]), fp[0], f)
    f = torch.where(torch.as_tensor(tau > xp[-1]), fp[-1], f)
    return f


def simple_interpolate(t, times, dfs, method):
    """InterpolatorAd.simple_interpolate (interpolator_ad.py:186-249): exact
    knot hits (|t - x_k| < 1e-10) return that knot's DF (first such knot on
    ties), otherwise interpolate at t + 1e-12."""
    x = np.asarray(times, dtype=np.float64)
    d = dfs if isinstance(dfs, torch.Tensor) else torch.as_tensor(np.asarray(dfs), dtype=_F64)
    tt = np.atleast_1d(np.asarray(t, dtype=np.float64))

    dist = np.abs(tt[:, None] - x[None, :])
    grid_idx = np.argmin(dist, axis=1)  # first index on ties, like jnp.argmin
    at_grid = dist[np.arange(len(tt)), grid_idx] < 1e-10

    tau = tt + 1e-12
    tt_t = torch.as_tensor(tt, dtype=_F64)
    if method == LINEAR_ZERO_RATES:
        r = -torch.log(d) / torch.as_tensor(np.maximum(x, 1e-15), dtype=_F64)
        interp_result = torch.exp(-_interp(tau, x, r) * tt_t)
    elif method == FLAT_FWD_RATES:
        interp_result = torch.exp(-_interp(tau, x, -torch.log(d)))
    elif method == LINEAR_FWD_RATES:
        interp_result = _interp(tau, x, d)
    else:
        raise ValueError("Invalid interpolation scheme.")
    out = torch.where(torch.as_tensor(at_grid), d[torch.as_tensor(grid_idx)], interp_result)
    return out[0] if np.ndim(t) == 0 else out


# --------------------------------------------------------------------------- legs
def price_fixed_leg(dfs, times, method, payment_times, payments, principal, leg_sign,
                    value_time=0.0):
    """engine.py:2414-2448."""
    payment_times = np.asarray(payment_times, dtype=np.float64)
    df_val = simple_interpolate(value_time, times, dfs, method)
    df_pmts = simple_interpolate(payment_times, times, dfs, method)
    mask = torch.as_tensor(payment_times > value_time)
    df_rel = df_pmts / df_val
    pays = torch.as_tensor(np.asarray(payments, dtype=np.float64))
    pv_coupons = torch.where(mask, pays * df_rel, torch.zeros_like(df_rel))
    pv_prin = torch.where(mask[-1], principal * df_rel[-1], torch.zeros((), dtype=_F64))
    return leg_sign * (pv_coupons.sum() + pv_prin)


def float_leg(dfs, times, method, payment_times, start_times, end_times, pay_alphas,
              spreads, notionals, principal, leg_sign, value_time=0.0):
    """engine.py:2639-2728 for the single-curve case (index curve = discount
    curve, no first-fixing override, no notional exchange - the OIS call,
    engine.py:167-177)."""
    payment_times = np.asarray(payment_times, dtype=np.float64)
    alphas_np = np.asarray(pay_alphas, dtype=np.float64)
    alphas = torch.as_tensor(alphas_np)
    spreads = torch.as_tensor(np.asarray(spreads, dtype=np.float64))
    notionals = torch.as_tensor(np.asarray(notionals, dtype=np.float64))

    df_val = simple_interpolate(value_time, times, dfs, method)
    df_start = simple_interpolate(np.asarray(start_times, dtype=np.float64), times, dfs, method)
    df_end = simple_interpolate(np.asarray(end_times, dtype=np.float64), times, dfs, method)
    pos = torch.as_tensor(alphas_np > 0)
    safe_alpha = torch.where(pos, alphas, torch.ones_like(alphas))
    fwd = torch.where(pos, (df_start / df_end - 1.0) / safe_alpha, torch.zeros_like(alphas))
    cf_amounts = (fwd + spreads) * alphas * notionals
    df_pmts = simple_interpolate(payment_times, times, dfs, method)
    df_rel = df_pmts / df_val
    valid = torch.as_tensor(payment_times >= value_time)
    pv_coupons = torch.where(valid, cf_amounts * df_rel, torch.zeros_like(df_rel))
    pv_prin = torch.where(valid[-1], principal * df_rel[-1], torch.zeros((), dtype=_F64))
    return leg_sign * (pv_coupons.sum() + pv_prin)


def _leg_analytics(pv_fn, cache, want_gamma=True):
    """VALUE / DELTA / GAMMA assembly shared by engine.py:2541-2576 and
    :2899-2934: gradient and dense Hessian w.r.t. the knot DFs, then the chain
    rule through the curve Jacobian / Hessian; units 1e-4 (per bp) and 1e-8."""
    dfs = torch.as_tensor(cache["dfs"], dtype=_F64)
    jac = torch.as_tensor(cache["jac"], dtype=_F64)
    out = dict(value=float(pv_fn(dfs)))
    grad_dfs = grad(pv_fn)(dfs)
    out["delta"] = (grad_dfs @ jac).numpy() * 1e-4
    out["grad_dfs"] = grad_dfs.numpy().copy()
    if want_gamma:
        hess_curve = torch.as_tensor(cache["hess"], dtype=_F64)
        hess_dfs = hessian(pv_fn)(dfs)
        term1 = jac.T @ hess_dfs @ jac
        term2 = torch.sum(grad_dfs[:, None, None] * hess_curve, dim=0)
        out["gamma"] = np.array((term1 + term2).numpy(), dtype=np.float64) * 1e-8
    return out


def ois_value(cache, method, fixed, floating):
    """PV only (no AD), for bump-and-reprice checks."""
    dfs = torch.as_tensor(cache["dfs"], dtype=_F64)
    m = len(floating["payment_times"])
    v = price_fixed_leg(dfs, cache["times"], method, fixed["payment_times"], fixed["payments"],
                        fixed.get("principal", 0.0), fixed["leg_sign"])
    v = v + float_leg(dfs, cache["times"], method, floating["payment_times"], floating["start_times"],
                      floating["end_times"], floating["pay_alphas"], np.full(m, floating["spread"]),
                      np.full(m, floating["notional"]), floating.get("principal", 0.0), floating["leg_sign"])
    return float(v)


def ois_analytics(cache, method, fixed, floating, want_gamma=True):
    """PV, delta[P], gamma[P,P] of one OIS (engine.py:153-215).

    ``fixed``    : dict(payment_times, payments, principal, leg_sign)
    ``floating`` : dict(payment_times, start_times, end_times, pay_alphas,
                        spread, notional, principal, leg_sign)
    """
    times = cache["times"]

    def fixed_pv(d):
        return price_fixed_leg(d, times, method, fixed["payment_times"], fixed["payments"],
                               fixed.get("principal", 0.0), fixed["leg_sign"])

    m = len(floating["payment_times"])

    def float_pv(d):
        return float_leg(d, times, method, floating["payment_times"], floating["start_times"],
                         floating["end_times"], floating["pay_alphas"],
                         np.full(m, floating["spread"]), np.full(m, floating["notional"]),
                         floating.get("principal", 0.0), floating["leg_sign"])

    fx = _leg_analytics(fixed_pv, cache, want_gamma)
    fl = _leg_analytics(float_pv, cache, want_gamma)
    out = dict(value=fx["value"] + fl["value"], delta=fx["delta"] + fl["delta"],
               fixed=fx, floating=fl)
    if want_gamma:
        out["gamma"] = fx["gamma"] + fl["gamma"]
    return out


# ------------------------------------------------------- adapters for test inputs
def leg_inputs_from_swap(swap, value_dt, times_from_dates):
    """Per-trade arrays exactly as the engine extracts them from the leg objects
    (engine.py:2519-2527, 2858-2877).  ``swap`` is any object exposing the
    reference's leg attributes; ``times_from_dates`` is passed in so that the
    oracle does not import the product package."""
    fl, xl = swap._fixed_leg, swap._float_leg
    receive = type(fl._leg_type).RECEIVE
    fixed = dict(
        payment_times=np.array([times_from_dates(dt, value_dt, fl._dc_type) for dt in fl._payment_dts]),
        payments=np.array(fl._payments, dtype=np.float64),
        principal=fl._principal,
        leg_sign=+1.0 if fl._leg_type == receive else -1.0)
    floating = dict(
        payment_times=np.array([times_from_dates(dt, value_dt, xl._dc_type) for dt in xl._payment_dts]),
        start_times=np.array([times_from_dates(dt, value_dt, xl._dc_type) for dt in xl._start_accrued_dts]),
        end_times=np.array([times_from_dates(dt, value_dt, xl._dc_type) for dt in xl._end_accrued_dts]),
        pay_alphas=np.array(xl._year_fracs, dtype=np.float64),
        spread=xl._spread, notional=xl._notional, principal=xl._principal,
        leg_sign=+1.0 if xl._leg_type == receive else -1.0)
    return fixed, floating
