"""TEST INFRASTRUCTURE - CPU restatement of the reference's *own-node* OIS curve bootstrap and queries.

Only tests may import this module (see oracle/cavour_oracle.py for the rule).  It restates, in the
reference's own recursive shape and with plain Python / numpy float64 arithmetic:

* `OISCurve._build_curve_ad`            cavour/trades/rates/ois_curve.py:156-212
* `DiscountCurve._linear_forward_interp` cavour/market/curves/discount_curve.py:385-415   (`df_ad`)
* `interpolate` / `_uinterpolate`        cavour/market/curves/interpolator.py:35-170      (`df`)

Parity status: unpinned in the absolute sense - the reference's tests for this API
(tests/test_curve_bootstrap_validation.py) assert properties (monotone DFs, rate ranges, smoothness),
not values, and the reference cannot be imported here (no jax / numba).  The product implementation
(adrates_amd/trades/rates/ois_curve.py) resolves the recursion iteratively; the two are compared
node by node in tests/test_ois_curve_nodes.py.
"""
import math

import numpy as np


def jnp_interp(x, xp, fp):
    """jax.numpy.interp for a scalar x: clamp outside, else fp[i-1] + (x - xp[i-1])/(xp[i]-xp[i-1]) * (fp[i]-fp[i-1])."""
    n = len(xp)
    if x < xp[0]:
        return fp[0]
    if x > xp[-1]:
        return fp[-1]
    i = int(np.searchsorted(np.asarray(xp), x, side="right"))
    i = min(max(i, 1), n - 1)
    dx = xp[i] - xp[i - 1]
    if abs(dx) <= np.finfo(np.float64).eps:
        return fp[i]
    return fp[i - 1] + ((x - xp[i - 1]) / dx) * (fp[i] - fp[i - 1])


def build_nodes(swap_rates, swap_times, year_fracs):
    """Returns (times, dfs, repr_dfs) as the reference appends them (ois_curve.py:156-212)."""
    times, dfs, repr_dfs = [0.0], [1.0], [1.0]
    pv01_dict = {}
    log_rates = [math.log(r) for r in swap_rates]
    df_settle = 1

    def interpolate_loglinear(t):
        return math.exp(jnp_interp(t, swap_times, log_rates))

    def calculate_single_df(i, target_maturity=None, step=0):
        if target_maturity is None:
            t_mat, swap_rate = swap_times[i], swap_rates[i]
        else:
            t_mat, swap_rate = target_maturity, interpolate_loglinear(target_maturity)
        fracs = year_fracs[i]
        if len(fracs) == 1:
            acc = fracs[0]
            df_mat = df_settle / (acc * swap_rate + 1.0)
            pv01 = acc * df_mat
        else:
            acc = fracs[-1 - step]
            last_payment = sum(fracs[:-1 - step])
            if round(last_payment, 2) not in pv01_dict:
                step += 1
                pv01_dict[round(last_payment, 2)] = calculate_single_df(i, last_payment, step)
            df_mat = (df_settle - swap_rate * pv01_dict[round(last_payment, 2)]) / (acc * swap_rate + 1)
            pv01 = pv01_dict[round(last_payment, 2)] + acc * df_mat
        times.append(t_mat)
        dfs.append(df_mat)
        if target_maturity is None:
            repr_dfs.append(df_mat)
        pv01_dict[round(t_mat, 2)] = pv01
        return pv01

    for i in range(len(swap_rates)):
        calculate_single_df(i)
    return np.array(times), np.array(dfs), np.array(repr_dfs)


def linear_forward_df(t, times, dfs):
    """`df_ad`: linear interpolation of the segments' forward rates (discount_curve.py:385-415), scalar t."""
    fwd = [-math.log(dfs[k + 1] / dfs[k]) / (times[k + 1] - times[k]) for k in range(len(times) - 1)]
    f = jnp_interp(t, list(times[:-1]), fwd)
    i0 = int(np.searchsorted(np.asarray(times), t, side="right")) - 1
    return dfs[i0] * math.exp(-f * (t - times[i0]))


def uinterpolate(t, times, dfs, method):
    """`_uinterpolate` (interpolator.py:69-170), methods 1 (FLAT_FWD), 2 (LINEAR_FWD), 4 (LINEAR_ZERO)."""
    n = len(times)
    if t == times[0]:
        return dfs[0]
    i = 0
    while times[i] < t and i < n - 1:
        i += 1
    if t > times[i]:
        i = n
    if method == 4:
        if i == 1:
            r1 = -math.log(dfs[i]) / times[i]; r2 = r1
            dt = times[i] - times[i - 1]
            return math.exp(-(((times[i] - t) * r1 + (t - times[i - 1]) * r2) / dt) * t)
        if i < n:
            r1 = -math.log(dfs[i - 1]) / times[i - 1]; r2 = -math.log(dfs[i]) / times[i]
            dt = times[i] - times[i - 1]
            return math.exp(-(((times[i] - t) * r1 + (t - times[i - 1]) * r2) / dt) * t)
        r1 = -math.log(dfs[i - 1]) / times[i - 1]; r2 = r1
        dt = times[i - 1] - times[i - 2]
        return math.exp(-(((times[i - 1] - t) * r1 + (t - times[i - 2]) * r2) / dt) * t)
    if method == 1:
        if i < n:
            rt1, rt2 = -math.log(dfs[i - 1]), -math.log(dfs[i])
            dt = times[i] - times[i - 1]
            return math.exp(-(((times[i] - t) * rt1 + (t - times[i - 1]) * rt2) / dt))
        rt1, rt2 = -math.log(dfs[i - 2]), -math.log(dfs[i - 1])
        dt = times[i - 1] - times[i - 2]
        return math.exp(-(((times[i - 1] - t) * rt1 + (t - times[i - 2]) * rt2) / dt))
    if method == 2:
        small = 1e-10
        if i == 1:
            return math.exp(-(t * -math.log(dfs[i] + small) / (times[i] + small)))
        fwd1 = -math.log(dfs[i - 1] / dfs[i - 2]) / (times[i - 1] - times[i - 2])
        if i < n:
            fwd2 = -math.log(dfs[i] / dfs[i - 1]) / (times[i] - times[i - 1])
            dt = times[i] - times[i - 1]
            fwd = ((times[i] - t) * fwd1 + (t - times[i - 1]) * fwd2) / dt
            return dfs[i - 1] * math.exp(-fwd * (t - times[i - 1]))
        return dfs[i - 1] * math.exp(-fwd1 * (t - times[i - 1]))
    raise ValueError("Invalid interpolation scheme.")
