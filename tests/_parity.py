"""Helpers that price a list of `OIS` objects on the GPU and with the oracle."""
import numpy as np

from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades.compiler import compile_ois
from adrates_amd.utils.helpers import times_from_dates
from oracle import cavour_oracle as O

REL_TOL = 1e-10   # north_star: "match the JAX-CPU reference's delta/gamma to 1e-10"


def rel_err(got, ref, notional):
    """max |a-b| / max(1, |b|) on per-unit-notional quantities ("Tolerance definition")."""
    a = np.asarray(got, dtype=np.float64) / notional
    b = np.asarray(ref, dtype=np.float64) / notional
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def gpu_price(ctx, curve, swaps, value_dt, **kw):
    host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    dc = _native.DeviceCurve(ctx, curve._interp_type.value, host.times, host.dfs, host.jac, host.hess)
    dt = _native.DeviceTrades(ctx, compile_ois(swaps, value_dt))
    try:
        return _native.price(ctx, dc, dt, **kw)
    finally:
        dt.close()
        dc.close()


def oracle_price(curve, swaps, value_dt, cache=None, want_gamma=True):
    cache = cache or O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    out = []
    for s in swaps:
        fx, fl = O.leg_inputs_from_swap(s, value_dt, times_from_dates)
        out.append(O.ois_analytics(cache, curve._interp_type.value, fx, fl, want_gamma=want_gamma))
    return out


def assert_parity(got, refs, swaps, tol=REL_TOL):
    worst = 0.0
    for i, (r, s) in enumerate(zip(refs, swaps)):
        n = s._notional
        e = max(rel_err(got["pv"][i], r["value"], n), rel_err(got["delta"][i], r["delta"], n))
        if "gamma" in r and "gamma" in got:
            e = max(e, rel_err(got["gamma"][i], r["gamma"], n))
        assert e <= tol, f"trade {i}: error {e:.3e} > {tol}"
        worst = max(worst, e)
    return worst
