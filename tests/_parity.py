"""Helpers that price a list of `OIS` objects on the GPU and with the oracle, and the parity metric."""
import numpy as np

from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades.compiler import compile_ois
from adrates_amd.utils.helpers import times_from_dates
from oracle import cavour_oracle as O

# north_star: "Results match the JAX-CPU reference's delta/gamma to 1e-10".
REL_TOL = 1e-10


def unit_notional_err(got, ref, notional):
    """SURVEY.md section 7 definition: max |a-b| / max(1, |b|) on per-unit-notional quantities."""
    a = np.asarray(got, dtype=np.float64) / notional
    b = np.asarray(ref, dtype=np.float64) / notional
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def ladder_err(got, ref, floor):
    """Stricter max-norm relative error: max |a-b| / max(max |b|, floor).  The floor keeps a ladder
    that is pure rounding noise (a par swap's PV) from being judged against its own noise."""
    a = np.asarray(got, dtype=np.float64)
    b = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), floor))


def trade_errors(got_pv, got_delta, got_gamma, ref_pv, ref_delta, ref_gamma, notional):
    n = abs(notional)
    errs = [unit_notional_err(got_pv, ref_pv, n), ladder_err(got_pv, ref_pv, 1e-4 * n)]
    if ref_delta is not None and got_delta is not None:
        errs += [unit_notional_err(got_delta, ref_delta, n), ladder_err(got_delta, ref_delta, 1e-8 * n)]
    if ref_gamma is not None and got_gamma is not None:
        errs += [unit_notional_err(got_gamma, ref_gamma, n), ladder_err(got_gamma, ref_gamma, 1e-12 * n)]
    return max(errs)


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


def assert_parity(got, refs, notionals, tol=REL_TOL):
    """``refs``: list of dicts with value/delta/(gamma); ``got``: dict of arrays from the GPU."""
    worst = 0.0
    for i, (r, n) in enumerate(zip(refs, notionals)):
        e = trade_errors(got["pv"][i], got.get("delta", [None] * (i + 1))[i] if "delta" in got else None,
                         got["gamma"][i] if "gamma" in got else None,
                         r["value"], r.get("delta"), r.get("gamma"), n)
        assert e <= tol, f"trade {i}: error {e:.3e} > {tol}"
        worst = max(worst, e)
    return worst


def assert_batch_parity(got, ref, notional, tol=REL_TOL):
    """Array form (GPU vs oracle/port.c on big batches): per-trade max-norm errors, vectorised."""
    n = np.abs(np.asarray(notional, dtype=np.float64))
    worst = 0.0
    for key, floor in (("pv", 1e-4), ("delta", 1e-8), ("gamma", 1e-12)):
        if ref.get(key) is None or got.get(key) is None:
            continue
        a = np.asarray(got[key]).reshape(len(n), -1)
        b = np.asarray(ref[key]).reshape(len(n), -1)
        diff = np.max(np.abs(a - b), axis=1)
        scale = np.maximum(np.max(np.abs(b), axis=1), floor * n)
        unit = np.max(np.abs(a - b) / n[:, None] / np.maximum(1.0, np.abs(b) / n[:, None]), axis=1)
        e = float(max(np.max(diff / scale), np.max(unit)))
        assert e <= tol, f"{key}: worst trade {int(np.argmax(diff / scale))} error {e:.3e} > {tol}"
        worst = max(worst, e)
    return worst
