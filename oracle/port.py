"""ctypes wrapper of oracle/port.c (batched C restatement).  TEST INFRASTRUCTURE ONLY.

Imported by tests/ and by the ``cpu_baseline`` leg of bench.py - never by the
product package.  ``price`` takes the reference's curve cache arrays
(times, dfs, jac, hess) and a CSR trade batch and returns PV / delta / gamma.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libadr_port.so")
_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", _HERE])
        _lib = C.CDLL(_SO)
        _lib.adr_port_price.restype = C.c_int
        _lib.adr_port_price.argtypes = ([C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int64, _ip, _ip]
                                        + [_dp] * 10 + [_dp, _dp, _dp, C.c_int])
        _lib.adr_port_price_weighted.restype = C.c_int
        _lib.adr_port_price_weighted.argtypes = ([C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int64, _ip, _ip]
                                                 + [_dp] * 11 + [_dp, _dp, _dp, C.c_int])
        _lib.adr_port_max_threads.restype = C.c_int
    return _lib


def max_threads() -> int:
    return int(_load().adr_port_max_threads())


def _p(a, t=_dp):
    return None if a is None else a.ctypes.data_as(t)


def price(method, times, dfs, jac, hess, batch, want_delta=True, want_gamma=True, n_threads=0):
    lib = _load()
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    times, dfs, jac = f(times), f(dfs), f(jac)
    hess = None if hess is None else f(hess)
    K, P = jac.shape
    n = batch.n_trades
    arr = {k: f(getattr(batch, k)) for k in ("fix_tp", "fix_pay", "flt_tp", "flt_ts", "flt_te", "flt_alpha",
                                              "notional", "spread", "fix_sign", "flt_sign")}
    fo = np.ascontiguousarray(batch.fix_off, dtype=np.int64)
    lo = np.ascontiguousarray(batch.flt_off, dtype=np.int64)
    pv = np.empty(n)
    delta = np.empty((n, P)) if (want_delta or want_gamma) else None
    gamma = np.empty((n, P, P)) if want_gamma else None
    w = getattr(batch, "flt_weight", None)          # per-coupon weights of the cross-currency assembly, or None
    w = None if w is None else f(w)
    rc = lib.adr_port_price_weighted(K, P, int(method), _p(times), _p(dfs), _p(jac), _p(hess), n, _p(fo, _ip),
                                     _p(lo, _ip), _p(arr["fix_tp"]), _p(arr["fix_pay"]), _p(arr["flt_tp"]),
                                     _p(arr["flt_ts"]), _p(arr["flt_te"]), _p(arr["flt_alpha"]), _p(w),
                                     _p(arr["notional"]), _p(arr["spread"]), _p(arr["fix_sign"]), _p(arr["flt_sign"]),
                                     _p(pv), _p(delta), _p(gamma), int(n_threads))
    if rc != 0:
        raise RuntimeError("adr_port_price rejected its arguments")
    return dict(pv=pv, delta=delta, gamma=gamma)


def timed_baseline(curve, value_dt, method, want_gamma, budget_s, kind="offgrid"):
    """CPU baseline for bench.py: the same synthetic workload, a bounded sample, all host cores.

    The curve derivatives come from the torch.func oracle (one-off, untimed, like the GPU's table
    upload); the timed region is the batched pricing only."""
    from adrates_amd.trades import synthetic   # workload generator (inputs only)
    from . import cavour_oracle as O

    cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    # the GPU box gives one GPU a share of 16 host cores: do not fan out over the whole machine
    threads = min(max_threads(), 16)
    probe = synthetic.synthesize(value_dt, 2000, kind=kind)
    t0 = time.perf_counter()
    price(method, cache["times"], cache["dfs"], cache["jac"], cache["hess"], probe, want_gamma=want_gamma,
          n_threads=threads)
    rate = 2000 / (time.perf_counter() - t0)
    n = int(max(2000, min(2_000_000, rate * budget_s)))
    sample = synthetic.synthesize(value_dt, n, kind=kind)
    t0 = time.perf_counter()
    price(method, cache["times"], cache["dfs"], cache["jac"], cache["hess"], sample, want_gamma=want_gamma,
          n_threads=threads)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "trades/s", "cores": threads, "kind": "port",
            "sample": f"{n} trades of the same synthetic portfolio ({kind}), PV+delta"
                      + ("+gamma" if want_gamma else "") + f", {dt:.1f} s wall on {threads} OpenMP threads "
                      "(oracle/port.c: C restatement of the reference algorithm; the JAX reference cannot run here)"}


def parity_error(got, ref, notional):
    """Worst error of a batch against this port, in the metric of tests/_parity.py (north_star's 1e-10): per trade and
    ladder max|a-b| / max(max|b|, floor N) with floors 1e-4 (PV), 1e-8 (delta), 1e-12 (gamma), and SURVEY section 7's
    per-unit-notional |a-b| / max(1, |b|)."""
    n = np.abs(np.asarray(notional, dtype=np.float64))
    worst = 0.0
    for key, floor in (("pv", 1e-4), ("delta", 1e-8), ("gamma", 1e-12)):
        if ref.get(key) is None or got.get(key) is None:
            continue
        a = np.asarray(got[key], dtype=np.float64).reshape(len(n), -1)
        b = np.asarray(ref[key], dtype=np.float64).reshape(len(n), -1)
        diff = np.max(np.abs(a - b), axis=1)
        scale = np.maximum(np.max(np.abs(b), axis=1), floor * n)
        unit = np.max(np.abs(a - b) / n[:, None] / np.maximum(1.0, np.abs(b) / n[:, None]), axis=1)
        worst = max(worst, float(np.max(diff / scale)), float(np.max(unit)))
    return worst


def timed_baseline_b0(curve, value_dt, method, want_gamma, budget_s, kind="offgrid"):
    """the reference-style loop - one trade at a time through the autodiff restatement
    (oracle/cavour_oracle.py: grad / hessian w.r.t. the knot DFs, then the chain rule, as engine.py:2541-2576 does per
    trade), ONE thread, one curve cache shared by all trades (kinder than the reference, which rebuilds it per trade)."""
    import torch
    from adrates_amd.trades import synthetic
    from . import cavour_oracle as O

    cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    batch = synthetic.synthesize(value_dt, 256, kind=kind)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        done, t0 = 0, time.perf_counter()
        while done < batch.n_trades and (done < 3 or time.perf_counter() - t0 < budget_s):
            f0, f1 = int(batch.fix_off[done]), int(batch.fix_off[done + 1])
            l0, l1 = int(batch.flt_off[done]), int(batch.flt_off[done + 1])
            fixed = dict(payment_times=batch.fix_tp[f0:f1], payments=batch.fix_pay[f0:f1], principal=0.0,
                         leg_sign=float(batch.fix_sign[done]))
            floating = dict(payment_times=batch.flt_tp[l0:l1], start_times=batch.flt_ts[l0:l1],
                            end_times=batch.flt_te[l0:l1], pay_alphas=batch.flt_alpha[l0:l1],
                            spread=float(batch.spread[done]), notional=float(batch.notional[done]), principal=0.0,
                            leg_sign=float(batch.flt_sign[done]))
            O.ois_analytics(cache, method, fixed, floating, want_gamma=want_gamma)
            done += 1
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(threads)
    return {"value": done / dt, "unit": "trades/s", "cores": 1, "kind": "port",
            "sample": f"{done} trades of the same synthetic portfolio ({kind}), one at a time, PV+delta"
                      + ("+gamma" if want_gamma else "") + f", {dt:.1f} s wall on 1 thread (oracle/cavour_oracle.py: "
                      "per-trade autodiff through the knot DFs + chain rule, the reference's method; shared curve cache)"}
