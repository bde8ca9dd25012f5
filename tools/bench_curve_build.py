#!/usr/bin/env python3
"""Throughput of the device curve builder: S shocked versions of the README
curve, bootstrap + d/dr + d2/dr2 + table conversion, next to the host (numpy) builder on the host cores.  (The
reference's own method - differentiating the scan - is timed by tests/test_curve_tables.py, which may use the
oracle.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades.market_data import gbp_model

curve = gbp_model().curves.GBP_OIS_SONIA
base = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
K, P = base.jac.shape
ctx = _native.Context(0)
plan = _native.CurvePlan(ctx, curve._interp_type.value, base)
rng = np.random.default_rng(3)
out = {"metric": "shocked curves/sec: bootstrap + Jacobian + Hessian + kernel tables", "knots": K, "pillars": P,
       "dense_bytes_per_curve": 8 * (K + K * P + K * P * P), "runs": []}
for S in (65, 256, 1024):
    rates = np.array(curve.swap_rates)[None, :] + rng.uniform(-1e-3, 1e-3, size=(S, P))
    plan.build(rates).close()                                  # warm-up (allocator, code load)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        cs = plan.build(rates)
        cs.close()
    dt = (time.perf_counter() - t0) / reps
    out["runs"].append({"scenarios": S, "ms": 1e3 * dt, "curves_per_s": S / dt,
                        "dense_GBps": S * out["dense_bytes_per_curve"] / dt / 1e9})
t0 = time.perf_counter()
for i in range(3):
    build_engine_curve(list(np.array(curve.swap_rates) + 1e-4 * i), curve.swap_times, curve.year_fracs)
out["host_numpy_curves_per_s"] = 3 / (time.perf_counter() - t0)
print(json.dumps(out))
