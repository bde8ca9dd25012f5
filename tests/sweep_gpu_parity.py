#!/usr/bin/env python3
"""Parity sweep at scale (run on the GPU box; not collected by pytest): random curves x the three interpolation
schemes x mixed portfolios of 100 000 trades (frequencies, spreads, payment lags, seasoned / forward-starting,
pay / receive), HIP path against the C oracle (oracle/port.c).  Prints one JSON line per case and a summary;
`python tests/sweep_gpu_parity.py [cases] [trades] [first case]`.  Error metric as in tests/_parity.py."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades.compiler import OISTerms, compile_ois_terms
from adrates_amd.utils import BusDayAdjustTypes, CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes, InterpTypes
from oracle import port
from tests import _fixtures as F

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # first case number (cases are seeded by their number)
vd = F.README_VALUE_DT
tenors = list(F.TENORS)
ctx = _native.default_context()
schemes = [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_FWD_RATES]
worst_all = 0.0
t_start = time.time()
skipped = 0
for case in range(first, first + cases):
    rng = np.random.default_rng(9000 + case)
    interp = schemes[case % 3]
    x = np.linspace(0.0, 1.0, len(tenors))
    level = rng.uniform(0.6, 8.0)
    px = np.maximum(level + rng.uniform(-0.4, 0.6) * level * x + rng.uniform(-0.2, 0.2) * level * np.sin(np.pi * x)
                    + rng.normal(0, 0.01, len(tenors)), 0.05)
    curve = F.gbp_model(vd, interp, px=list(px), tenors=tenors).curves.GBP_OIS_SONIA
    host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    if not (np.all(np.isfinite(host.dfs)) and np.all(host.dfs > 0.0)):
        skipped += 1           # quotes this steep bootstrap to a negative discount factor: the library refuses them
        print(json.dumps({"case": case, "interp": interp.name, "level_pct": round(level, 2), "skipped": "non-positive DF"}), flush=True)
        continue
    dc = _native.DeviceCurve(ctx, interp.value, host.times, host.dfs, host.jac, host.hess)
    starts = [vd, vd.add_months(-7), vd.add_years(-2), vd.add_months(5), vd.add_weekdays(2)]
    eff = [starts[i] for i in rng.choice(5, size=n, p=[0.5, 0.15, 0.1, 0.15, 0.1])]
    lfreq = [[FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL, FrequencyTypes.QUARTERLY][i]
             for i in rng.choice(3, size=n, p=[0.6, 0.25, 0.15])]
    ffreq = [[FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL][i] for i in rng.choice(2, size=n, p=[0.8, 0.2])]
    terms = OISTerms(effective_dt=eff, tenor=[f"{int(m)}M" for m in rng.integers(1, 361, n)],
                     coupon=rng.uniform(0.0, 0.09, n), notional=np.round(rng.uniform(1e5, 9e7, n), -4),
                     pay_fixed=rng.random(n) < 0.5, fixed_freq_type=ffreq, fixed_dc_type=DayCountTypes.ACT_365F,
                     floating_index=CurveTypes.GBP_OIS_SONIA, currency=CurrencyTypes.GBP, float_freq_type=lfreq,
                     float_dc_type=[[DayCountTypes.ACT_365F, DayCountTypes.ACT_360][i] for i in rng.integers(0, 2, n)],
                     float_spread=np.where(rng.random(n) < 0.3, rng.uniform(-0.002, 0.004, n), 0.0),
                     payment_lag=rng.choice([0, 0, 0, 1, 2], size=n), bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
    batch = compile_ois_terms(terms, vd)
    trades = _native.DeviceTrades(ctx, batch)
    got = _native.price(ctx, dc, trades, aggregate=True)
    ref = port.price(interp.value, host.times, host.dfs, host.jac, host.hess, batch)
    N = np.abs(batch.notional)
    errs = {}
    for key, floor in (("pv", 1e-4), ("delta", 1e-8), ("gamma", 1e-12)):
        a, b = got[key].reshape(n, -1), ref[key].reshape(n, -1)
        diff = np.max(np.abs(a - b), axis=1)
        errs[key] = float(np.max(np.maximum(diff / np.maximum(np.max(np.abs(b), axis=1), floor * N),
                                            np.max(np.abs(a - b) / N[:, None] / np.maximum(1.0, np.abs(b) / N[:, None]), axis=1))))
    agg = float(np.max(np.abs(got["agg_gamma"] - ref["gamma"].sum(0))) / max(1e-30, np.max(np.abs(ref["gamma"].sum(0)))))
    worst = max(errs.values())
    worst_all = max(worst_all, worst)
    print(json.dumps({"case": case, "interp": interp.name, "level_pct": round(level, 2), "trades": n,
                      "cash_flows": int(batch.flt_tp.size + batch.fix_tp.size), "max_err": errs, "agg_gamma_rel": agg}), flush=True)
    trades.close(); dc.close()
print(json.dumps({"summary": True, "cases": cases, "first_case": first, "skipped": skipped, "trades_per_case": n, "worst_error": worst_all, "tolerance": 1e-10,
                  "pass": bool(worst_all <= 1e-10), "seconds": round(time.time() - t_start, 1)}))
