#!/usr/bin/env python3
"""Throughput on trades with more than 32 coupons per leg (quarterly floats, 10-30Y): fast kernel with chained
rows vs what the general kernel did before (run with ADRATES_HIP_LIB pointing at an older build to compare)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades.compiler import OISTerms, compile_ois_terms
from adrates_amd.utils import BusDayAdjustTypes, CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes
from adrates_amd.trades.market_data import README_VALUE_DT as vd, gbp_model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
mode = sys.argv[2] if len(sys.argv) > 2 else "long"
noagg = len(sys.argv) > 4 and sys.argv[4] == "noagg"     # per-trade outputs only (no aggregate ladder)
aggonly = len(sys.argv) > 4 and sys.argv[4] == "aggonly"  # the ladder alone (no per-trade output: the knot-space passes)
mask = int(sys.argv[3]) if len(sys.argv) > 3 else 7       # 1 value, 3 value+delta, 7 value+delta+gamma       # "long": quarterly 10-30Y; "lag": annual, 2-day payment lag
curve = gbp_model().curves.GBP_OIS_SONIA
host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
ctx = _native.Context(0)
interp = int(os.environ.get("ADR_BENCH_INTERP", "4"))       # 4 LINEAR_ZERO_RATES, 1 FLAT_FWD_RATES, 2 LINEAR_FWD_RATES
dc = _native.DeviceCurve(ctx, interp, host.times, host.dfs, host.jac, host.hess)
rng = np.random.default_rng(2)
months = rng.integers(120, 361, n) if mode in ("long", "longlag") else rng.integers(1, 361, n)
terms = OISTerms(vd, [f"{int(m)}M" for m in months], rng.uniform(0.01, 0.07, n), np.round(rng.uniform(1e6, 5e7, n), -5),
                 rng.random(n) < 0.5, FrequencyTypes.ANNUAL, DayCountTypes.ACT_365F, CurveTypes.GBP_OIS_SONIA,
                 CurrencyTypes.GBP, float_freq_type=FrequencyTypes.QUARTERLY if mode in ("long", "longlag") else FrequencyTypes.ANNUAL,
                 float_dc_type=DayCountTypes.ACT_365F, payment_lag=0 if mode == "long" else 2,       # "longlag": quarterly 10-30Y float legs paid 2 business days late
                 bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
batch = compile_ois_terms(terms, vd)
dt = _native.DeviceTrades(ctx, batch)
dev = torch.device("cuda", 0); P = 32
pv = torch.empty(n, dtype=torch.float64, device=dev); de = torch.empty((n, P), dtype=torch.float64, device=dev)
ga = torch.empty((n, P, P), dtype=torch.float64, device=dev); ag = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(2):
        _native.price_dev(ctx, dc, dt, mask, 0 if aggonly else pv.data_ptr(), 0 if aggonly else de.data_ptr(), 0 if aggonly else ga.data_ptr(), 0 if noagg else ag.data_ptr(), s.cuda_stream)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(5):
        _native.price_dev(ctx, dc, dt, mask, 0 if aggonly else pv.data_ptr(), 0 if aggonly else de.data_ptr(), 0 if aggonly else ga.data_ptr(), 0 if noagg else ag.data_ptr(), s.cuda_stream)
    b.record(s); s.synchronize()
ms = a.elapsed_time(b) / 5
print(json.dumps({"mode": mode, "mask": mask, "trades": n, "mean_float_coupons": float(np.diff(batch.flt_off).mean()), "ms": ms,
                  "trades_per_s": n / ms * 1e3, "aggregate": not noagg, "aggregate_only": aggonly, "interp": interp, "lib": os.environ.get("ADRATES_HIP_LIB", "default")}))
