#!/usr/bin/env python3
"""Timings of the lite kernel's passes on the benchmark portfolio with the loaded library (ADRATES_HIP_LIB picks a build):
PV + delta at 100 k and 1 M trades (BASELINE configs[1] and its large form), PV alone, the aggregate-only ladder.  One line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades import synthetic
from adrates_amd.trades.market_data import README_VALUE_DT, gbp_model

curve = gbp_model().curves.GBP_OIS_SONIA
host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
ctx = _native.Context(0)
dc = _native.DeviceCurve(ctx, 4, host.times, host.dfs, host.jac, host.hess)
dev = torch.device("cuda", 0)
P = 32
s = torch.cuda.Stream(dev)
out = {}
for n in (100_000, 1_000_000):
    dt = _native.DeviceTrades(ctx, synthetic.synthesize(README_VALUE_DT, n))
    pv = torch.empty(n, dtype=torch.float64, device=dev); de = torch.empty((n, P), dtype=torch.float64, device=dev)
    ag = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
    for label, args in (("pv_delta", (3, pv.data_ptr(), de.data_ptr(), 0, ag.data_ptr())), ("pv", (1, pv.data_ptr(), 0, 0, ag.data_ptr())),
                        ("ladder_only", (7, 0, 0, 0, ag.data_ptr()))):
        with torch.cuda.stream(s):
            for _ in range(20):
                _native.price_dev(ctx, dc, dt, *args, s.cuda_stream)
            best = 1e9
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(s)
                for _ in range(50):
                    _native.price_dev(ctx, dc, dt, *args, s.cuda_stream)
                b.record(s); s.synchronize()
                best = min(best, a.elapsed_time(b) / 50)
        out[f"{label}_{n // 1000}k_us"] = round(best * 1e3, 2)
    dt.close()
print(json.dumps(out))
