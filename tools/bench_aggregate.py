#!/usr/bin/env python3
"""Aggregate-only mode (agg and no per-trade output) on the benchmark portfolio: ms per launch and trades/s for
PV + delta + gamma and PV + delta, next to the per-trade kernels with the stores off (the route before round 4).
usage: bench_aggregate.py [n_trades] [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades import synthetic
from adrates_amd.trades.market_data import README_VALUE_DT, gbp_model
from adrates_amd.utils import InterpTypes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = _native.Context(0)
dev = torch.device("cuda", 0)
out = {"trades": n}
batch = synthetic.synthesize(README_VALUE_DT, n)
dt = _native.DeviceTrades(ctx, batch)
s = torch.cuda.Stream(dev)
for interp in (InterpTypes.LINEAR_ZERO_RATES, InterpTypes.LINEAR_FWD_RATES):
    curve = gbp_model(README_VALUE_DT, interp).curves.GBP_OIS_SONIA
    host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    dc = _native.DeviceCurve(ctx, interp.value, host.times, host.dfs, host.jac, host.hess)
    P = dc.n_pillars
    ag = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
    pv = torch.empty(n, dtype=torch.float64, device=dev)
    for label, mask, pvp in (("value+delta+gamma", 7, 0), ("value+delta", 3, 0), ("value+delta+gamma, per-trade kernels, stores off (pv kept)", 7, pv.data_ptr())):
        with torch.cuda.stream(s):
            for _ in range(200 if mask == 3 or not pvp else 20):            # clock ramp
                _native.price_dev(ctx, dc, dt, mask, pvp, 0, 0, ag.data_ptr(), s.cuda_stream)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            for _ in range(reps):
                _native.price_dev(ctx, dc, dt, mask, pvp, 0, 0, ag.data_ptr(), s.cuda_stream)
            b.record(s)
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        out[f"{interp.name}: {label}"] = {"ms": ms, "trades_per_s": n / ms * 1e3, "input_GBps": dt.input_bytes / ms / 1e6}
print(json.dumps(out, indent=1))
