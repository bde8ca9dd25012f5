#!/usr/bin/env python3
"""Ablation timings of the wide route (33-64 pillars) on the GPU box: which part of the pass costs what.
usage: tools/ablate_wide.py [pillars=40] [trades=100000] [kind=offgrid]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades import synthetic
from adrates_amd.trades.market_data import GBP_PX, README_VALUE_DT as vd, TENORS, gbp_model

P = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
kind = sys.argv[3] if len(sys.argv) > 3 else "offgrid"
years = lambda t: int(t[:-1]) / {"D": 365.0, "W": 52.0, "M": 12.0, "Y": 1.0}[t[-1]]
base_t = np.array([years(t) for t in TENORS])
extra = ["11Y", "13Y", "14Y", "16Y", "17Y", "18Y", "19Y", "35Y"] if P == 40 else [f"{y}Y" for y in range(1, 50) if f"{y}Y" not in TENORS]
tenors = sorted(list(TENORS) + extra, key=years)[:P]
px = [float(np.interp(years(t), base_t, GBP_PX)) if t not in TENORS else GBP_PX[TENORS.index(t)] for t in tenors]
curve = gbp_model(vd, px=px, tenors=tenors).curves.GBP_OIS_SONIA
host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
ctx = _native.Context(0)
dc = _native.DeviceCurve(ctx, 4, host.times, host.dfs, host.jac, host.hess)
dt = _native.DeviceTrades(ctx, synthetic.synthesize(vd, n, kind=kind))
dev = torch.device("cuda", 0)
pv = torch.empty(n, dtype=torch.float64, device=dev)
de = torch.empty((n, P), dtype=torch.float64, device=dev)
ga = torch.empty((n, P, P), dtype=torch.float64, device=dev)
ag = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
s = torch.cuda.Stream(dev)


def run(label, mask, pvp, dep, gap, agp, reps=10):
    with torch.cuda.stream(s):
        for _ in range(2):
            _native.price_dev(ctx, dc, dt, mask, pvp, dep, gap, agp, s.cuda_stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps):
            _native.price_dev(ctx, dc, dt, mask, pvp, dep, gap, agp, s.cuda_stream)
        b.record(s)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{P} pillars {kind:8s} {label:46s} {ms:8.3f} ms  {n / ms / 1e3:8.1f} M trades/s", flush=True)


only = os.environ.get("ABLATE_ONLY")       # "stored" / "notstored": one configuration only (for counter passes)
if only:
    if only == "stored": run("value+delta+gamma, all outputs, no agg", 7, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), 0)
    else: run("value+delta+gamma computed, gamma not stored", 7, pv.data_ptr(), de.data_ptr(), 0, ag.data_ptr())
    sys.exit(0)
run("value+delta+gamma, all outputs + agg", 7, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), ag.data_ptr())
run("value+delta+gamma, all outputs, no agg", 7, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), 0)
run("value+delta+gamma computed, gamma not stored", 7, pv.data_ptr(), de.data_ptr(), 0, ag.data_ptr())
run("value+delta (no gamma)", 3, pv.data_ptr(), de.data_ptr(), 0, ag.data_ptr())
run("value only", 1, pv.data_ptr(), 0, 0, ag.data_ptr())
