#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fast kernel (needs a library built with -DADR_STAMPS)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades import synthetic
from adrates_amd.trades.market_data import README_VALUE_DT, gbp_model
n = 1_000_000
curve = gbp_model().curves.GBP_OIS_SONIA
host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
ctx = _native.Context(0)
dc = _native.DeviceCurve(ctx, 4, host.times, host.dfs, host.jac, host.hess)
dt = _native.DeviceTrades(ctx, synthetic.synthesize(README_VALUE_DT, n))
dev = torch.device("cuda", 0); P = 32
pv = torch.empty(n, dtype=torch.float64, device=dev); de = torch.empty((n, P), dtype=torch.float64, device=dev)
ga = torch.empty((n, P, P), dtype=torch.float64, device=dev); ag = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
for mask, lab in ((7, "value+delta+gamma"), (3, "value+delta"), (1, "value")):
    _native.price_dev(ctx, dc, dt, mask, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), ag.data_ptr()); ctx.sync()
    lib = _native.load(); nw = 256 * 12
    buf = np.zeros((nw, 8), dtype=np.uint64)
    lib.adr_debug_stamps(ctx._h, buf.ctypes.data_as(C.c_void_p), nw)
    tot = buf.sum(0).astype(float)
    names = ["header", "cashflow loads+folding", "lookup+exp", "walk: entry / exit", "outputs", "walk: record, Jacobian rows, loop tail", "walk: v, first-order sums, convexity coefficient", "walk: rank-one update"]
    print(lab, "cycles per wave:", int(buf.sum(1).mean()))
    for nm, v in zip(names, tot[:8]): print(f"   {nm:26s} {100 * v / tot.sum():5.1f} %")
