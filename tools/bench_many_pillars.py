#!/usr/bin/env python3
"""Throughput on curves of more than 32 pillars - 40 and 64 pillars on the wide variants of the general kernel (one launch),
the 40-pillar curve again on the tiled route (ADR_CURVE_PILLAR_TILES at upload: one launch per pair of 32-pillar tiles), a
96-pillar curve (beyond 64 pillars the tiles are the route) - and on
the 32-pillar curve for comparison: 100 000 benchmark trades, PV + delta + gamma and PV + delta."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades import synthetic
from adrates_amd.trades.market_data import GBP_PX, README_VALUE_DT as vd, TENORS, gbp_model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
EXTRA = ["11Y", "13Y", "14Y", "16Y", "17Y", "18Y", "19Y", "35Y"]
years = lambda t: int(t[:-1]) / {"D": 365.0, "W": 52.0, "M": 12.0, "Y": 1.0}[t[-1]]
base_t = np.array([years(t) for t in TENORS])
tenors = sorted(list(TENORS) + EXTRA, key=years)
px = [float(np.interp(years(t), base_t, GBP_PX)) if t in EXTRA else GBP_PX[TENORS.index(t)] for t in tenors]
ctx = _native.Context(0)
batch = synthetic.synthesize(vd, n)
dt = _native.DeviceTrades(ctx, batch)
dev = torch.device("cuda", 0)
extra64 = [f"{y}Y" for y in range(1, 50) if f"{y}Y" not in TENORS]
tenors64 = sorted(list(TENORS) + extra64, key=years)[:64]
px64 = [float(np.interp(years(t), base_t, GBP_PX)) if t not in TENORS else GBP_PX[TENORS.index(t)] for t in tenors64]
more = [f"{m}M" for m in range(13, 24) if m != 18] + [f"{m}M" for m in range(30, 600, 12)]
tenors96 = sorted(list(TENORS) + extra64 + more[:96 - 32 - len(extra64)], key=years)
px96 = [float(np.interp(years(t), base_t, GBP_PX)) if t not in TENORS else GBP_PX[TENORS.index(t)] for t in tenors96]
for label, model, wide in (("40 pillars", gbp_model(vd, px=px, tenors=tenors), "1"), ("40 pillars, tiled route", gbp_model(vd, px=px, tenors=tenors), "0"),
                           ("64 pillars", gbp_model(vd, px=px64, tenors=tenors64), "1"),
                           ("96 pillars (three tiles: six launches)", gbp_model(vd, px=px96, tenors=tenors96), "1"), ("32 pillars", gbp_model(vd), "1"),
                           ("31 pillars (odd count)", gbp_model(vd, px=list(GBP_PX[:13]) + list(GBP_PX[14:]), tenors=list(TENORS[:13]) + list(TENORS[14:])), "1")):
    curve = model.curves.GBP_OIS_SONIA
    host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    dc = _native.DeviceCurve(ctx, 4, host.times, host.dfs, host.jac, host.hess,
                             flags=0 if wide == "1" else _native.DeviceCurve.PILLAR_TILES)
    P = dc.n_pillars
    pv = torch.empty(n, dtype=torch.float64, device=dev); de = torch.empty((n, P), dtype=torch.float64, device=dev)
    ga = torch.empty((n, P, P), dtype=torch.float64, device=dev); ag = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
    for mask, agg_only in ((7, False), (3, False), (7, True)):      # the last: the ladder alone (knot-space sums, one projection)
        ptrs = (0, 0, 0) if agg_only else (pv.data_ptr(), de.data_ptr(), ga.data_ptr() if mask & 4 else 0)
        for _ in range(5):
            _native.price_dev(ctx, dc, dt, mask, ptrs[0], ptrs[1], ptrs[2], ag.data_ptr())
        ctx.sync()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(10):
            _native.price_dev(ctx, dc, dt, mask, ptrs[0], ptrs[1], ptrs[2], ag.data_ptr())
        ctx.sync(); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        out_bytes = 8 * (1 + P + P * P) if agg_only else 8 * (1 + P + (P * P if mask & 4 else 0)) * n
        print(json.dumps({"curve": label, "pillars": P, "trades": n, "mask": mask, "aggregate_only": agg_only, "ms": ms,
                          "trades_per_s": n / ms * 1e3, "output_GBps": out_bytes / ms / 1e6}))
