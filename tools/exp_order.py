#!/usr/bin/env python3
"""Does the ADDRESS ORDER of the gamma stores matter?  The fast kernel walks its row table sorted by coupon count and writes
each trade's 8 KB matrix at the trade's ORIGINAL index - scattered over the 8 GB buffer.  Same portfolio, trades handed over
(a) in random order (the bench), (b) pre-sorted by coupon count, longest first (the row table's order = the batch's order: the
waves of a launch then write neighbouring chunks), (c) sorted inside windows of W trades."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades import synthetic
from adrates_amd.trades.compiler import TradeBatch
from adrates_amd.trades.market_data import README_VALUE_DT, gbp_model

def permute(b, order):
    lf, ll = np.diff(b.fix_off)[order], np.diff(b.flt_off)[order]
    def gather(off, lens):
        starts = off[:-1][order]
        idx = np.repeat(starts - np.r_[0, np.cumsum(lens)[:-1]], lens) + np.arange(int(lens.sum()))
        return idx
    fi, li = gather(b.fix_off, lf), gather(b.flt_off, ll)
    return TradeBatch(np.r_[0, np.cumsum(lf)].astype(np.int64), np.r_[0, np.cumsum(ll)].astype(np.int64), b.fix_tp[fi], b.fix_pay[fi],
                      b.flt_tp[li], b.flt_ts[li], b.flt_te[li], b.flt_alpha[li], b.notional[order], b.spread[order], b.fix_sign[order], b.flt_sign[order])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
curve = gbp_model().curves.GBP_OIS_SONIA
host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
ctx = _native.Context(0)
dc = _native.DeviceCurve(ctx, 4, host.times, host.dfs, host.jac, host.hess)
base = synthetic.synthesize(README_VALUE_DT, n)
m = np.diff(base.flt_off)
orders = {"random (bench)": np.arange(n), "sorted by coupon count": np.argsort(-m, kind="stable")}
for W in (64, 1024, 16384):
    o = np.arange(n)
    for lo in range(0, n, W):
        seg = o[lo:lo + W]
        o[lo:lo + W] = seg[np.argsort(-m[seg], kind="stable")]
    orders[f"sorted inside windows of {W}"] = o
dev = torch.device("cuda", 0)
P = 32
pv = torch.empty(n, dtype=torch.float64, device=dev); de = torch.empty((n, P), dtype=torch.float64, device=dev)
ga = torch.empty((n, P, P), dtype=torch.float64, device=dev); ag = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
s = torch.cuda.Stream(dev)
dts = {k: _native.DeviceTrades(ctx, permute(base, o)) for k, o in orders.items()}
res = {k: [] for k in orders}
with torch.cuda.stream(s):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10): _native.price_dev(ctx, dc, dts["random (bench)"], 7, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), ag.data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
    for r in range(8):
        for k, dt in dts.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(2): _native.price_dev(ctx, dc, dt, 7, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), ag.data_ptr(), s.cuda_stream)
            a.record(s)
            for _ in range(8): _native.price_dev(ctx, dc, dt, 7, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), ag.data_ptr(), s.cuda_stream)
            b.record(s); torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) / 8)
for k, x in res.items():
    print(f"{k:36s} median {statistics.median(x):8.4f} ms   min {min(x):8.4f} ms")
