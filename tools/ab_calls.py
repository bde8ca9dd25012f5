#!/usr/bin/env python3
"""Interleaved timing of pricing-call variants on the benchmark portfolio (1 M trades): after a 300 ms warm-up the variants
are launched in turn, 8 launches each per round, for R rounds; prints the median / minimum of the per-round averages.  Unlike
tools/ablate.py (one variant after the other) no variant runs on a colder clock than another."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades import synthetic
from adrates_amd.trades.market_data import README_VALUE_DT, gbp_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
curve = gbp_model().curves.GBP_OIS_SONIA
host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
ctx = _native.Context(0)
dc = _native.DeviceCurve(ctx, 4, host.times, host.dfs, host.jac, host.hess)
dt = _native.DeviceTrades(ctx, synthetic.synthesize(README_VALUE_DT, n))
dev = torch.device("cuda", 0)
P = 32
pv = torch.empty(n, dtype=torch.float64, device=dev)
de = torch.empty((n, P), dtype=torch.float64, device=dev)
ga = torch.empty((n, P, P), dtype=torch.float64, device=dev)
ag = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
s = torch.cuda.Stream(dev)
variants = {
    "all outputs + agg": (7, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), ag.data_ptr()),
    "all outputs, no agg": (7, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), 0),
    "gamma not stored + agg": (7, pv.data_ptr(), de.data_ptr(), 0, ag.data_ptr()),
    "gamma not stored, no agg": (7, pv.data_ptr(), de.data_ptr(), 0, 0),
    "aggregate only (knot space)": (7, 0, 0, 0, ag.data_ptr()),
}
def launch(v, k):
    m, a, b, c, d = variants[v]
    for _ in range(k):
        _native.price_dev(ctx, dc, dt, m, a, b, c, d, s.cuda_stream)
with torch.cuda.stream(s):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        launch("all outputs + agg", 10); torch.cuda.synchronize()
    res = {v: [] for v in variants}
    for r in range(rounds):
        for v in variants:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            launch(v, 2)
            a.record(s); launch(v, 8); b.record(s)
            torch.cuda.synchronize()
            res[v].append(a.elapsed_time(b) / 8)
for v, x in res.items():
    print(f"{v:34s} median {statistics.median(x):8.4f} ms   min {min(x):8.4f} ms")
