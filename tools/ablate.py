#!/usr/bin/env python3
"""Ablation timings on the GPU box: which part of the pricing pass costs what (1e6-trade benchmark portfolio).
Round 4: every variant runs after a 300 ms warm-up of the clock (the first variant used to run on a colder clock than the
others - that, not the aggregate, was most of the 6.7 % round 3 charged to it); tools/ab_calls.py interleaves the
variants and is the figure to quote for small differences."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades import synthetic
from adrates_amd.trades.market_data import README_VALUE_DT, gbp_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "offgrid"
curve = gbp_model().curves.GBP_OIS_SONIA
host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
ctx = _native.Context(0)
dc = _native.DeviceCurve(ctx, 4, host.times, host.dfs, host.jac, host.hess)
dt = _native.DeviceTrades(ctx, synthetic.synthesize(README_VALUE_DT, n, kind=kind))
dev = torch.device("cuda", 0)
P = 32
pv = torch.empty(n, dtype=torch.float64, device=dev)
de = torch.empty((n, P), dtype=torch.float64, device=dev)
ga = torch.empty((n, P, P), dtype=torch.float64, device=dev)
ag = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
s = torch.cuda.Stream(dev)

def run(label, mask, pvp, dep, gap, agp, reps=10):
    with torch.cuda.stream(s):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:          # clock ramp
            for _ in range(5):
                _native.price_dev(ctx, dc, dt, mask, pvp, dep, gap, agp, s.cuda_stream)
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(reps):
            _native.price_dev(ctx, dc, dt, mask, pvp, dep, gap, agp, s.cuda_stream)
        b.record(s)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{label:46s} {ms:8.3f} ms  {n / ms / 1e3:8.1f} M trades/s")

run("value+delta+gamma, all outputs + agg", 7, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), ag.data_ptr())
run("value+delta+gamma, all outputs, no agg", 7, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), 0)
run("value+delta+gamma computed, gamma not stored", 7, pv.data_ptr(), de.data_ptr(), 0, ag.data_ptr())
run("value+delta (no gamma)", 3, pv.data_ptr(), de.data_ptr(), 0, ag.data_ptr())
run("value only", 1, pv.data_ptr(), 0, 0, ag.data_ptr())
run("aggregate only (ladder, no per-trade output)", 7, 0, 0, 0, ag.data_ptr())
run("aggregate only, value+delta", 3, 0, 0, 0, ag.data_ptr())
# pure store bandwidth reference: memset of the gamma buffer
with torch.cuda.stream(s):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ga.zero_(); a.record(s)
    for _ in range(10): ga.zero_()
    b.record(s)
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print(f"{'reference: torch zero_ of the gamma buffer':46s} {ms:8.3f} ms  {ga.numel() * 8 / ms / 1e6:8.1f} GB/s")
