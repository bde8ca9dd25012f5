#!/usr/bin/env python3
"""Is adr_price_dev stream-capture safe?  Capture a bump ladder (65 scenario curves x one small book) into a HIP
graph through torch.cuda.CUDAGraph, replay it, compare with eager launches and time both."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adrates_amd import _native
from adrates_amd.market.position.scenarios import ScenarioGrid, bump_ladder
from adrates_amd.trades import synthetic
from adrates_amd.trades.market_data import README_VALUE_DT as vd, TENORS, readme_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
m = readme_model()
grid = ScenarioGrid(m, "GBP_OIS_SONIA", bump_ladder(TENORS, 1.0), with_gamma=False)
ctx = grid._ctx
trades = _native.DeviceTrades(ctx, synthetic.synthesize(vd, n, seed=3))
S, P = len(grid), 32
dev = torch.device("cuda", 0)
pv = torch.zeros((S, n), dtype=torch.float64, device=dev)
agg = torch.zeros((S, 1 + P + P * P), dtype=torch.float64, device=dev)
stream = torch.cuda.Stream(dev)


def launch_all():
    for i in range(S):
        _native.price_dev(ctx, grid.device_curve(i), trades, 1, pv[i].data_ptr(), 0, 0, agg[i].data_ptr(), stream.cuda_stream)


def timed(fn, reps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record(stream)
    for _ in range(reps):
        fn()
    b.record(stream); stream.synchronize()
    return a.elapsed_time(b) / reps, 1e3 * (time.perf_counter() - t0) / reps


with torch.cuda.stream(stream):
    launch_all(); stream.synchronize()
    eager = pv.clone()
    gpu_ms, wall_ms = timed(launch_all)
    pv.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        launch_all()
    g.replay(); stream.synchronize()
    same = bool(torch.equal(pv, eager))
    g_gpu_ms, g_wall_ms = timed(g.replay)
print(json.dumps({"scenarios": S, "trades": n, "graph_equals_eager": same, "eager_ms": gpu_ms, "eager_wall_ms": wall_ms,
                  "graph_ms": g_gpu_ms, "graph_wall_ms": g_wall_ms}))
