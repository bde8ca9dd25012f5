#!/usr/bin/env python3
"""Latency of the reference's README call on one trade: swap.position(model).compute([VALUE, DELTA, GAMMA]), warm (curve cached on the
device), and of its parts: compile the trade, upload, adr_price (blocking form), results objects."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adrates_amd import _native
from adrates_amd.trades.compiler import compile_ois
from adrates_amd.trades.market_data import README_VALUE_DT as vd, make_swap, readme_model
from adrates_amd.utils import RequestTypes

m = readme_model()
swap = make_swap(vd, "10Y", 0.045, 1e7)
reqs = [RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA]
pos = swap.position(m)
for _ in range(20):
    pos.compute(reqs)
N = 300
t0 = time.perf_counter()
for _ in range(N):
    swap.position(m).compute(reqs)
full = (time.perf_counter() - t0) / N
eng = pos._engine if hasattr(pos, "_engine") else None
ctx = _native.default_context()
from adrates_amd.market.position.engine import Engine
cur = Engine(m)._device_curve(m.curves.GBP_OIS_SONIA)
t0 = time.perf_counter()
for _ in range(N):
    batch = compile_ois([swap], vd)
compile_s = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N):
    tr = _native.DeviceTrades(ctx, batch); tr.close()
upload_s = (time.perf_counter() - t0) / N
tr = _native.DeviceTrades(ctx, batch)
t0 = time.perf_counter()
for _ in range(N):
    _native.price(ctx, cur["dev"], tr)
price_s = (time.perf_counter() - t0) / N
print(json.dumps({"call": "swap.position(model).compute([VALUE, DELTA, GAMMA]), one 10Y OIS, warm", "us_per_call": round(1e6 * full, 1),
                  "us_compile_trade": round(1e6 * compile_s, 1), "us_upload_and_free": round(1e6 * upload_s, 1),
                  "us_adr_price_blocking": round(1e6 * price_s, 1)}))
