#!/usr/bin/env python3
"""Wall-clock latency of the reference-style calls: one trade through `position(model).compute(...)`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adrates_amd.utils import RequestTypes, CollateralType
from adrates_amd.trades.market_data import README_VALUE_DT as vd, make_swap, readme_model

REQ = [RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA]


def clock(fn, reps):
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return 1e3 * (time.perf_counter() - t0) / reps


t0 = time.perf_counter()
m = readme_model()
build_ms = 1e3 * (time.perf_counter() - t0)
swap = make_swap(vd, "10Y", 0.045, 1e7)
first_ms = clock(lambda: swap.position(m).compute(REQ), 1)          # curve derivatives + tables + upload + price
again_ms = clock(lambda: swap.position(m).compute(REQ), 50)         # a new Position (and Engine) per call, like the reference
pos = swap.position(m)
same_pos_ms = clock(lambda: pos.compute(REQ), 50)
out = {"model_build_curve_ms": build_ms, "ois_first_compute_ms": first_ms, "ois_compute_new_position_ms": again_ms,
       "ois_compute_same_position_ms": same_pos_ms}

from adrates_amd.trades import synthetic_xccy as SX
from adrates_amd.trades.market_data import GBP_PX, TENORS, USD_PX
t0 = time.perf_counter()
mx = SX.build_market(vd, GBP_PX, USD_PX, TENORS)
out["xccy_market_build_ms"] = 1e3 * (time.perf_counter() - t0)
x = SX.template_swaps(vd)[60]
out["xccy_first_compute_ms"] = clock(lambda: x.position(mx).compute(REQ), 1)
out["xccy_compute_ms"] = clock(lambda: x.position(mx).compute(REQ), 20)
print(json.dumps(out))
