#!/usr/bin/env python3
"""Parity of the cross-currency book at scale (run on the GPU box; not collected by pytest): the three trade batches
of a synthetic book of GBP/USD basis swaps (adrates_amd/trades/synthetic_xccy.py) priced by the HIP kernels and by
the C oracle (oracle/port.c, including its per-coupon weights) on the same curve tables.
`python tests/sweep_gpu_xccy.py [swaps]`; error metric as in tests/_parity.py."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from adrates_amd import _native
from adrates_amd.market.position import xccy_engine as XE
from adrates_amd.market.position.engine import Engine
from adrates_amd.trades import synthetic_xccy as SX
from adrates_amd.trades.market_data import GBP_PX, README_VALUE_DT as vd, TENORS, USD_PX
from oracle import port

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
m = SX.build_market(vd, GBP_PX, USD_PX, TENORS)
engine = Engine(m)
parts, spot = SX.synthesize_book(engine, vd, n)
dom_model, for_model, xccy, dom_cur, for_cur, x_dev = XE._curves(engine, SX.template_swaps(vd)[:1])
jac, hess = np.asarray(xccy._jac_basis), np.asarray(xccy._hess_basis)
if jac.shape[1] % 2:                      # the engine pads an odd basis ladder (xccy_engine._xccy_device_curve)
    jac, hess = np.pad(jac, ((0, 0), (0, 1))), np.pad(hess, ((0, 0), (0, 1), (0, 1)))
tables = [(dom_model._interp_type.value, dom_cur["host"].times, dom_cur["host"].dfs, dom_cur["host"].jac, dom_cur["host"].hess),
          (for_model._interp_type.value, for_cur["host"].times, for_cur["host"].dfs, for_cur["host"].jac, for_cur["host"].hess),
          (xccy._interp_type.value, np.asarray(xccy._times), np.asarray(xccy._dfs), jac, hess)]
ctx = _native.default_context()
worst_all = 0.0
for name, (batch, dev), tab in zip(("domestic", "foreign_rates", "foreign_flows"), parts, tables):
    t0 = time.time()
    ref = port.price(*tab, batch)
    cpu_s = time.time() - t0
    trades = _native.DeviceTrades(ctx, batch)
    got = _native.price(ctx, dev, trades, aggregate=True)
    N = np.abs(batch.notional)
    unit, ladder = {}, {}
    for key, floor in (("pv", 1e-4), ("delta", 1e-8), ("gamma", 1e-12)):
        a, b = got[key].reshape(n, -1), ref[key].reshape(n, -1)
        # |a - b| <= 1e-10 max(1, |b|) on per-unit-notional quantities
        unit[key] = float(np.max(np.abs(a - b) / N[:, None] / np.maximum(1.0, np.abs(b) / N[:, None])))
        # the stricter per-ladder relative error of tests/_parity.py; meaningless where the true ladder is zero -
        # the domestic leg of these swaps (no spread, notional exchanged) is worth par and has no sensitivity
        ladder[key] = float(np.max(np.max(np.abs(a - b), axis=1) / np.maximum(np.max(np.abs(b), axis=1), floor * N)))
    judged = max(unit.values()) if name == "domestic" else max(max(unit.values()), max(ladder.values()))
    worst_all = max(worst_all, judged)
    print(json.dumps({"piece": name, "swaps": n, "cash_flows": int(batch.flt_tp.size + batch.fix_tp.size),
                      "pillars": int(dev.n_pillars), "per_unit_notional_err": unit, "ladder_rel_err": ladder,
                      "judged": judged, "max_abs_ladder_per_notional": float(np.max(np.abs(ref["gamma"]) / N[:, None, None])),
                      "cpu_oracle_s": round(cpu_s, 2)}), flush=True)
    trades.close()
print(json.dumps({"summary": True, "swaps": n, "worst_error": worst_all, "tolerance": 1e-10, "pass": bool(worst_all <= 1e-10)}))
