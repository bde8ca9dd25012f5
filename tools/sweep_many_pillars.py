#!/usr/bin/env python3
"""One-off parity sweep of the tiled route beyond 64 pillars: mixed batches (payment lag, spreads, semi-annual floats, 1-40Y) of a
few thousand trades on 70-, 96- and 123-pillar curves, all three schemes, against oracle/port.c; prints the worst ladder error."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adrates_amd import _native
from adrates_amd.utils import InterpTypes
from oracle import port
from tests import _fixtures as F
from tests._parity import assert_batch_parity
from tests.test_gpu_many_pillars import _device_curve, _mixed_batch, many_pillar_quotes

ctx = _native.Context(0)
vd = F.README_VALUE_DT
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4001
for P in (70, 96, 123):
    for interp in (InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_FWD_RATES):
        px, tenors = many_pillar_quotes(P)
        curve = F.gbp_model(vd, interp, px=px, tenors=tenors).curves.GBP_OIS_SONIA
        host, dc = _device_curve(ctx, curve)
        batch = _mixed_batch(vd, n, seed=1000 + P)
        dt = _native.DeviceTrades(ctx, batch)
        ref = port.price(interp.value, host.times, host.dfs, host.jac, host.hess, batch)
        got = _native.price(ctx, dc, dt, aggregate=True)
        worst = assert_batch_parity(got, ref, batch.notional)
        book = _native.price(ctx, dc, dt, per_trade=False, aggregate=True)
        scale = np.abs(ref["gamma"]).sum(0).max()
        err_book = float(np.max(np.abs(book["agg_gamma"] - ref["gamma"].sum(0))) / scale)
        err_agg = float(np.max(np.abs(got["agg_gamma"] - ref["gamma"].sum(0))) / scale)
        print(f"{P} pillars {interp.name}: {n} trades, worst per-trade ladder error {worst:.2e}, aggregate {err_agg:.2e}, ladder-only {err_book:.2e}", flush=True)
        assert err_book < 1e-10 and err_agg < 1e-10
        dt.close()
