#!/usr/bin/env python3
"""Diagnostic: where the host time of a cross-currency book goes (terms -> batches -> upload), under cProfile."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adrates_amd import _native
from adrates_amd.market.position.engine import Engine
from adrates_amd.market.position import xccy_engine as XE
from adrates_amd.trades import synthetic_xccy as SX
from adrates_amd.trades.market_data import GBP_PX, README_VALUE_DT as vd, TENORS, USD_PX
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
m = SX.build_market(vd, GBP_PX, USD_PX, TENORS)
engine = Engine(m)
XE.book_batches(engine, SX.draw_terms(vd, 100)[0])
terms, _ = SX.draw_terms(vd, n)
ctx = _native.default_context()
for rep in range(2):
    t0 = time.perf_counter()
    out = XE.book_batches(engine, terms)
    t1 = time.perf_counter()
    book = [_native.DeviceTrades(ctx, b) for b in out[6]]
    t2 = time.perf_counter()
    print(f"rep {rep}: terms -> batches {t1 - t0:.3f} s, upload {t2 - t1:.3f} s")
    del book
pr = cProfile.Profile(); pr.enable()
out = XE.book_batches(engine, terms)
book = [_native.DeviceTrades(ctx, b) for b in out[6]]
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
