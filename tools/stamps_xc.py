#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the lite kernel's two-curve launch (adr_price_xccy_foreign) and of the two batches it
replaces, on the 100 k-swap book of tools/bench_xccy.py (needs a library built with -DADR_STAMPS: ADRATES_HIP_LIB=...)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adrates_amd import _native
from adrates_amd.market.position.engine import Engine
from adrates_amd.market.position import xccy_engine as XE
from adrates_amd.trades import synthetic_xccy as SX
from adrates_amd.trades.market_data import GBP_PX, README_VALUE_DT as vd, TENORS, USD_PX
n = 100_000
m = SX.build_market(vd, GBP_PX, USD_PX, TENORS)
engine = Engine(m)
terms, _ = SX.draw_terms(vd, n)
dom_model, for_model, xccy, dom_cur, for_cur, x_dev, batches, pv_const, spot, raw = XE.book_batches(engine, terms)
ctx = _native.default_context()
dom_b, for_b, _ = XE.compile_xccy_legs(raw, spot)
rates_tr, flows_tr, for_tr = _native.upload_many(ctx, [batches[1], batches[2], for_b])
dev = torch.device("cuda", 0)
Pf, Px = for_cur["dev"].n_pillars, x_dev.n_pillars
pv = torch.empty(n, dtype=torch.float64, device=dev)
df = torch.empty((n, Pf), dtype=torch.float64, device=dev); dx = torch.empty((n, Px), dtype=torch.float64, device=dev)
lib = _native.load()
names = ["waiting for the step's inputs", "folding", "lookups + exp", "entries + ladders", "outputs", "requesting the next inputs"]


def report(label, launch, waves):
    launch(); ctx.sync()
    buf = np.zeros((waves, 8), dtype=np.uint64)
    lib.adr_debug_stamps(ctx._h, buf.ctypes.data_as(C.c_void_p), waves)
    live = buf[buf.sum(1) > 0]
    if len(live) == 0:
        print(label, ": no stamps recorded"); return
    tot = live.sum(0).astype(float)
    print(label, "waves", len(live), "cycles per wave:", int(live.sum(1).mean()))
    for nm, v in zip(names, tot[:6]):
        print(f"   {nm:32s} {100 * v / tot.sum():5.1f} %")


report("two-curve launch", lambda: _native.price_xccy_foreign_dev(ctx, for_cur["dev"], x_dev, for_tr, 3, pv.data_ptr(), df.data_ptr(), dx.data_ptr()), 256 * 12)
report("foreign-rates batch", lambda: _native.price_dev(ctx, for_cur["dev"], rates_tr, 2, 0, df.data_ptr(), 0, 0), 256 * 12)
report("foreign-flows batch", lambda: _native.price_dev(ctx, x_dev, flows_tr, 3, pv.data_ptr(), dx.data_ptr(), 0, 0), 512 * 8)
