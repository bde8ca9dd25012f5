#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the lite kernel (needs a library built with -DADR_STAMPS:
make -C adrates_amd/csrc OUT=../../variants_stamps.so OBJDIR=build_stamps EXTRA=-DADR_STAMPS, then
ADRATES_HIP_LIB=$PWD/variants_stamps.so python tools/stamps_lite.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades import synthetic
from adrates_amd.trades.market_data import README_VALUE_DT, gbp_model
n = 1_000_000
curve = gbp_model().curves.GBP_OIS_SONIA
host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
ctx = _native.Context(0)
dc = _native.DeviceCurve(ctx, 4, host.times, host.dfs, host.jac, host.hess)
dt = _native.DeviceTrades(ctx, synthetic.synthesize(README_VALUE_DT, n))
dev = torch.device("cuda", 0); P = 32
pv = torch.empty(n, dtype=torch.float64, device=dev); de = torch.empty((n, P), dtype=torch.float64, device=dev)
ag = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
lib = _native.load()
lib.adr_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for mask, lab in ((3, "value+delta"), (1, "value")):
    _native.price_dev(ctx, dc, dt, mask, pv.data_ptr(), de.data_ptr(), 0, ag.data_ptr()); ctx.sync()
    nw = 512 * 8
    buf = np.zeros((nw, 8), dtype=np.uint64)
    lib.adr_debug_stamps(ctx._h, buf.ctypes.data_as(C.c_void_p), nw)
    tot = buf.sum(0).astype(float)
    names = ["input wait", "folding", "lookup+exp", "entries+ladder", "outputs", "next-step requests"]
    print(lab, "cycles per wave:", int(buf.sum(1).mean()), " waves with work:", int((buf.sum(1) > 0).sum()))
    for nm, v in zip(names, tot[:6]): print(f"   {nm:26s} {100 * v / tot.sum():5.1f} %")
