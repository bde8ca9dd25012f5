#!/usr/bin/env python3
"""BASELINE.json configs[3]: a book of GBP/USD cross-currency basis swaps, PV + delta to the SONIA, SOFR and
basis pillars (32 + 32 + 17), three launches (market/position/xccy_engine.py).  The book is a few hundred distinct
swaps (tenor x seasoning x foreign frequency) compiled once and tiled with random notionals; everything is
resident in HBM when the timed region starts.  Usage: bench_xccy.py [n_swaps] [mask: 3 = PV+delta, 7 = +gamma]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adrates_amd import _native
from adrates_amd.market.position.engine import Engine
from adrates_amd.market.position import xccy_engine as XE
from adrates_amd.models.models import Model
from adrates_amd.trades.compiler import TradeBatch
from adrates_amd.trades.rates.xccy_basis_swap import XccyBasisSwap
from adrates_amd.utils import (BusDayAdjustTypes, CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes,
                               InterpTypes, SwapTypes)
from tests._fixtures import GBP_PX, README_VALUE_DT as vd, TENORS, USD_PX

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
mask = int(sys.argv[2]) if len(sys.argv) > 2 else 3
SPOT = 0.79
BASIS_TENORS = ["1Y", "18M", "2Y", "3Y", "4Y", "5Y", "6Y", "7Y", "8Y", "9Y", "10Y", "12Y", "15Y", "20Y", "25Y", "30Y", "40Y"]

m = Model(vd)
for name, px, dc in (("GBP_OIS_SONIA", GBP_PX, DayCountTypes.ACT_365F), ("USD_OIS_SOFR", USD_PX, DayCountTypes.ACT_360)):
    m.build_curve(name=name, px_list=list(px), tenor_list=list(TENORS), spot_days=0, swap_type=SwapTypes.PAY,
                  fixed_dcc_type=dc, fixed_freq_type=FrequencyTypes.ANNUAL, float_freq_type=FrequencyTypes.ANNUAL,
                  float_dc_type=dc, bus_day_type=BusDayAdjustTypes.MODIFIED_FOLLOWING,
                  interp_type=InterpTypes.FLAT_FWD_RATES)
t0 = time.perf_counter()
m.build_xccy_curve(name="USD_GBP_BASIS", domestic_curve_name="GBP_OIS_SONIA", foreign_curve_name="USD_OIS_SOFR",
                   basis_spreads=list(np.linspace(25.0, 45.0, len(BASIS_TENORS))), tenor_list=BASIS_TENORS, spot_fx=SPOT,
                   domestic_dc_type=DayCountTypes.ACT_365F, foreign_dc_type=DayCountTypes.ACT_360,
                   interp_type=InterpTypes.FLAT_FWD_RATES)
xccy_build_s = time.perf_counter() - t0

templates = []
for years in range(1, 31):
    for back in (0, 4, 9):
        for freq in (FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL):
            eff = vd.add_months(-back)
            templates.append(XccyBasisSwap(effective_dt=eff, term_dt_or_tenor=eff.add_months(12 * years + back),
                                           domestic_notional=SPOT * 1e6, foreign_notional=1e6, domestic_spread=0.0,
                                           foreign_spread=0.0030, domestic_freq_type=FrequencyTypes.ANNUAL,
                                           foreign_freq_type=freq, domestic_dc_type=DayCountTypes.ACT_365F,
                                           foreign_dc_type=DayCountTypes.ACT_360,
                                           domestic_floating_index=CurveTypes.GBP_OIS_SONIA,
                                           foreign_floating_index=CurveTypes.USD_OIS_SOFR,
                                           domestic_currency=CurrencyTypes.GBP, foreign_currency=CurrencyTypes.USD))
engine = Engine(m)
dom_model, for_model, xccy, dom_cur, for_cur, x_dev = XE._curves(engine, templates)
t0 = time.perf_counter()
parts = XE.compile_xccy(templates, vd, xccy, for_cur["host"].times, for_cur["host"].dfs, for_model._interp_type.value)
compile_s = time.perf_counter() - t0


def tile(b, pick, scale):
    """Trades `pick` of the batch, notionals (and fixed amounts) scaled."""
    def gather(off, cols, factor=None):
        cnt = np.diff(off)[pick]
        new_off = np.concatenate(([0], np.cumsum(cnt))).astype(np.int64)
        idx = np.repeat(off[:-1][pick] - new_off[:-1], cnt) + np.arange(new_off[-1])
        return new_off, [c[idx] for c in cols], np.repeat(np.arange(len(pick)), cnt)
    fo, (ftp, fpay), f_owner = gather(b.fix_off, (b.fix_tp, b.fix_pay))
    lo, cols, _ = gather(b.flt_off, (b.flt_tp, b.flt_ts, b.flt_te, b.flt_alpha) + ((b.flt_weight,) if b.flt_weight is not None else ()))
    return TradeBatch(fo, lo, ftp, fpay * scale[f_owner], cols[0], cols[1], cols[2], cols[3], b.notional[pick] * scale,
                      b.spread[pick], b.fix_sign[pick], b.flt_sign[pick], cols[4] if b.flt_weight is not None else None)


rng = np.random.default_rng(7)
pick = rng.integers(0, len(templates), n)
scale = np.round(rng.uniform(1.0, 50.0, n), 1)
ctx = dom_cur["ctx"]
book = [(_native.DeviceTrades(ctx, tile(parts[i], pick, scale)), cur) for i, cur in ((0, dom_cur["dev"]), (1, for_cur["dev"]), (2, x_dev))]
flows = [int(np.diff(parts[i].flt_off)[pick].sum() + np.diff(parts[i].fix_off)[pick].sum()) for i in range(3)]

dev = torch.device("cuda", 0)
bufs = []
for trades, cur in book:
    P = cur.n_pillars
    bufs.append((torch.empty(n, dtype=torch.float64, device=dev), torch.empty((n, P), dtype=torch.float64, device=dev),
                 torch.empty((n, P, P), dtype=torch.float64, device=dev) if mask & 4 else None,
                 torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)))
s = torch.cuda.Stream(dev)


def step():
    for (trades, cur), (pv, de, ga, ag) in zip(book, bufs):
        _native.price_dev(ctx, cur, trades, mask, pv.data_ptr(), de.data_ptr(), ga.data_ptr() if ga is not None else 0,
                          ag.data_ptr(), s.cuda_stream)


with torch.cuda.stream(s):
    for _ in range(3):
        step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    reps = 10
    per = np.zeros(3)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps):
        step()
    b.record(s); s.synchronize()
    ms = a.elapsed_time(b) / reps
    for i, ((trades, cur), (pv, de, ga, ag)) in enumerate(zip(book, bufs)):     # per-launch split
        a.record(s)
        for _ in range(reps):
            _native.price_dev(ctx, cur, trades, mask, pv.data_ptr(), de.data_ptr(), ga.data_ptr() if ga is not None else 0,
                              ag.data_ptr(), s.cuda_stream)
        b.record(s); s.synchronize()
        per[i] = a.elapsed_time(b) / reps
pillars = [cur.n_pillars for _, cur in book]
out_bytes = 8 * n * sum(1 + P + (P * P if mask & 4 else 0) for P in pillars)
print(json.dumps({"workload": "GBP/USD basis swaps, 1-30Y, annual / semi-annual foreign leg, a third seasoned",
                  "swaps": n, "mask": mask, "pillars": pillars, "cash_flows": flows, "ms": ms,
                  "swaps_per_s": n / ms * 1e3, "ms_domestic_foreignrates_foreignflows": list(per),
                  "output_GBps": out_bytes / ms / 1e6, "templates": len(templates),
                  "host_compile_templates_s": compile_s, "xccy_curve_build_s": xccy_build_s}))
