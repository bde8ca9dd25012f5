#!/usr/bin/env python3
"""BASELINE.json configs[3]: a book of GBP/USD cross-currency basis swaps, PV + delta to the SONIA, SOFR and
basis pillars (32 + 32 + 17), three launches (market/position/xccy_engine.py).  The book
(adrates_amd/trades/synthetic_xccy.py) is resident in HBM when the timed region starts.
Usage: bench_xccy.py [n_swaps] [mask: 3 = PV+delta, 7 = +gamma]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from adrates_amd import _native
from adrates_amd.market.position.engine import Engine
from adrates_amd.trades import synthetic_xccy as SX
from adrates_amd.trades.market_data import GBP_PX, README_VALUE_DT as vd, TENORS, USD_PX

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
mask = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = time.perf_counter()
m = SX.build_market(vd, GBP_PX, USD_PX, TENORS)
market_s = time.perf_counter() - t0
from adrates_amd.market.position import xccy_engine as XE
engine = Engine(m)
XE._curves(engine, SX.template_swaps(vd)[:1])             # curve tables on the device (once per market, untimed)
t0 = time.perf_counter()
terms, _ = SX.draw_terms(vd, n)
draw_s = time.perf_counter() - t0
t0 = time.perf_counter()
dom_model, for_model, xccy, dom_cur, for_cur, x_dev, batches, pv_const, spot, _ = XE.book_batches(engine, terms)
compile_s = time.perf_counter() - t0                      # terms -> three trade batches, device curve lookups included
parts = [(batches[0], dom_cur["dev"]), (batches[1], for_cur["dev"]), (batches[2], x_dev)]
ctx = _native.default_context()
t0 = time.perf_counter()
book = list(zip(_native.upload_many(ctx, [b for b, _ in parts]), [cur for _, cur in parts]))
upload_s = time.perf_counter() - t0
flows = [int(b.flt_tp.size + b.fix_tp.size) for b, _ in parts]

dev = torch.device("cuda", 0)
bufs = []
for trades, cur in book:
    P = cur.n_pillars
    bufs.append((torch.empty(n, dtype=torch.float64, device=dev), torch.empty((n, P), dtype=torch.float64, device=dev),
                 torch.empty((n, P, P), dtype=torch.float64, device=dev) if mask & 4 else None,
                 torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)))
s = torch.cuda.Stream(dev)


def launch(i):
    (trades, cur), (pv, de, ga, ag) = book[i], bufs[i]
    _native.price_dev(ctx, cur, trades, mask, pv.data_ptr(), de.data_ptr(), ga.data_ptr() if ga is not None else 0,
                      ag.data_ptr(), s.cuda_stream)


def timed(fn, reps=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps):
        fn()
    b.record(s); s.synchronize()
    return a.elapsed_time(b) / reps


def launch_ladder_only(i):            # the book's three ladders alone (no per-swap output): the knot-space passes
    (trades, cur), (_, _, _, ag) = book[i], bufs[i]
    _native.price_dev(ctx, cur, trades, mask, 0, 0, 0, ag.data_ptr(), s.cuda_stream)


with torch.cuda.stream(s):
    for _ in range(3):
        [launch_ladder_only(i) for i in range(3)]
    ms_ladders = timed(lambda: [launch_ladder_only(i) for i in range(3)])
    step = lambda: [launch(i) for i in range(3)]
    for _ in range(3):
        step()
    ms = timed(step)
    per = [timed(lambda i=i: launch(i)) for i in range(3)]
# the host side once more, warm (the first pass above includes first-use costs: thread pool, page faults of fresh buffers)
t0 = time.perf_counter()
again = XE.book_batches(engine, terms)[6]
compile_warm_s = time.perf_counter() - t0
t0 = time.perf_counter()
book2 = _native.upload_many(ctx, again)
upload_warm_s = time.perf_counter() - t0
del book2
# VALUE / DELTA: the foreign leg in ONE launch on two curves (adr_price_xccy_foreign) next to the domestic launch
fused = {}
if not mask & 4:
    t0 = time.perf_counter()
    raw = XE.raw_from_terms(terms, engine.model.value_dt, xccy._dc_type)
    dom_b, for_b, _ = XE.compile_xccy_legs(raw, spot)
    legs_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    dom_tr, for_tr = _native.upload_many(ctx, [dom_b, for_b])
    legs_upload_s = time.perf_counter() - t0
    Pd, Pf, Px = dom_cur["dev"].n_pillars, for_cur["dev"].n_pillars, x_dev.n_pillars
    f_pv = torch.empty(n, dtype=torch.float64, device=dev)
    f_df = torch.empty((n, Pf), dtype=torch.float64, device=dev)
    f_dx = torch.empty((n, Px), dtype=torch.float64, device=dev)
    f_af = torch.empty(1 + Pf + Pf * Pf, dtype=torch.float64, device=dev)
    f_ax = torch.empty(1 + Px + Px * Px, dtype=torch.float64, device=dev)
    d_pv, d_de, _, d_ag = bufs[0]

    def two(per_swap=True):
        _native.price_dev(ctx, dom_cur["dev"], dom_tr, mask, d_pv.data_ptr() if per_swap else 0, d_de.data_ptr() if per_swap else 0, 0,
                          d_ag.data_ptr(), s.cuda_stream)
        _native.price_xccy_foreign_dev(ctx, for_cur["dev"], x_dev, for_tr, mask, f_pv.data_ptr() if per_swap else 0,
                                       f_df.data_ptr() if per_swap else 0, f_dx.data_ptr() if per_swap else 0,
                                       f_af.data_ptr(), f_ax.data_ptr(), s.cuda_stream)

    def foreign_only():
        _native.price_xccy_foreign_dev(ctx, for_cur["dev"], x_dev, for_tr, mask, f_pv.data_ptr(), f_df.data_ptr(), f_dx.data_ptr(),
                                       f_af.data_ptr(), f_ax.data_ptr(), s.cuda_stream)

    with torch.cuda.stream(s):
        for _ in range(3):
            two()
        fused = {"ms_two_launches": timed(two), "ms_foreign_leg_one_launch": timed(foreign_only),
                 "ms_two_launches_aggregate_only": timed(lambda: two(False)),
                 "host_terms_to_legs_s": legs_s, "host_legs_upload_s": legs_upload_s}
    # the foreign ladders of the two paths agree (three-batch: foreign rates on the OIS curve, flows on the XCCY curve)
    with torch.cuda.stream(s):
        step(); two()
    s.synchronize()
    ref_f, ref_x = bufs[1][1], bufs[2][1]
    fused["max_rel_diff_delta_foreign"] = float((f_df - ref_f).abs().max() / ref_f.abs().max())
    fused["max_rel_diff_delta_basis"] = float((f_dx[:, :ref_x.shape[1]] - ref_x).abs().max() / ref_x.abs().max())
pillars = [cur.n_pillars for _, cur in book]
out_bytes = 8 * n * sum(1 + P + (P * P if mask & 4 else 0) for P in pillars)
print(json.dumps({"workload": "distinct GBP/USD basis swaps, 1-30Y, annual / semi-annual foreign leg, two thirds seasoned",
                  "swaps": n, "mask": mask, "pillars": pillars, "cash_flows": flows, "ms": ms, "ms_aggregate_only": ms_ladders,
                  "swaps_per_s": n / ms * 1e3, "ms_domestic_foreignrates_foreignflows": per,
                  "output_GBps": out_bytes / ms / 1e6, "distinct_swaps": n,
                  "host_draw_terms_s": draw_s, "host_terms_to_batches_s": compile_s, "host_upload_s": upload_s,
                  "host_terms_to_batches_warm_s": compile_warm_s, "host_upload_warm_s": upload_warm_s,
                  "host_end_to_end_warm_ms": 1e3 * (compile_warm_s + upload_warm_s) + ms,
                  "market_build_s": market_s, **fused}))
