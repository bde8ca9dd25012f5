"""Seeded large-scale parity cases shared by the collected GPU tests (tests/test_gpu_sweeps.py) and the
stand-alone sweep scripts (tests/sweep_gpu_parity.py, tests/sweep_gpu_xccy.py): HIP path against the C oracle
(oracle/port.c) on the same curve tables.  Error metric as in tests/_parity.py."""
import numpy as np

from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades.compiler import OISTerms, compile_ois_terms
from adrates_amd.utils import BusDayAdjustTypes, CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes, InterpTypes
from oracle import port
from tests import _fixtures as F

SCHEMES = [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_FWD_RATES]


def batch_errors(got, ref, notional):
    """Worst per-trade error per output, both metrics of tests/_parity.py (ladder-relative with its floor, and
    per-unit-notional)."""
    N = np.abs(np.asarray(notional, dtype=np.float64))
    n = len(N)
    errs = {}
    for key, floor in (("pv", 1e-4), ("delta", 1e-8), ("gamma", 1e-12)):
        if got.get(key) is None or ref.get(key) is None:
            continue
        a, b = np.asarray(got[key]).reshape(n, -1), np.asarray(ref[key]).reshape(n, -1)
        diff = np.max(np.abs(a - b), axis=1)
        errs[key] = float(np.max(np.maximum(diff / np.maximum(np.max(np.abs(b), axis=1), floor * N),
                                            np.max(np.abs(a - b) / N[:, None] / np.maximum(1.0, np.abs(b) / N[:, None]), axis=1))))
    return errs


def ois_case(ctx, case, n):
    """Case number ``case`` of the OIS sweep: a random 32-pillar curve (level 0.6-8 %, tilted, humped, noisy), the
    interpolation scheme ``case % 3``, and ``n`` mixed trades - payment frequencies, float day counts, spreads,
    payment lags 0-2 days, seasoned / forward-starting / spot, pay / receive.  Returns a result dict (``skipped``
    set when the quotes bootstrap to a non-positive discount factor, which the library refuses)."""
    vd = F.README_VALUE_DT
    tenors = list(F.TENORS)
    rng = np.random.default_rng(9000 + case)
    interp = SCHEMES[case % 3]
    x = np.linspace(0.0, 1.0, len(tenors))
    level = rng.uniform(0.6, 8.0)
    px = np.maximum(level + rng.uniform(-0.4, 0.6) * level * x + rng.uniform(-0.2, 0.2) * level * np.sin(np.pi * x)
                    + rng.normal(0, 0.01, len(tenors)), 0.05)
    curve = F.gbp_model(vd, interp, px=list(px), tenors=tenors).curves.GBP_OIS_SONIA
    host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    out = {"case": case, "interp": interp.name, "level_pct": round(float(level), 2)}
    if not (np.all(np.isfinite(host.dfs)) and np.all(host.dfs > 0.0)):
        out["skipped"] = "non-positive DF"
        return out
    dc = _native.DeviceCurve(ctx, interp.value, host.times, host.dfs, host.jac, host.hess)
    starts = [vd, vd.add_months(-7), vd.add_years(-2), vd.add_months(5), vd.add_weekdays(2)]
    eff = [starts[i] for i in rng.choice(5, size=n, p=[0.5, 0.15, 0.1, 0.15, 0.1])]
    lfreq = [[FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL, FrequencyTypes.QUARTERLY][i]
             for i in rng.choice(3, size=n, p=[0.6, 0.25, 0.15])]
    ffreq = [[FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL][i] for i in rng.choice(2, size=n, p=[0.8, 0.2])]
    terms = OISTerms(effective_dt=eff, tenor=[f"{int(m)}M" for m in rng.integers(1, 361, n)],
                     coupon=rng.uniform(0.0, 0.09, n), notional=np.round(rng.uniform(1e5, 9e7, n), -4),
                     pay_fixed=rng.random(n) < 0.5, fixed_freq_type=ffreq, fixed_dc_type=DayCountTypes.ACT_365F,
                     floating_index=CurveTypes.GBP_OIS_SONIA, currency=CurrencyTypes.GBP, float_freq_type=lfreq,
                     float_dc_type=[[DayCountTypes.ACT_365F, DayCountTypes.ACT_360][i] for i in rng.integers(0, 2, n)],
                     float_spread=np.where(rng.random(n) < 0.3, rng.uniform(-0.002, 0.004, n), 0.0),
                     payment_lag=rng.choice([0, 0, 0, 1, 2], size=n), bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
    batch = compile_ois_terms(terms, vd)
    trades = _native.DeviceTrades(ctx, batch)
    try:
        got = _native.price(ctx, dc, trades, aggregate=True)
    finally:
        trades.close(); dc.close()
    ref = port.price(interp.value, host.times, host.dfs, host.jac, host.hess, batch)
    errs = batch_errors(got, ref, batch.notional)
    agg = float(np.max(np.abs(got["agg_gamma"] - ref["gamma"].sum(0))) / max(1e-30, np.max(np.abs(ref["gamma"].sum(0)))))
    out.update(trades=n, cash_flows=int(batch.flt_tp.size + batch.fix_tp.size), max_err=errs, agg_gamma_rel=agg,
               worst=max(errs.values()))
    return out


def xccy_book_case(ctx, n, seed=20240430):
    """The three trade batches of a synthetic book of ``n`` DISTINCT GBP/USD basis swaps
    (adrates_amd/trades/synthetic_xccy.py) priced by the HIP kernels and by the C oracle (oracle/port.c, including its
    per-coupon weights) on the same curve tables.  Returns one result dict per piece; ``judged`` is the per-unit-notional
    error for the domestic piece (its swaps are worth par: no spread to speak of, notional exchanged, so the ladder-
    relative error would compare rounding noise with itself) and the worse of both metrics for the other two."""
    from adrates_amd.market.position import xccy_engine as XE
    from adrates_amd.market.position.engine import Engine
    from adrates_amd.trades import synthetic_xccy as SX
    vd = F.README_VALUE_DT
    m = SX.build_market(vd, F.GBP_PX, F.USD_PX, F.TENORS)
    _native.set_default_context(ctx)           # the engine uploads the book's curves through this context
    engine = Engine(m)
    parts, spot = SX.synthesize_book(engine, vd, n, seed=seed)
    dom_model, for_model, xccy, dom_cur, for_cur, x_dev = XE._curves(engine, SX.template_swaps(vd)[:1])
    jac, hess = np.asarray(xccy._jac_basis), np.asarray(xccy._hess_basis)
    if jac.shape[1] % 2:                      # the engine pads an odd basis ladder (xccy_engine._xccy_device_curve)
        jac, hess = np.pad(jac, ((0, 0), (0, 1))), np.pad(hess, ((0, 0), (0, 1), (0, 1)))
    tables = [(dom_model._interp_type.value, dom_cur["host"].times, dom_cur["host"].dfs, dom_cur["host"].jac, dom_cur["host"].hess),
              (for_model._interp_type.value, for_cur["host"].times, for_cur["host"].dfs, for_cur["host"].jac, for_cur["host"].hess),
              (xccy._interp_type.value, np.asarray(xccy._times), np.asarray(xccy._dfs), jac, hess)]
    out = []
    for name, (batch, dev), tab in zip(("domestic", "foreign_rates", "foreign_flows"), parts, tables):
        ref = port.price(*tab, batch)
        trades = _native.DeviceTrades(ctx, batch)
        try:
            got = _native.price(ctx, dev, trades, aggregate=True)
        finally:
            trades.close()
        N = np.abs(batch.notional)
        unit, ladder = {}, {}
        for key, floor in (("pv", 1e-4), ("delta", 1e-8), ("gamma", 1e-12)):
            a, b = got[key].reshape(n, -1), ref[key].reshape(n, -1)
            unit[key] = float(np.max(np.abs(a - b) / N[:, None] / np.maximum(1.0, np.abs(b) / N[:, None])))
            ladder[key] = float(np.max(np.max(np.abs(a - b), axis=1) / np.maximum(np.max(np.abs(b), axis=1), floor * N)))
        judged = max(unit.values()) if name == "domestic" else max(max(unit.values()), max(ladder.values()))
        agg_rel = float(np.max(np.abs(got["agg_gamma"] - ref["gamma"].sum(0))) / max(1e-30, np.max(np.abs(ref["gamma"].sum(0)))))
        out.append({"piece": name, "swaps": n, "cash_flows": int(batch.flt_tp.size + batch.fix_tp.size),
                    "pillars": int(dev.n_pillars), "per_unit_notional_err": unit, "ladder_rel_err": ladder,
                    "judged": judged, "agg_gamma_rel": agg_rel})
    return out
