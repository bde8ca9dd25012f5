"""The launch plan of adr_price_dev (adrates_amd/csrc/route.hpp), enumerated on the CPU: for the cross product of trade
classes {plain, long, very long, payment lag, long payment lag, very long payment lag, weighted} x pillar counts {32, 31,
17, 40, 40 on tiles, 64, 96 (three tiles)} x the three interpolation schemes x requests {V, VD, VDG} x outputs {per trade, per trade + aggregate,
aggregate only}, every trade of a mixed batch is priced by exactly one launch.  The reference has a single route
(Engine._compute_ois_natural, cavour/market/position/engine.py:153-215); here seven kernel families share the work, and
every new route so far had cost a correctness fix in the routing - this test walks the table without a GPU."""
import numpy as np
import pytest

from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades.compiler import OISTerms, compile_ois_terms
from adrates_amd.utils import BusDayAdjustTypes, CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes, InterpTypes

from . import _fixtures as F


def _batch(vd, classes):
    """A few trades of each requested class; returns (batch, class label per trade)."""
    spec = {          # class -> (float frequency, tenors in months, payment lag)
        "plain": (FrequencyTypes.ANNUAL, [7, 60, 133, 360], 0),
        "long": (FrequencyTypes.QUARTERLY, [130, 240, 360], 0),                  # 44-120 coupons: chained rows
        "very_long": (FrequencyTypes.MONTHLY, [400, 480], 0),                    # > 384 coupons: general kernel
        "lag": (FrequencyTypes.ANNUAL, [9, 48, 200, 360], 2),
        "long_lag": (FrequencyTypes.QUARTERLY, [150, 300], 2),                   # 50-100 coupons: chained payment-lag rows
        "very_long_lag": (FrequencyTypes.MONTHLY, [200, 360], 2),                # > 128 coupons with lag: the rest list
    }
    tenors, freqs, lags, labels = [], [], [], []
    for c in classes:
        if c == "weighted":
            continue
        f, months, lag = spec[c]
        for m in months:
            tenors.append(f"{m}M"); freqs.append(f); lags.append(lag); labels.append(c)
    n_w = 3 if "weighted" in classes else 0
    for m in (30, 96, 250)[:n_w]:
        tenors.append(f"{m}M"); freqs.append(FrequencyTypes.SEMI_ANNUAL); lags.append(0); labels.append("weighted")
    n = len(tenors)
    terms = OISTerms(effective_dt=vd, tenor=tenors, coupon=np.full(n, 0.04), notional=np.full(n, 1e7), pay_fixed=np.arange(n) % 2 == 0,
                     fixed_freq_type=FrequencyTypes.ANNUAL, fixed_dc_type=DayCountTypes.ACT_365F, floating_index=CurveTypes.GBP_OIS_SONIA,
                     currency=CurrencyTypes.GBP, float_freq_type=freqs, float_dc_type=DayCountTypes.ACT_365F, payment_lag=lags,
                     bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
    batch = compile_ois_terms(terms, vd)
    if n_w:
        w = np.ones(batch.flt_tp.shape[0])
        for t in range(n - n_w, n):
            w[batch.flt_off[t]:batch.flt_off[t + 1]] = 0.97          # per-coupon notional multipliers (the XCCY foreign leg)
        batch.flt_weight = w
    return batch, labels


def _curves(vd):
    from .test_gpu_many_pillars import forty_pillar_quotes
    px40, t40 = forty_pillar_quotes()
    years = lambda s: float(s[:-1]) * {"D": 1 / 365, "W": 7 / 365, "M": 1 / 12, "Y": 1.0}[s[-1]]
    extra = [f"{y}Y" for y in range(1, 50) if f"{y}Y" not in F.TENORS]
    t64 = sorted(list(F.TENORS) + extra, key=years)[:64]
    base_t = [years(t) for t in F.TENORS]
    px64 = [float(np.interp(years(t), base_t, F.GBP_PX)) if t not in F.TENORS else F.GBP_PX[F.TENORS.index(t)] for t in t64]
    from .test_gpu_many_pillars import many_pillar_quotes
    px96, t96 = many_pillar_quotes(96)
    sets = {32: (None, None), 96: (px96, t96), 31: (list(F.GBP_PX[:13]) + list(F.GBP_PX[14:]), list(F.TENORS[:13]) + list(F.TENORS[14:])),
            17: (list(F.GBP_PX[8:9] + F.GBP_PX[14:30]), list(F.TENORS[8:9] + F.TENORS[14:30])), 40: (px40, t40), 64: (px64, t64)}
    out = {}
    for P, (px, tenors) in sets.items():
        curve = (F.gbp_model(vd) if px is None else F.gbp_model(vd, px=px, tenors=tenors)).curves.GBP_OIS_SONIA
        host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
        assert host.n_pillars == P
        out[P] = host
    return out


ALL = ("plain", "long", "very_long", "lag", "long_lag", "very_long_lag", "weighted")


def test_every_trade_is_priced_exactly_once_over_the_route_table():
    vd = F.README_VALUE_DT
    curves = _curves(vd)
    mixes = [ALL, ("plain",), ("lag",), ("long",), ("long_lag",), ("weighted",), ("plain", "very_long_lag"), ("very_long",)]
    seen = set()
    for classes in mixes:
        batch, labels = _batch(vd, classes)
        for P, host in curves.items():
            for flags in ((0, _native.DeviceCurve.PILLAR_TILES) if P == 40 else (0,)):
                for interp in (InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_FWD_RATES):
                    for mask in (1, 3, 7):
                        for per_trade, aggregate in ((True, False), (True, True), (False, True)):
                            launches, cover = _native.route_host(interp.value, host.times, host.dfs, host.jac, host.hess, batch, mask,
                                                                 per_trade=per_trade, aggregate=aggregate, curve_flags=flags)
                            bad = [(labels[i], int(c)) for i, c in enumerate(cover) if c != 1]
                            assert not bad, (classes, P, flags, interp.name, mask, per_trade, aggregate, launches, bad)
                            seen.update(f for f, *_ in launches)
    assert seen == set(_native.ROUTE_FAMILIES), seen          # the table exercised every kernel family


def test_the_plans_of_the_reported_configurations():
    """The routes DESIGN.md section 5 states for the benchmark configurations."""
    vd = F.README_VALUE_DT
    host = _curves(vd)[32]
    plain, _ = _batch(vd, ("plain",))
    fam = lambda launches: [(f, s) for f, s, *_ in launches]
    r = lambda b, mask, **kw: fam(_native.route_host(4, host.times, host.dfs, host.jac, host.hess, b, mask, **kw)[0])
    assert r(plain, 7, aggregate=True) == [("fast", "rows")]                       # BASELINE configs[2]: the bench kernel
    assert r(plain, 3, aggregate=True) == [("lite", "lite")]                       # configs[1]
    assert r(plain, 7, per_trade=False, aggregate=True) == [("knot", "lite")]      # Portfolio.compute: the ladder alone
    lag, _ = _batch(vd, ("lag",))
    assert r(lag, 7) == [("fast_lag", "lagged")] and r(lag, 3) == [("lite_lag", "lite_lag")]
    assert fam(_native.route_host(2, host.times, host.dfs, host.jac, host.hess, lag, 7)[0]) == [("general", "general")]   # LINEAR_FWD_RATES
    mixed, _ = _batch(vd, ALL)
    assert r(mixed, 7, per_trade=False, aggregate=True)[-2:] == [("knot", "lite"), ("knot_lag", "lite_lag")]    # the projections add last
    assert r(lag, 7, per_trade=False, aggregate=True) == [("knot_lag", "lite_lag")]
