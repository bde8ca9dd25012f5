"""XCCY basis swap analytics through the HIP kernels against the torch-autodiff restatement of
Engine._compute_xccy (oracle/xccy_oracle.py::xccy_analytics; cavour/market/position/engine.py:1411-1988)."""
import numpy as np
import pytest

from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades.rates.xccy_basis_swap import XccyBasisSwap
from adrates_amd.utils import (CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes, InterpTypes, RequestTypes,
                               SwapTypes)
from adrates_amd.utils.helpers import times_from_dates
from oracle import xccy_oracle as XO
from tests.test_xccy_curve import BASIS, GBP, SPOT, TENORS, USD, VALUE_DT, _basis_swaps, _ois_curves

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _model(interp=InterpTypes.FLAT_FWD_RATES):
    m, gbp, usd = _ois_curves()
    m.build_xccy_curve(name="USD_GBP_BASIS", domestic_curve_name="GBP_OIS_SONIA", foreign_curve_name="USD_OIS_SOFR",
                       basis_spreads=[b * 1e4 for b in BASIS], tenor_list=TENORS, spot_fx=SPOT,
                       domestic_dc_type=DayCountTypes.ACT_365F, foreign_dc_type=DayCountTypes.ACT_360,
                       interp_type=interp)
    return m


def _cache(curve):
    h = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    return dict(times=h.times, dfs=h.dfs, jac=h.jac, hess=h.hess)


def _swap(tenor, spread, lag=0, effective=VALUE_DT, freq=FrequencyTypes.ANNUAL, notional=1_000_000):
    return XccyBasisSwap(effective_dt=effective, term_dt_or_tenor=tenor, domestic_payment_lag=lag, foreign_payment_lag=lag,
                         domestic_notional=SPOT * notional, foreign_notional=notional, domestic_spread=0.0005,
                         foreign_spread=spread, domestic_freq_type=FrequencyTypes.ANNUAL, foreign_freq_type=freq,
                         domestic_dc_type=DayCountTypes.ACT_365F, foreign_dc_type=DayCountTypes.ACT_360,
                         domestic_floating_index=CurveTypes.GBP_OIS_SONIA,
                         foreign_floating_index=CurveTypes.USD_OIS_SOFR, domestic_currency=CurrencyTypes.GBP,
                         foreign_currency=CurrencyTypes.USD)


_FLOOR = {1.0: 1e-4, 1e-4: 1e-8, 1e-6: 1e-12}


def _close(got, want, scale):
    """The parity metric of tests/_parity.py (`ladder_err`): max |a - b| relative to the ladder's own largest entry,
    with a floor per unit notional (1e-4 N for PV, 1e-8 N for delta, 1e-12 N for gamma) below which a ladder counts as
    rounding noise.  ``scale`` = notional x {1, 1e-4, 1e-6} names the kind (PV, delta, gamma) as before.  The oracle's
    curve Jacobians / Hessians are the product's own (`build_engine_curve`), which tests/test_curve_tables.py checks
    against autodiff of the bootstrap separately."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    kind = min(_FLOOR, key=lambda k: abs(np.log(max(scale, 1e-300)) - np.log(k * max(_close.notional, 1.0))))
    floor = _FLOOR[kind] * _close.notional
    assert np.max(np.abs(got - want)) <= TOL * max(float(np.max(np.abs(want))), floor), \
        (float(np.max(np.abs(got - want))), float(np.max(np.abs(want))), floor)


_close.notional = 1.0


@pytest.mark.parametrize("case", ["par_5y", "off_market_7y_lagged", "semi_annual_10y", "forward_start", "seasoned"])
def test_basis_swap_value_delta_gamma(case):
    m = _model()
    if case == "par_5y":
        swap = _swap("5Y", 0.0034)
    elif case == "off_market_7y_lagged":
        swap = _swap("7Y", 0.0060, lag=2, notional=25_000_000)
    elif case == "semi_annual_10y":
        swap = _swap("10Y", 0.0030, freq=FrequencyTypes.SEMI_ANNUAL)
    elif case == "forward_start":
        swap = _swap("4Y", 0.0040, effective=VALUE_DT.add_months(9))
    else:
        swap = _swap("6Y", 0.0035, effective=VALUE_DT.add_months(-8))
    res = swap.position(m).compute([RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA])
    gbp, usd, x = m.curves.GBP_OIS_SONIA, m.curves.USD_OIS_SOFR, m.curves.USD_GBP_BASIS
    want = XO.xccy_analytics(swap, VALUE_DT, _cache(gbp), gbp._interp_type.value, _cache(usd), usd._interp_type.value,
                             x, times_from_dates)
    scale = _close.notional = abs(swap._domestic_leg._notional)
    _close(res.value.amount, want["value"], scale)
    curves = (CurveTypes.GBP_OIS_SONIA, CurveTypes.USD_OIS_SOFR, CurveTypes.USD_GBP_BASIS)
    for curve, key in zip(curves, ("delta_dom", "delta_for", "delta_basis")):
        _close(res.risk(curve).risk_ladder, want[key], scale * 1e-4)
        assert np.any(res.risk(curve).risk_ladder != 0.0)
    for curve, key in zip(curves, ("gamma_dom", "gamma_for", "gamma_basis")):
        _close(res.gamma(curve).risk_ladder, want[key], scale * 1e-6)
    # cross-gamma foreign OIS x basis: the mixed term through the two curves' knot DFs (xccy_engine.cross_gamma_for_basis)
    cross = res.gamma.cross_gamma(CurveTypes.USD_OIS_SOFR, CurveTypes.USD_GBP_BASIS)
    assert cross.risk_matrix.shape == (len(usd.swap_times), len(x.swap_times)) and np.any(cross.risk_matrix != 0.0)
    _close(cross.risk_matrix, want["cross_for_basis"], scale * 1e-6)
    assert cross.curve_type_1 == CurveTypes.USD_OIS_SOFR and cross.currency == CurrencyTypes.GBP


def test_value_only_and_missing_curve():
    m = _model()
    swap = _swap("3Y", 0.0030)
    res = swap.position(m).compute([RequestTypes.VALUE])
    assert res.risk is None and res.gamma is None and np.isfinite(res.value.amount)
    bare, _, _ = _ois_curves()
    with pytest.raises(Exception, match="BASIS"):
        swap.position(bare).compute([RequestTypes.VALUE])


@pytest.mark.parametrize("interp", [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.LINEAR_FWD_RATES])
def test_other_interpolation_schemes_on_all_three_curves(interp):
    """All three curves on LINEAR_ZERO_RATES / LINEAR_FWD_RATES (interpolator_ad.py:227-235)."""
    m, _, _ = _ois_curves()
    for name in ("GBP_OIS_SONIA", "USD_OIS_SOFR"):
        getattr(m.curves, name)._interp_type = interp
    m.build_xccy_curve(name="USD_GBP_BASIS", domestic_curve_name="GBP_OIS_SONIA", foreign_curve_name="USD_OIS_SOFR",
                       basis_spreads=[b * 1e4 for b in BASIS], tenor_list=TENORS, spot_fx=SPOT,
                       domestic_dc_type=DayCountTypes.ACT_365F, foreign_dc_type=DayCountTypes.ACT_360,
                       interp_type=interp)
    swap = _swap("8Y", 0.0045, effective=VALUE_DT.add_months(-5), freq=FrequencyTypes.SEMI_ANNUAL)
    res = swap.position(m).compute([RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA])
    gbp, usd, x = m.curves.GBP_OIS_SONIA, m.curves.USD_OIS_SOFR, m.curves.USD_GBP_BASIS
    want = XO.xccy_analytics(swap, VALUE_DT, _cache(gbp), gbp._interp_type.value, _cache(usd), usd._interp_type.value,
                             x, times_from_dates)
    scale = _close.notional = abs(swap._domestic_leg._notional)
    _close(res.value.amount, want["value"], scale)
    for curve, d, g in ((CurveTypes.GBP_OIS_SONIA, "delta_dom", "gamma_dom"), (CurveTypes.USD_OIS_SOFR, "delta_for", "gamma_for"),
                        (CurveTypes.USD_GBP_BASIS, "delta_basis", "gamma_basis")):
        _close(res.risk(curve).risk_ladder, want[d], scale * 1e-4)
        _close(res.gamma(curve).risk_ladder, want[g], scale * 1e-6)


def test_book_in_three_launches_per_trade_and_aggregate():
    from adrates_amd.market.position.engine import Engine
    from adrates_amd.market.position.xccy_engine import price_xccy_batch
    m = _model()
    book = [_swap("5Y", 0.0034), _swap("7Y", 0.0060, lag=2, notional=25_000_000),
            _swap("10Y", 0.0030, freq=FrequencyTypes.SEMI_ANNUAL), _swap("4Y", 0.0040, effective=VALUE_DT.add_months(9)),
            _swap("6Y", 0.0035, effective=VALUE_DT.add_months(-8)), _swap("20Y", 0.0045, freq=FrequencyTypes.QUARTERLY),
            _swap("1Y", 0.0025)]
    out = price_xccy_batch(Engine(m), book, {RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA},
                           per_trade=True, aggregate=True)
    gbp, usd, x = m.curves.GBP_OIS_SONIA, m.curves.USD_OIS_SOFR, m.curves.USD_GBP_BASIS
    wants = [XO.xccy_analytics(s, VALUE_DT, _cache(gbp), gbp._interp_type.value, _cache(usd), usd._interp_type.value,
                               x, times_from_dates) for s in book]
    for i, (s, w) in enumerate(zip(book, wants)):
        scale = _close.notional = abs(s._domestic_leg._notional)
        _close(out["pv"][i], w["value"], scale)
        for k in ("delta_dom", "delta_for", "delta_basis"):
            _close(out[k][i], w[k], scale * 1e-4)
        for k in ("gamma_dom", "gamma_for", "gamma_basis"):
            _close(out[k][i], w[k], scale * 1e-6)
    total = _close.notional = sum(abs(s._domestic_leg._notional) for s in book)
    _close(out["agg_pv"], sum(w["value"] for w in wants), total)
    for k in ("delta_dom", "delta_for", "delta_basis"):
        _close(out["agg_" + k], sum(w[k] for w in wants), total * 1e-4)
    for k in ("gamma_dom", "gamma_for", "gamma_basis"):
        _close(out["agg_" + k], sum(w[k] for w in wants), total * 1e-6)
    # delta-only request of BASELINE.json's config 3: same ladders, no gamma keys
    d = price_xccy_batch(Engine(m), book, {RequestTypes.DELTA}, per_trade=True)
    assert "gamma_dom" not in d and "pv" not in d
    for k in ("delta_dom", "delta_for", "delta_basis"):
        assert np.allclose(d[k], out[k], rtol=1e-12, atol=1e-9)


def test_foreign_leg_in_one_launch_on_two_curves(monkeypatch):
    """VALUE / DELTA requests price the foreign leg in one launch that looks D_x(tp), D_f(ts), D_f(te) up itself
    (adr_price_xccy_foreign; xccy_engine._price_fused): against the oracle per swap, against the three-batch assembly, per
    swap and as book sums, on the hand-made book and on a drawn one."""
    from adrates_amd.market.position import xccy_engine as XE
    from adrates_amd.market.position.engine import Engine
    from adrates_amd.trades import synthetic_xccy as SX
    m = _model()
    book = [_swap("5Y", 0.0034), _swap("7Y", 0.0060, lag=2, notional=25_000_000),
            _swap("10Y", 0.0030, freq=FrequencyTypes.SEMI_ANNUAL), _swap("4Y", 0.0040, effective=VALUE_DT.add_months(9)),
            _swap("6Y", 0.0035, effective=VALUE_DT.add_months(-8)), _swap("20Y", 0.0045, freq=FrequencyTypes.QUARTERLY),
            _swap("1Y", 0.0025)]
    reqs = {RequestTypes.VALUE, RequestTypes.DELTA}
    calls = []
    real = XE._native.price_xccy_foreign
    monkeypatch.setattr(XE._native, "price_xccy_foreign", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    assert XE.FUSED_FOREIGN_LEG
    one = XE.price_xccy_batch(Engine(m), book, reqs, per_trade=True, aggregate=True)
    assert calls, "the one-launch path was not taken"
    monkeypatch.setattr(XE, "FUSED_FOREIGN_LEG", False)
    three = XE.price_xccy_batch(Engine(m), book, reqs, per_trade=True, aggregate=True)
    gbp, usd, x = m.curves.GBP_OIS_SONIA, m.curves.USD_OIS_SOFR, m.curves.USD_GBP_BASIS
    for i, s in enumerate(book):
        w = XO.xccy_analytics(s, VALUE_DT, _cache(gbp), gbp._interp_type.value, _cache(usd), usd._interp_type.value,
                              x, times_from_dates)
        scale = _close.notional = abs(s._domestic_leg._notional)
        _close(one["pv"][i], w["value"], scale)
        for k in ("delta_dom", "delta_for", "delta_basis"):
            _close(one[k][i], w[k], scale * 1e-4)
            assert np.any(one[k][i] != 0.0)
    total = _close.notional = sum(abs(s._domestic_leg._notional) for s in book)
    _close(one["agg_pv"], three["agg_pv"], total)
    _close(one["agg_pv"], float(np.sum(one["pv"])), total)
    for k in ("delta_dom", "delta_for", "delta_basis"):
        _close(one[k], three[k], total * 1e-4)
        _close(one["agg_" + k], three["agg_" + k], total * 1e-4)
        _close(one["agg_" + k], one[k].sum(0), total * 1e-4)
    # a drawn book (distinct swaps, 1-30Y, annual / semi-annual foreign leg, two thirds seasoned), book sums only as well
    terms, _ = SX.draw_terms(VALUE_DT, 3000)
    three = XE.price_xccy_batch(Engine(m), terms, reqs, per_trade=True, aggregate=True)
    monkeypatch.setattr(XE, "FUSED_FOREIGN_LEG", True)
    del calls[:]
    one = XE.price_xccy_batch(Engine(m), terms, reqs, per_trade=True, aggregate=True)
    sums = XE.price_xccy_batch(Engine(m), terms, reqs, per_trade=False, aggregate=True)
    assert len(calls) == 2
    big = _close.notional = float(np.max(np.abs(three["pv"]))) + 1e6
    _close(one["pv"], three["pv"], big)
    for k in ("delta_dom", "delta_for", "delta_basis"):
        _close(one[k], three[k], big * 1e-4)
        tot = _close.notional = float(np.sum(np.abs(three[k]))) * 1e4 + 1e6
        _close(one["agg_" + k], three["agg_" + k], tot * 1e-4)
        _close(sums["agg_" + k], three["agg_" + k], tot * 1e-4)
    _close.notional = float(np.sum(np.abs(three["pv"]))) + 1e6
    _close(sums["agg_pv"], three["agg_pv"], _close.notional)
    _close(one["agg_pv"], three["agg_pv"], _close.notional)


def test_two_curve_entry_refuses_what_it_does_not_take(monkeypatch):
    """adr_price_xccy_foreign answers ADR_ERR_UNSUPPORTED (-2) for books outside its scope - legs without a coupon whose
    accrual end differs from its payment time, a foreign curve on LINEAR_FWD_RATES with an XCCY curve on a log-linear scheme (or
    the other way round) - and the engine then prices the three batches (`_price_fused` returns None); with both curves on
    LINEAR_FWD_RATES the launch takes the book (a factor's effective log weights, kernels_lite.hip `factor`); a curve of another
    context is ADR_ERR_INVALID."""
    from adrates_amd import _native
    from adrates_amd.market.position import xccy_engine as XE
    from adrates_amd.market.position.engine import Engine
    from adrates_amd.trades import synthetic
    from adrates_amd.utils.error import LibError
    m = _model()
    engine = Engine(m)
    dom_model, for_model, xccy, dom_cur, for_cur, x_dev = XE._curves(engine, [_swap("5Y", 0.0034)])
    ctx = dom_cur["ctx"]
    plain = _native.DeviceTrades(ctx, synthetic.synthesize(VALUE_DT, 64, seed=2))          # OIS without payment lag
    with pytest.raises(LibError, match=r"\(-2\)"):
        _native.price_xccy_foreign(ctx, for_cur["dev"], x_dev, plain)
    plain.close()
    swap = _swap("6Y", 0.0041, freq=FrequencyTypes.SEMI_ANNUAL)
    for x_interp, fused in ((InterpTypes.LINEAR_FWD_RATES, True), (InterpTypes.FLAT_FWD_RATES, False)):
        m2, _, _ = _ois_curves()
        for name in ("GBP_OIS_SONIA", "USD_OIS_SOFR"):
            getattr(m2.curves, name)._interp_type = InterpTypes.LINEAR_FWD_RATES
        m2.build_xccy_curve(name="USD_GBP_BASIS", domestic_curve_name="GBP_OIS_SONIA", foreign_curve_name="USD_OIS_SOFR",
                            basis_spreads=[b * 1e4 for b in BASIS], tenor_list=TENORS, spot_fx=SPOT,
                            domestic_dc_type=DayCountTypes.ACT_365F, foreign_dc_type=DayCountTypes.ACT_360,
                            interp_type=x_interp)
        assert (XE._price_fused(Engine(m2), [swap], True, True, True, False) is not None) == fused
        res = swap.position(m2).compute([RequestTypes.VALUE, RequestTypes.DELTA])
        gbp, usd, x = m2.curves.GBP_OIS_SONIA, m2.curves.USD_OIS_SOFR, m2.curves.USD_GBP_BASIS
        want = XO.xccy_analytics(swap, VALUE_DT, _cache(gbp), gbp._interp_type.value, _cache(usd), usd._interp_type.value,
                                 x, times_from_dates)
        scale = _close.notional = abs(swap._domestic_leg._notional)
        _close(res.value.amount, want["value"], scale)
        _close(res.risk(CurveTypes.GBP_OIS_SONIA).risk_ladder, want["delta_dom"], scale * 1e-4)
        _close(res.risk(CurveTypes.USD_OIS_SOFR).risk_ladder, want["delta_for"], scale * 1e-4)
        _close(res.risk(CurveTypes.USD_GBP_BASIS).risk_ladder, want["delta_basis"], scale * 1e-4)
    other = _native.Context(0)
    with pytest.raises(LibError):
        _native.price_xccy_foreign(other, for_cur["dev"], x_dev, _native.DeviceTrades(other, synthetic.synthesize(VALUE_DT, 8, seed=3)))


def test_weighted_coupons_vs_c_oracle():
    """adr_trades_upload_weighted on a mixed batch: weighted trades (with and without payment lag) go to the
    general kernel, weight-1 trades of the same batch stay on the fast kernel."""
    from adrates_amd import _native
    from adrates_amd.trades.compiler import OISTerms, compile_ois_terms
    from adrates_amd.utils import BusDayAdjustTypes
    from oracle import port
    m, gbp, _ = _ois_curves()
    host = build_engine_curve(gbp.swap_rates, gbp.swap_times, gbp.year_fracs)
    ctx = _native.default_context()
    dc = _native.DeviceCurve(ctx, gbp._interp_type.value, host.times, host.dfs, host.jac, host.hess)
    rng = np.random.default_rng(21)
    n = 3000
    terms = OISTerms(effective_dt=VALUE_DT, tenor=[f"{int(k)}M" for k in rng.integers(1, 241, n)],
                     coupon=rng.uniform(0.01, 0.07, n), notional=np.round(rng.uniform(1e6, 5e7, n), -5),
                     pay_fixed=rng.random(n) < 0.5, fixed_freq_type=FrequencyTypes.ANNUAL,
                     fixed_dc_type=DayCountTypes.ACT_365F, floating_index=CurveTypes.GBP_OIS_SONIA,
                     currency=CurrencyTypes.GBP, float_freq_type=FrequencyTypes.SEMI_ANNUAL,
                     float_dc_type=DayCountTypes.ACT_365F, float_spread=np.where(rng.random(n) < 0.3, 0.0015, 0.0),
                     payment_lag=rng.choice([0, 2], size=n), bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
    batch = compile_ois_terms(terms, VALUE_DT)
    w = rng.uniform(0.2, 1.3, batch.flt_tp.size)
    plain = np.repeat(rng.random(n) < 0.4, np.diff(batch.flt_off))      # 40 % of the trades keep weight 1
    w[plain] = 1.0
    batch.flt_weight = w
    trades = _native.DeviceTrades(ctx, batch)
    got = _native.price(ctx, dc, trades, aggregate=True)
    ref = port.price(gbp._interp_type.value, host.times, host.dfs, host.jac, host.hess, batch)
    from tests._parity import assert_batch_parity
    assert_batch_parity(got, ref, batch.notional)
    assert np.allclose(got["agg_delta"], ref["delta"].sum(0), rtol=1e-10, atol=1e-6)
    # the weights matter: the same batch without them prices differently
    batch.flt_weight = None
    unweighted = port.price(gbp._interp_type.value, host.times, host.dfs, host.jac, host.hess, batch)
    assert np.max(np.abs(unweighted["pv"] - ref["pv"])) > 1.0
    trades.close(); dc.close()


def test_ois_with_cross_currency_collateral():
    """Engine._compute_ois_xccy_collateral (engine.py:217-503): the cases of the CPU host test on the kernels."""
    from tests.test_xccy_engine_host import _collateral_model, check_ois_collateral
    check_ois_collateral(_collateral_model())


def test_matured_and_maturing_swaps():
    from tests.test_xccy_engine_host import check_matured_and_maturing
    check_matured_and_maturing(_model())
