"""Host side of the cross-currency assembly (market/position/xccy_engine.py) without a GPU: the three trade
batches it compiles are priced by the C restatement (oracle/port.c) standing in for the device calls, and the
assembled ladders are compared with the torch-autodiff restatement of Engine._compute_xccy.  The same cases run
through the HIP kernels in tests/test_gpu_xccy.py."""
import numpy as np
import pytest

from adrates_amd import _native
from adrates_amd.market.position import xccy_engine
from adrates_amd.utils import CurveTypes, FrequencyTypes, RequestTypes
from adrates_amd.utils.helpers import times_from_dates
from oracle import port
from oracle import xccy_oracle as XO
from tests.test_gpu_xccy import VALUE_DT, _cache, _model, _swap

CURVES = (CurveTypes.GBP_OIS_SONIA, CurveTypes.USD_OIS_SOFR, CurveTypes.USD_GBP_BASIS)


class _HostCurve:
    def __init__(self, ctx, method, times, dfs, jac, hess=None):
        self.args = (method, np.asarray(times, float), np.asarray(dfs, float), np.asarray(jac, float),
                     None if hess is None else np.asarray(hess, float))
        self.n_pillars = np.asarray(jac).shape[1]


class _HostTrades:
    def __init__(self, ctx, batch):
        self.batch, self.n_trades = batch, batch.n_trades

    def close(self):
        pass


def _host_price(ctx, curve, trades, want_value=True, want_delta=True, want_gamma=True, per_trade=True, aggregate=False):
    r = port.price(*curve.args, trades.batch, want_delta=want_delta, want_gamma=want_gamma)
    out = dict(r) if per_trade else {}
    if aggregate:
        out.update(agg_pv=float(r["pv"].sum()), agg_delta=r["delta"].sum(0),
                   agg_gamma=None if r["gamma"] is None else r["gamma"].sum(0))
    return out


def _host_curve_df(ctx, curve, t):
    """Stand-in for adr_curve_df: the oracle's restatement of InterpolatorAd.simple_interpolate on the same tables."""
    from oracle import cavour_oracle as O
    method, times, dfs = curve.args[0], curve.args[1], curve.args[2]
    out = O.simple_interpolate(np.asarray(t, dtype=np.float64), times, dfs, method).numpy()
    return float(out) if np.ndim(t) == 0 else out


def _host_no_two_curve_launch(ctx, *a, **k):
    """The two-curve foreign-leg launch (adr_price_xccy_foreign) has no C restatement: the stand-in answers as the library
    does for a book the launch does not take (ADR_ERR_UNSUPPORTED), so these tests run the three-batch assembly - the
    fallback of the product path.  The one-launch path is compared with it and with the oracle on the GPU
    (tests/test_gpu_xccy.py)."""
    from adrates_amd.utils.error import LibError
    raise LibError("adr_price_xccy_foreign failed (-2): no two-curve launch on the host stand-in")


def _host_two_curve_launch(ctx, for_curve, x_curve, legs, want_value=True, want_delta=True, per_trade=True, aggregate=False):
    """Stand-in for adr_price_xccy_foreign: the formula of the two-curve kernel (kernels_lite.hip, XC; engine.py:1640-1733) on
    torch tensors - per coupon  s N ((D_f(ts) / D_f(te) - 1) + spread alpha) D_x(tp), the exchanges as flows on the XCCY curve -
    differentiated w.r.t. the two curves' knot DFs and chained with their Jacobians.  Foreign currency, per bp."""
    import torch
    from oracle import cavour_oracle as O
    b = legs.batch
    n = b.n_trades
    (mf, tf, dfs_f, jac_f), (mx, tx, dfs_x, jac_x) = for_curve.args[:4], x_curve.args[:4]
    if (mf == 2) != (mx == 2):
        return _host_no_two_curve_launch(ctx)
    d_f = torch.tensor(dfs_f, dtype=torch.float64, requires_grad=True)
    d_x = torch.tensor(dfs_x, dtype=torch.float64, requires_grad=True)
    T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64))
    owner = torch.as_tensor(np.repeat(np.arange(n), np.diff(b.flt_off)))
    fown = torch.as_tensor(np.repeat(np.arange(n), np.diff(b.fix_off)))
    live, acc = torch.as_tensor(b.flt_tp >= 0.0), torch.as_tensor(b.flt_alpha > 0.0)
    Dx = O.simple_interpolate(b.flt_tp, tx, d_x, mx)
    R = torch.where(acc, O.simple_interpolate(b.flt_ts, tf, d_f, mf) / O.simple_interpolate(b.flt_te, tf, d_f, mf), torch.ones_like(Dx))
    w = T(np.ones_like(b.flt_tp) if b.flt_weight is None else b.flt_weight)
    amount = (T(b.flt_sign) * T(b.notional))[owner] * w * ((R - 1.0) + T(b.spread)[owner] * T(b.flt_alpha))
    pv = torch.zeros(n, dtype=torch.float64).index_add(0, owner, torch.where(live, amount * Dx, torch.zeros_like(Dx)))
    if b.fix_tp.size:
        Df = O.simple_interpolate(b.fix_tp, tx, d_x, mx)
        flows = T(b.fix_sign)[fown] * T(b.fix_pay) * Df
        pv = pv.index_add(0, fown, torch.where(torch.as_tensor(b.fix_tp > 0.0), flows, torch.zeros_like(flows)))
    Pf, Px = np.asarray(jac_f).shape[1], np.asarray(jac_x).shape[1]
    de_f, de_x = np.zeros((n, Pf)), np.zeros((n, Px))
    for i in range(n if want_delta else 0):
        g_f, g_x = torch.autograd.grad(pv[i], (d_f, d_x), retain_graph=True, allow_unused=True)
        de_f[i] = 1e-4 * np.asarray(jac_f).T @ (np.zeros(len(dfs_f)) if g_f is None else g_f.numpy())
        de_x[i] = 1e-4 * np.asarray(jac_x).T @ (np.zeros(len(dfs_x)) if g_x is None else g_x.numpy())
    pv = pv.detach().numpy()
    out = {}
    if per_trade and want_value:
        out["pv"] = pv
    if per_trade and want_delta:
        out["delta_foreign"], out["delta_basis"] = de_f, de_x
    if aggregate:
        out.update(agg_pv=float(pv.sum()), agg_delta_foreign=de_f.sum(0), agg_delta_basis=de_x.sum(0))
    return out


@pytest.fixture()
def host_engine(monkeypatch):
    monkeypatch.setattr(_native, "price_xccy_foreign", _host_no_two_curve_launch)
    monkeypatch.setattr(_native, "curve_df", _host_curve_df)
    monkeypatch.setattr(_native, "DeviceCurve", _HostCurve)
    monkeypatch.setattr(_native, "DeviceTrades", _HostTrades)
    monkeypatch.setattr(_native, "price", _host_price)
    monkeypatch.setattr(_native, "default_context", lambda: None)
    return _model()


def _book():
    return [_swap("5Y", 0.0034), _swap("7Y", 0.0060, lag=2, notional=25_000_000),
            _swap("10Y", 0.0030, freq=FrequencyTypes.SEMI_ANNUAL), _swap("4Y", 0.0040, effective=VALUE_DT.add_months(9)),
            _swap("6Y", 0.0035, effective=VALUE_DT.add_months(-8))]


def _oracle(m, swap):
    gbp, usd, x = m.curves.GBP_OIS_SONIA, m.curves.USD_OIS_SOFR, m.curves.USD_GBP_BASIS
    return XO.xccy_analytics(swap, VALUE_DT, _cache(gbp), gbp._interp_type.value, _cache(usd), usd._interp_type.value,
                             x, times_from_dates)


def test_single_swap_results_object(host_engine):
    m = host_engine
    swap = _book()[1]
    res = swap.position(m).compute([RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA])
    want = _oracle(m, swap)
    scale = abs(swap._domestic_leg._notional)
    assert abs(res.value.amount - want["value"]) <= 1e-10 * scale
    for c, d, g in zip(CURVES, ("delta_dom", "delta_for", "delta_basis"), ("gamma_dom", "gamma_for", "gamma_basis")):
        assert np.max(np.abs(res.risk(c).risk_ladder - want[d])) <= 1e-10 * scale * 1e-4
        assert np.max(np.abs(res.gamma(c).risk_ladder - want[g])) <= 1e-10 * scale * 1e-6
        assert len(res.risk(c).tenors) == len(want[d])
    assert res.risk(CurveTypes.GBP_OIS_SONIA).currency.name == "GBP"


def test_book_per_trade_and_aggregate(host_engine):
    m = host_engine
    book = _book()
    from adrates_amd.market.position.engine import Engine
    out = xccy_engine.price_xccy_batch(Engine(m), book, {RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA},
                                       per_trade=True, aggregate=True)
    wants = [_oracle(m, s) for s in book]
    for i, (s, w) in enumerate(zip(book, wants)):
        scale = abs(s._domestic_leg._notional)
        assert abs(out["pv"][i] - w["value"]) <= 1e-10 * scale
        for k in ("delta_dom", "delta_for", "delta_basis"):
            assert np.max(np.abs(out[k][i] - w[k])) <= 1e-10 * scale * 1e-4
        for k in ("gamma_dom", "gamma_for", "gamma_basis"):
            assert np.max(np.abs(out[k][i] - w[k])) <= 1e-10 * scale * 1e-6
    total = sum(abs(s._domestic_leg._notional) for s in book)
    assert abs(out["agg_pv"] - sum(w["value"] for w in wants)) <= 1e-10 * total
    for k in ("delta_dom", "delta_for", "delta_basis", "gamma_dom", "gamma_for", "gamma_basis"):
        assert np.max(np.abs(out["agg_" + k] - sum(w[k] for w in wants))) <= 1e-10 * total * 1e-4


def test_mixed_currency_pairs_are_refused(host_engine):
    from adrates_amd.market.position.engine import Engine
    from adrates_amd.utils import CurrencyTypes
    from adrates_amd.utils.error import LibError
    a, b = _book()[:2]
    b._foreign_currency = CurrencyTypes.EUR
    with pytest.raises(LibError, match="share"):
        xccy_engine.price_xccy_batch(Engine(host_engine), [a, b], {RequestTypes.VALUE})


def test_weighted_coupons_are_linear_in_the_weights():
    """oracle/port.c's per-coupon weights against its own unweighted path: a weighted trade equals the sum of
    its coupons priced one by one with notional N * w_j."""
    from adrates_amd.trades.compiler import TradeBatch
    m = _model()
    c = _cache(m.curves.USD_OIS_SOFR)
    rng = np.random.default_rng(3)
    k = 9
    ts = np.sort(rng.uniform(0.0, 12.0, k)); te = ts + rng.uniform(0.2, 1.1, k); tp = te + rng.choice([0.0, 0.01], k)
    al = te - ts; w = rng.uniform(0.3, 1.2, k)
    z = np.zeros(0)
    one = TradeBatch(np.array([0, 0]), np.array([0, k]), z, z, tp, ts, te, al, np.array([3e6]), np.array([0.002]),
                     np.array([1.0]), np.array([-1.0]), flt_weight=w)
    each = TradeBatch(np.zeros(k + 1, dtype=np.int64), np.arange(k + 1), z, z, tp, ts, te, al, 3e6 * w,
                      np.full(k, 0.002), np.ones(k), -np.ones(k))
    a = port.price(1, c["times"], c["dfs"], c["jac"], c["hess"], one)
    b = port.price(1, c["times"], c["dfs"], c["jac"], c["hess"], each)
    assert np.isclose(a["pv"][0], b["pv"].sum(), rtol=1e-13)
    assert np.allclose(a["delta"][0], b["delta"].sum(0), rtol=1e-12, atol=1e-9)
    assert np.allclose(a["gamma"][0], b["gamma"].sum(0), rtol=1e-12, atol=1e-9)


def _collateral_model():
    from adrates_amd.utils import DayCountTypes, InterpTypes
    from tests.test_gpu_xccy import BASIS, SPOT, TENORS
    m = _model()
    # GBP cash flows under USD collateral: GBP discount factors seen from USD
    m.build_xccy_curve(name="GBP_USD_XCCY", domestic_curve_name="USD_OIS_SOFR", foreign_curve_name="GBP_OIS_SONIA",
                       basis_spreads=[-b * 1e4 for b in BASIS], tenor_list=TENORS, spot_fx=1.0 / SPOT,
                       domestic_dc_type=DayCountTypes.ACT_360, foreign_dc_type=DayCountTypes.ACT_365F,
                       interp_type=InterpTypes.FLAT_FWD_RATES)
    return m


def _collateral_cases():
    from tests._fixtures import make_swap
    return [make_swap(VALUE_DT, "7Y", 0.047, notional=20e6, pay=True, spread=0.001),
            make_swap(VALUE_DT.add_months(-5), "4Y", 0.051, notional=5e6, pay=False, float_freq=FrequencyTypes.SEMI_ANNUAL),
            make_swap(VALUE_DT.add_months(6), "10Y", 0.046, notional=8e6, pay=True, payment_lag=2)]


def check_ois_collateral(m):
    from adrates_amd.utils import CollateralType
    gbp, x = m.curves.GBP_OIS_SONIA, m.curves.GBP_USD_XCCY
    for swap in _collateral_cases():
        res = swap.position(m).compute([RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.CASHFLOWS],
                                       collateral_type=CollateralType.USD)
        want = XO.ois_xccy_collateral_analytics(swap, VALUE_DT, _cache(gbp), gbp._interp_type.value, x, times_from_dates)
        scale = abs(swap._float_leg._notional)
        assert res.value.currency.name == "USD" and res.gamma is None and res.cashflows.cashflows == []
        assert abs(res.value.amount - want["value"]) <= 1e-10 * scale
        assert np.max(np.abs(res.risk(CurveTypes.GBP_OIS_SONIA).risk_ladder - want["delta_ois"])) <= 1e-10 * scale * 1e-4
        assert np.max(np.abs(res.risk(CurveTypes.USD_GBP_BASIS).risk_ladder - want["delta_basis"])) <= 1e-10 * scale * 1e-4
        assert np.any(want["delta_basis"] != 0.0) and abs(want["value"]) > 1.0
        # same collateral currency as the swap: the single-curve path (engine.py:126-151)
        natural = swap.position(m).compute([RequestTypes.VALUE], collateral_type=CollateralType.GBP)
        assert natural.value.currency.name == "GBP" and natural.value.amount != res.value.amount
        with pytest.raises(NotImplementedError, match="GAMMA"):
            swap.position(m).compute([RequestTypes.GAMMA], collateral_type=CollateralType.USD)
    from adrates_amd.utils.error import LibError
    with pytest.raises(LibError, match="GBP_EUR_XCCY"):
        _collateral_cases()[0].position(m).compute([RequestTypes.VALUE], collateral_type=CollateralType.EUR)


def test_ois_with_cross_currency_collateral(host_engine):
    check_ois_collateral(_collateral_model())


def test_portfolio_with_cross_currency_positions(host_engine):
    """Portfolio.compute (portfolio.py:39-66) adds results with `+`: values of XCCY positions add up, two `Risk`
    containers do not (no `__add__`, results.py:839-942) - same as the reference."""
    from adrates_amd.market.portfolio.portfolio import Portfolio
    m = host_engine
    book = _book()[:3]
    pf = Portfolio([s.position(m) for s in book])
    total = pf.compute([RequestTypes.VALUE]).value.amount
    assert abs(total - sum(_oracle(m, s)["value"] for s in book)) <= 1e-10 * sum(abs(s._domestic_leg._notional) for s in book)
    one = Portfolio([book[0].position(m)]).compute([RequestTypes.VALUE, RequestTypes.DELTA])
    assert one.risk(CurveTypes.USD_GBP_BASIS).risk_ladder.shape == (13,)
    with pytest.raises(TypeError):
        pf.compute([RequestTypes.DELTA])


def test_ladders_against_bump_and_reprice(host_engine):
    """The reference's own check of this path (tests/test_engine_basis_swap.py:152-332): each delta ladder against
    central differences of VALUE under a 1 bp move of one quote, the other curves held as the engine holds them
    (domestic / foreign: the XCCY curve object is reused; basis: the XCCY curve is re-bootstrapped)."""
    from adrates_amd.utils import BusDayAdjustTypes, DayCountTypes, InterpTypes, SwapTypes
    from tests.test_gpu_xccy import BASIS, GBP, SPOT, TENORS, USD

    def model(gbp=GBP, usd=USD, basis=BASIS, reuse=None):
        from adrates_amd.models.models import Model
        m = Model(VALUE_DT)
        for name, px, dc in (("GBP_OIS_SONIA", gbp, DayCountTypes.ACT_365F), ("USD_OIS_SOFR", usd, DayCountTypes.ACT_360)):
            m.build_curve(name=name, px_list=list(px), tenor_list=TENORS, spot_days=0, swap_type=SwapTypes.PAY,
                          fixed_dcc_type=dc, fixed_freq_type=FrequencyTypes.ANNUAL, float_freq_type=FrequencyTypes.ANNUAL,
                          float_dc_type=dc, bus_day_type=BusDayAdjustTypes.MODIFIED_FOLLOWING,
                          interp_type=InterpTypes.FLAT_FWD_RATES)
        if reuse is not None:
            m._curves_dict["USD_GBP_BASIS"] = reuse
            setattr(m.curves, "USD_GBP_BASIS", reuse)
        else:
            m.build_xccy_curve(name="USD_GBP_BASIS", domestic_curve_name="GBP_OIS_SONIA", foreign_curve_name="USD_OIS_SOFR",
                               basis_spreads=[b * 1e4 for b in basis], tenor_list=TENORS, spot_fx=SPOT,
                               domestic_dc_type=DayCountTypes.ACT_365F, foreign_dc_type=DayCountTypes.ACT_360,
                               interp_type=InterpTypes.FLAT_FWD_RATES)
        return m

    swap = _swap("7Y", 0.0060, notional=25_000_000, freq=FrequencyTypes.SEMI_ANNUAL)
    value = lambda m: swap.position(m).compute([RequestTypes.VALUE]).value.amount
    base = model()
    x = base.curves.USD_GBP_BASIS
    res = swap.position(base).compute([RequestTypes.VALUE, RequestTypes.DELTA])
    h = 0.01                                             # quotes in percent: 1 bp
    for curve, quotes, key in ((CurveTypes.GBP_OIS_SONIA, GBP, "gbp"), (CurveTypes.USD_OIS_SOFR, USD, "usd")):
        ladder = res.risk(curve).risk_ladder
        for i in (0, 3, 6, 7):
            up, dn = list(quotes), list(quotes)
            up[i] += h; dn[i] -= h
            fd = (value(model(**{key: up}, reuse=x)) - value(model(**{key: dn}, reuse=x))) / 2.0
            assert abs(fd - ladder[i]) <= 2e-6 * max(1.0, abs(ladder).max()), (curve, i, fd, ladder[i])
    ladder = res.risk(CurveTypes.USD_GBP_BASIS).risk_ladder
    for i in (2, 6, 7):
        up, dn = list(BASIS), list(BASIS)
        up[i] += 1e-4; dn[i] -= 1e-4                     # basis spreads are decimals: 1 bp = 1e-4
        fd = (value(model(basis=up)) - value(model(basis=dn))) / 2.0
        assert abs(fd - ladder[i]) <= 2e-6 * max(1.0, abs(ladder).max()), (i, fd, ladder[i])


def test_matured_and_maturing_swaps(host_engine):
    check_matured_and_maturing(host_engine)


def check_matured_and_maturing(m):
    """Edge cases of the masks (engine.py:2694 `>=`, :2706-2722): a swap that matured before the value date is worth
    nothing and has empty ladders; one maturing today keeps its last coupon and final exchange (discount factor 1,
    no sensitivity); an empty book is refused."""
    from adrates_amd.market.position.engine import Engine
    from adrates_amd.utils.error import LibError
    gone = _swap(VALUE_DT.add_months(-1), 0.003, effective=VALUE_DT.add_months(-25))
    today = _swap(VALUE_DT, 0.003, effective=VALUE_DT.add_months(-24))
    live = _book()[0]
    out = xccy_engine.price_xccy_batch(Engine(m), [gone, today, live], {RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA})
    assert out["pv"][0] == 0.0
    for k in ("delta_dom", "delta_for", "delta_basis", "gamma_dom", "gamma_for", "gamma_basis"):
        assert not np.any(out[k][0]) and not np.any(out[k][1])
    want = _oracle(m, today)
    assert abs(want["value"]) > 100.0 and abs(out["pv"][1] - want["value"]) <= 1e-10 * abs(today._domestic_leg._notional)
    assert abs(out["pv"][2] - _oracle(m, live)["value"]) <= 1e-10 * abs(live._domestic_leg._notional)
    with pytest.raises(LibError, match="at least one"):
        xccy_engine.price_xccy_batch(Engine(m), [], {RequestTypes.VALUE})


def test_terms_compiler_equals_the_object_path_and_the_book_is_distinct(host_engine):
    """`xccy_engine.raw_from_terms` (schedules on arrays, `utils.schedule_np`) gives, bit for bit, the arrays
    `raw_from_swaps` builds from the corresponding `XccyBasisSwap` objects; a synthetic book (trades/synthetic_xccy.py)
    is reproducible, every swap in it is its own, and rank shares of one book tile it."""
    import dataclasses
    from adrates_amd.market.position.engine import Engine
    from adrates_amd.trades import synthetic_xccy as SX
    from adrates_amd.trades.rates.xccy_basis_swap import XccyBasisSwap
    from adrates_amd.utils import Date
    from tests._fixtures import GBP_PX, README_VALUE_DT, TENORS, USD_PX
    m = SX.build_market(README_VALUE_DT, GBP_PX, USD_PX, TENORS)
    xdc = m.curves.USD_GBP_BASIS._dc_type
    terms, work = SX.draw_terms(README_VALUE_DT, 300, seed=3)
    fast = xccy_engine.raw_from_terms(terms, README_VALUE_DT, xdc)
    at = lambda col, i: col[1][int(col[0][i])]                       # (codes, table) columns
    swaps = [XccyBasisSwap(effective_dt=Date._from_serial(int(terms.effective_dt[i])), term_dt_or_tenor=at(terms.tenor, i),
                           domestic_notional=float(terms.domestic_notional[i]), foreign_notional=float(terms.foreign_notional[i]),
                           domestic_spread=float(terms.domestic_spread[i]), foreign_spread=float(terms.foreign_spread[i]),
                           domestic_freq_type=terms.domestic_freq_type, foreign_freq_type=at(terms.foreign_freq_type, i),
                           domestic_dc_type=terms.domestic_dc_type, foreign_dc_type=terms.foreign_dc_type,
                           domestic_floating_index=terms.domestic_floating_index,
                           foreign_floating_index=terms.foreign_floating_index, domestic_currency=terms.domestic_currency,
                           foreign_currency=terms.foreign_currency) for i in range(300)]
    slow = xccy_engine.raw_from_swaps(swaps, README_VALUE_DT, xdc)
    for f in dataclasses.fields(fast):
        assert np.array_equal(getattr(fast, f.name), getattr(slow, f.name)), f.name
    assert np.all(np.abs(np.diff(fast.dom_off) + np.diff(fast.for_off) - work) <= 3) and np.all(work >= 2)
    # the three batches from terms and from objects are the same batches
    a = xccy_engine.book_batches(Engine(m), terms)[6]
    b = xccy_engine.book_batches(Engine(m), swaps)[6]
    for x, y in zip(a, b):
        for f in ("fix_off", "flt_off", "fix_tp", "fix_pay", "flt_tp", "flt_ts", "flt_te", "flt_alpha", "notional", "spread"):
            assert np.array_equal(getattr(x, f), getattr(y, f)), f
    assert np.array_equal(a[1].flt_weight, b[1].flt_weight)

    parts, spot = SX.synthesize_book(Engine(m), README_VALUE_DT, 500, seed=11)
    again, _ = SX.synthesize_book(Engine(m), README_VALUE_DT, 500, seed=11)
    for (x, _), (y, _) in zip(parts, again):
        assert np.array_equal(x.fix_pay, y.fix_pay) and np.array_equal(x.notional, y.notional)
    rates = parts[1][0]
    assert rates.flt_weight is not None and rates.flt_weight.shape == rates.flt_tp.shape
    assert np.all(np.diff(rates.flt_off) >= 1) and rates.n_trades == 500
    t500, _ = SX.draw_terms(README_VALUE_DT, 500, seed=11)
    key = np.stack([t500.effective_dt, t500.foreign_notional, t500.foreign_spread, t500.domestic_spread], axis=1)
    assert len(np.unique(key, axis=0)) == 500                     # no two swaps alike
    shares = [SX.synthesize_book(Engine(m), README_VALUE_DT, 500, seed=11, rank=r, world_size=3)[0] for r in range(3)]
    for piece in range(3):
        whole = parts[piece][0]
        assert sum(s[piece][0].n_trades for s in shares) == 500
        assert np.array_equal(np.concatenate([s[piece][0].notional for s in shares]), whole.notional)
        assert np.array_equal(np.concatenate([s[piece][0].fix_pay for s in shares]), whole.fix_pay)
    templates = SX.template_swaps(README_VALUE_DT)[:6]
    dom, rts, flows = xccy_engine.book_batches(Engine(m), templates)[6]
    same = SX.take(flows, np.arange(6), np.ones(6))
    assert np.array_equal(same.fix_off, flows.fix_off) and np.array_equal(same.fix_pay, flows.fix_pay)
    w = SX.take(rts, [5, 0], [1.0, 1.0])
    assert np.array_equal(w.flt_weight[:w.flt_off[1]], rts.flt_weight[rts.flt_off[5]:rts.flt_off[6]])


def test_cashflows_request_is_refused_for_cross_currency_swaps(host_engine):
    with pytest.raises(NotImplementedError, match="CASHFLOWS"):
        _book()[0].position(host_engine).compute([RequestTypes.VALUE, RequestTypes.CASHFLOWS])


def test_linear_fwd_rates_on_all_three_curves(host_engine):
    """The assembly under LINEAR_FWD_RATES (linear in the knot DFs): D_x and the forwards come from the curve
    lookups' stand-in, the kernels' stand-in (oracle/port.c) differentiates the linear scheme."""
    import tests.test_gpu_xccy as G
    from adrates_amd.utils import InterpTypes
    G.test_other_interpolation_schemes_on_all_three_curves(InterpTypes.LINEAR_FWD_RATES)


def test_two_curves_with_identical_pillar_times_do_not_collide(host_engine):
    """The reference's per-Engine curve cache is keyed by tuple(swap_times) only (engine.py:2362-2412): a domestic
    and a foreign curve built on the same day count and tenors would share one entry.  Here they do not."""
    from adrates_amd.market.position.engine import Engine
    from adrates_amd.models.models import Model
    from adrates_amd.utils import BusDayAdjustTypes, DayCountTypes, InterpTypes, SwapTypes
    from tests.test_gpu_xccy import GBP, TENORS, USD
    m = Model(VALUE_DT)
    for name, px in (("GBP_OIS_SONIA", GBP), ("USD_OIS_SOFR", USD)):
        m.build_curve(name=name, px_list=list(px), tenor_list=TENORS, spot_days=0, swap_type=SwapTypes.PAY,
                      fixed_dcc_type=DayCountTypes.ACT_365F, fixed_freq_type=FrequencyTypes.ANNUAL,
                      float_freq_type=FrequencyTypes.ANNUAL, float_dc_type=DayCountTypes.ACT_365F,
                      bus_day_type=BusDayAdjustTypes.MODIFIED_FOLLOWING, interp_type=InterpTypes.FLAT_FWD_RATES)
    gbp, usd = m.curves.GBP_OIS_SONIA, m.curves.USD_OIS_SOFR
    assert tuple(gbp.swap_times) == tuple(usd.swap_times)
    engine = Engine(m)
    a, b = engine._device_curve(gbp), engine._device_curve(usd)
    assert a is not b and not np.array_equal(a["host"].dfs, b["host"].dfs)


def test_cross_gamma_is_the_bump_of_the_basis_delta(host_engine):
    """`Risk.cross_gamma(foreign OIS, basis)`: d2 PV / d r_for d s_basis with the XCCY curve's knot DFs and basis
    Jacobian held as they are - so it must equal the central difference of the basis delta ladder when one foreign
    quote is bumped and the SAME XCCY curve object is kept (the curve is not re-bootstrapped: that coupling is the
    other, reference-only, term - DESIGN.md section 9).  Also against the autodiff restatement, and the book form."""
    from adrates_amd.market.position.engine import Engine
    from adrates_amd.models.models import Model
    from adrates_amd.utils import BusDayAdjustTypes, DayCountTypes, InterpTypes, SwapTypes
    from tests.test_gpu_xccy import TENORS, USD
    m = host_engine
    swap = _swap("7Y", 0.0045, lag=2, notional=25_000_000, freq=FrequencyTypes.SEMI_ANNUAL)
    reqs = {RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA}
    res = swap.position(m).compute(list(reqs))
    cross = res.gamma.cross_gamma(CurveTypes.USD_OIS_SOFR, CurveTypes.USD_GBP_BASIS)
    want = _oracle(m, swap)["cross_for_basis"]
    scale = max(np.max(np.abs(want)), 1e-12 * 25e6)
    assert np.max(np.abs(cross.risk_matrix - want)) <= 1e-10 * scale
    assert abs(cross.value.amount - want.sum()) <= 1e-9 * np.abs(want).sum()
    assert res.gamma.has_cross_gamma(CurveTypes.USD_OIS_SOFR, CurveTypes.USD_GBP_BASIS)
    assert res.gamma.cross_gamma(CurveTypes.GBP_OIS_SONIA, CurveTypes.USD_GBP_BASIS) is None

    def basis_delta(shift_bp, tenor):
        usd = m.scenario("USD_OIS_SOFR", {tenor: shift_bp * 0.01}).curves.USD_OIS_SOFR
        bumped = Model(VALUE_DT)
        bumped._curves_dict.update(GBP_OIS_SONIA=m.curves.GBP_OIS_SONIA, USD_OIS_SOFR=usd, USD_GBP_BASIS=m.curves.USD_GBP_BASIS)
        return swap.position(bumped).compute([RequestTypes.DELTA]).risk(CurveTypes.USD_GBP_BASIS).risk_ladder
    for tenor in ("2Y", "5Y", "7Y"):
        l = TENORS.index(tenor)
        fd = (basis_delta(+1.0, tenor) - basis_delta(-1.0, tenor)) / 2.0           # per bp of the foreign quote
        assert np.max(np.abs(cross.risk_matrix[l] - fd)) <= 1e-6 * np.max(np.abs(cross.risk_matrix)), tenor
    # a book: per swap and aggregated
    book = _book()
    out = xccy_engine.price_xccy_batch(Engine(m), book, reqs, per_trade=True, aggregate=True, cross_gamma=True)
    wants = [_oracle(m, s)["cross_for_basis"] for s in book]
    for i, w in enumerate(wants):
        assert np.max(np.abs(out["cross_for_basis"][i] - w)) <= 1e-10 * max(np.max(np.abs(w)), 1e-12 * abs(book[i]._domestic_leg._notional))
    tot = sum(wants)
    assert np.max(np.abs(out["agg_cross_for_basis"] - tot)) <= 1e-10 * np.max(np.abs(tot))


def test_cross_gamma_is_labelled_and_can_be_switched_off(host_engine, monkeypatch):
    """The foreign OIS x basis matrix is not the reference's block (engine.py:1892-1958): it says so
    (`definition == "direct"`), is attached under the reference's condition (the XCCY curve carries
    `_mixed_hess_foreign_basis`, :1894) and `xccy_engine.CROSS_GAMMA_MODE = "off"` leaves the slot empty."""
    m = host_engine
    swap = _swap("5Y", 0.003, notional=10_000_000)
    reqs = [RequestTypes.VALUE, RequestTypes.GAMMA]
    res = swap.position(m).compute(reqs)
    cross = res.gamma.cross_gamma(CurveTypes.USD_OIS_SOFR, CurveTypes.USD_GBP_BASIS)
    assert cross is not None and cross.definition == "direct"
    monkeypatch.setattr(xccy_engine, "CROSS_GAMMA_MODE", "off")
    off = swap.position(m).compute(reqs)
    assert not off.gamma.has_cross_gamma(CurveTypes.USD_OIS_SOFR, CurveTypes.USD_GBP_BASIS)
    assert np.array_equal(off.gamma(CurveTypes.USD_GBP_BASIS).risk_ladder, res.gamma(CurveTypes.USD_GBP_BASIS).risk_ladder)
    monkeypatch.setattr(xccy_engine, "CROSS_GAMMA_MODE", "direct")
    x = m.curves.USD_GBP_BASIS
    monkeypatch.setattr(x, "_mixed_hess_foreign_basis", None)
    assert not swap.position(m).compute(reqs).gamma.has_cross_gamma(CurveTypes.USD_OIS_SOFR, CurveTypes.USD_GBP_BASIS)


def test_value_and_delta_requests_through_the_two_curve_launch(host_engine, monkeypatch):
    """VALUE / DELTA requests: `compile_xccy_legs` + ONE foreign launch (`_price_fused`) - here with the kernel's formula restated
    on torch tensors standing in for adr_price_xccy_foreign - against the autodiff restatement of Engine._compute_xccy per swap,
    and against the three-batch assembly as book sums."""
    from adrates_amd.market.position.engine import Engine
    m = host_engine
    book = _book()
    calls = []
    monkeypatch.setattr(_native, "price_xccy_foreign", lambda *a, **k: (calls.append(1), _host_two_curve_launch(*a, **k))[1])
    reqs = {RequestTypes.VALUE, RequestTypes.DELTA}
    out = xccy_engine.price_xccy_batch(Engine(m), book, reqs, per_trade=True, aggregate=True)
    assert calls and "gamma_dom" not in out
    for i, s in enumerate(book):
        w = _oracle(m, s)
        scale = abs(s._domestic_leg._notional)
        assert abs(out["pv"][i] - w["value"]) <= 1e-10 * scale
        for k in ("delta_dom", "delta_for", "delta_basis"):
            assert np.max(np.abs(out[k][i] - w[k])) <= 1e-10 * scale * 1e-4, (i, k)
    monkeypatch.setattr(xccy_engine, "FUSED_FOREIGN_LEG", False)
    three = xccy_engine.price_xccy_batch(Engine(m), book, reqs, per_trade=True, aggregate=True)
    assert len(calls) == 1
    total = sum(abs(s._domestic_leg._notional) for s in book)
    assert abs(out["agg_pv"] - three["agg_pv"]) <= 1e-10 * total
    for k in ("delta_dom", "delta_for", "delta_basis"):
        assert np.max(np.abs(out["agg_" + k] - three["agg_" + k])) <= 1e-10 * total * 1e-4
    res = book[2].position(m).compute([RequestTypes.VALUE, RequestTypes.DELTA])          # the public API takes the same path
    assert abs(res.value.amount - _oracle(m, book[2])["value"]) <= 1e-10 * abs(book[2]._domestic_leg._notional)
