"""Aggregate-only requests (agg and no per-trade output) - Portfolio.compute's single ladder
(cavour/market/portfolio/portfolio.py:39-66; BASELINE configs[4]'s aggregated risk ladder): the trades of the lite table are
summed in knot space and projected once per launch (kernels_lite.hip KNOT instantiations, kernels_knot.hip), every other trade
takes its usual kernel with the stores off.  The reference's chain rule (engine.py:2551-2567) is linear in a trade's
knot-space gradient and Hessian, so the book's ladder must equal the sum of the per-trade ladders: checked against
oracle/port.c's sums, against the per-trade kernels' own sums, for every scheme, pillar count and trade class, and bit for bit
between repeated launches."""
import numpy as np
import pytest

from adrates_amd import _native
from adrates_amd.trades import synthetic
from adrates_amd.utils import InterpTypes
from oracle import port

from . import _fixtures as F
from .test_gpu_many_pillars import _mixed_batch, forty_pillar_quotes
from .test_gpu_parity_batch import _device_curve

pytestmark = pytest.mark.gpu

SCHEMES = [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_FWD_RATES]


def assert_book(got, ref, what=""):
    """|book ladder - sum of the oracle's per-trade ladders| <= 1e-10 of the largest sum of absolute per-trade entries (the
    scale tests/test_gpu_parity_batch.py::test_full_size_properties uses: pay / receive trades cancel in the sum itself)."""
    worst = 0.0
    for key, r in (("agg_pv", ref["pv"]), ("agg_delta", ref.get("delta")), ("agg_gamma", ref.get("gamma"))):
        if r is None or key not in got:
            continue
        scale = float(np.max(np.abs(r).sum(0))) if r.ndim > 1 else float(np.abs(r).sum())
        err = float(np.max(np.abs(np.asarray(got[key]) - r.sum(0)))) / max(scale, 1e-300)
        assert err <= 1e-10, f"{what} {key}: {err:.2e}"
        worst = max(worst, err)
    return worst


@pytest.mark.parametrize("interp", SCHEMES)
@pytest.mark.parametrize("kind", ["offgrid", "ongrid"])
def test_plain_book_in_knot_space_vs_c_oracle(gpu_ctx, interp, kind):
    vd = F.README_VALUE_DT
    curve = F.gbp_model(vd, interp).curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    batch = synthetic.synthesize(vd, 20000, kind=kind)
    dt = _native.DeviceTrades(gpu_ctx, batch)
    ref = port.price(interp.value, host.times, host.dfs, host.jac, host.hess, batch)
    only = _native.price(gpu_ctx, dc, dt, per_trade=False, aggregate=True)
    worst = assert_book(only, ref, f"{interp.name} {kind}")
    assert np.array_equal(only["agg_gamma"], only["agg_gamma"].T) or np.allclose(only["agg_gamma"], only["agg_gamma"].T, rtol=1e-13, atol=0)
    # the per-trade kernels' own aggregate: the same book ladder by the other route
    full = _native.price(gpu_ctx, dc, dt, aggregate=True)
    scale = np.abs(full["gamma"]).sum(0).max()
    assert np.max(np.abs(full["agg_gamma"] - only["agg_gamma"])) <= 1e-10 * scale
    assert np.max(np.abs(full["agg_delta"] - only["agg_delta"])) <= 1e-10 * np.abs(full["delta"]).sum(0).max()
    # PV + delta only, and bit-for-bit repeatability (per-wave tables, fixed-order reductions)
    d_only = _native.price(gpu_ctx, dc, dt, want_gamma=False, per_trade=False, aggregate=True)
    assert_book(d_only, {"pv": ref["pv"], "delta": ref["delta"]}, "delta only")
    assert not np.any(d_only["agg_gamma"])
    again = _native.price(gpu_ctx, dc, dt, per_trade=False, aggregate=True)
    for key in ("agg_pv", "agg_delta", "agg_gamma"):
        assert np.array_equal(np.asarray(only[key]), np.asarray(again[key])), key
    dt.close()
    print(f"knot-space book, {interp.name} {kind}: worst error {worst:.2e}")


@pytest.mark.parametrize("interp", SCHEMES)
def test_mixed_book_knot_pass_plus_other_kernels(gpu_ctx, interp):
    """Payment lags, semi-annual legs of up to 80 coupons, spreads: the lite table's trades in knot space, the others on the
    payment-lag variant / chained rows / general kernel with the stores off, one ladder."""
    vd = F.README_VALUE_DT
    curve = F.gbp_model(vd, interp).curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    batch = _mixed_batch(vd, 6001, seed=21)
    dt = _native.DeviceTrades(gpu_ctx, batch)
    ref = port.price(interp.value, host.times, host.dfs, host.jac, host.hess, batch)
    only = _native.price(gpu_ctx, dc, dt, per_trade=False, aggregate=True)
    worst = assert_book(only, ref, interp.name)
    d_only = _native.price(gpu_ctx, dc, dt, want_gamma=False, per_trade=False, aggregate=True)
    assert_book(d_only, {"pv": ref["pv"], "delta": ref["delta"]}, "delta only")
    dt.close()
    print(f"mixed book, {interp.name}: worst error {worst:.2e}")


@pytest.mark.parametrize("pillars", [17, 31, 40, 64])
def test_other_pillar_counts(gpu_ctx, pillars):
    """The knot-space sums do not know the pillar count; the projection reads the 32-wide tiles (17, 31 pillars; 40 on the
    tiled route) or the wide layout's packed triangle (40, 64)."""
    from adrates_amd.trades.market_data import GBP_PX, TENORS
    vd = F.README_VALUE_DT
    if pillars == 40:
        px, tenors = forty_pillar_quotes()
    elif pillars == 64:
        years = lambda s: float(s[:-1]) * {"D": 1 / 365, "W": 7 / 365, "M": 1 / 12, "Y": 1.0}[s[-1]]
        extra = [f"{y}Y" for y in range(1, 50) if f"{y}Y" not in TENORS]
        tenors = sorted(list(TENORS) + extra, key=years)[:64]
        base_t = [years(t) for t in TENORS]
        px = [float(np.interp(years(t), base_t, GBP_PX)) if t not in TENORS else GBP_PX[TENORS.index(t)] for t in tenors]
    elif pillars == 31:
        px, tenors = list(GBP_PX[:13]) + list(GBP_PX[14:]), list(TENORS[:13]) + list(TENORS[14:])
    else:
        px, tenors = list(GBP_PX[8:9] + GBP_PX[14:30]), list(TENORS[8:9] + TENORS[14:30])      # 6M, 1Y ... 30Y
    curve = F.gbp_model(vd, px=px, tenors=tenors).curves.GBP_OIS_SONIA
    batch = _mixed_batch(vd, 3001, seed=8)
    dt = _native.DeviceTrades(gpu_ctx, batch)
    flag_sets = [0] + ([_native.DeviceCurve.PILLAR_TILES] if pillars == 40 else [])
    for flags in flag_sets:
        host, dc = _device_curve(gpu_ctx, curve, flags=flags)
        assert dc.n_pillars == pillars
        ref = port.price(4, host.times, host.dfs, host.jac, host.hess, batch)
        only = _native.price(gpu_ctx, dc, dt, per_trade=False, aggregate=True)
        worst = assert_book(only, ref, f"{pillars} pillars")
        d_only = _native.price(gpu_ctx, dc, dt, want_gamma=False, per_trade=False, aggregate=True)
        assert_book(d_only, {"pv": ref["pv"], "delta": ref["delta"]}, "delta only")
        print(f"{pillars} pillars (flags {flags}): worst error {worst:.2e}")
    dt.close()


def test_bench_portfolio_at_full_size(gpu_ctx):
    """1 M benchmark trades: the aggregate-only ladder against the per-trade kernels' ladder of the same launch family and,
    for 50 000 of the trades priced as a book of their own, against oracle/port.c's sums."""
    import torch
    vd = F.README_VALUE_DT
    curve = F.readme_model().curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    n = 1_000_000
    batch = synthetic.synthesize(vd, n, seed=synthetic.DEFAULT_SEED)
    dt = _native.DeviceTrades(gpu_ctx, batch)
    P = dc.n_pillars
    dev = torch.device("cuda", 0)
    pv = torch.empty(n, dtype=torch.float64, device=dev)
    de = torch.empty((n, P), dtype=torch.float64, device=dev)
    ga = torch.empty((n, P, P), dtype=torch.float64, device=dev)
    ag = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
    ak = torch.empty(1 + P + P * P, dtype=torch.float64, device=dev)
    _native.price_dev(gpu_ctx, dc, dt, 7, pv.data_ptr(), de.data_ptr(), ga.data_ptr(), ag.data_ptr())
    _native.price_dev(gpu_ctx, dc, dt, 7, 0, 0, 0, ak.data_ptr())
    gpu_ctx.sync()
    scale_g, scale_d = float(ga.abs().sum(0).max()), float(de.abs().sum(0).max())
    assert float((ak[1 + P:] - ga.sum(0).reshape(-1)).abs().max()) <= 1e-10 * scale_g
    assert float((ak[1:1 + P] - de.sum(0)).abs().max()) <= 1e-10 * scale_d
    assert abs(float(ak[0] - pv.sum())) <= 1e-10 * float(pv.abs().sum())
    assert float((ak - ag).abs().max()) <= 1e-10 * max(scale_g, scale_d, float(pv.abs().sum()))
    del pv, de, ga
    dt.close()
    sub = batch.slice(300_000, 350_000)
    dts = _native.DeviceTrades(gpu_ctx, sub)
    ref = port.price(4, host.times, host.dfs, host.jac, host.hess, sub)
    only = _native.price(gpu_ctx, dc, dts, per_trade=False, aggregate=True)
    worst = assert_book(only, ref, "50 000 bench trades")
    dts.close()
    print(f"bench portfolio: worst error of the 50 000-trade book {worst:.2e}")


def test_xccy_book_ladders_aggregate_only(gpu_ctx):
    """A book of 3 000 distinct basis swaps: the aggregate-only call (domestic legs and foreign flows in knot space, the
    weighted foreign-rate coupons on their own kernels with the stores off) against the sums of the per-trade ladders, which
    tests/test_gpu_xccy.py / test_gpu_mixed_book.py hold to the oracles."""
    from adrates_amd.market.position import xccy_engine as XE
    from adrates_amd.market.position.engine import Engine
    from adrates_amd.trades import synthetic_xccy as SX
    from adrates_amd.utils import RequestTypes
    vd = F.README_VALUE_DT
    m = SX.build_market(vd, F.GBP_PX, F.USD_PX, F.TENORS)
    _native.set_default_context(gpu_ctx)
    terms, _ = SX.draw_terms(vd, 3000, seed=5)
    reqs = {RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA}
    full = XE.price_xccy_batch(Engine(m), terms, reqs, per_trade=True, aggregate=False)
    only = XE.price_xccy_batch(Engine(m), terms, reqs, per_trade=False, aggregate=True)
    assert abs(only["agg_pv"] - full["pv"].sum()) <= 1e-10 * np.abs(full["pv"]).sum()
    for key in ("delta_dom", "delta_for", "delta_basis", "gamma_dom", "gamma_for", "gamma_basis"):
        per = np.asarray(full[key])
        scale = float(np.max(np.abs(per).sum(0)))
        err = float(np.max(np.abs(np.asarray(only["agg_" + key]) - per.sum(0))))
        assert err <= 1e-10 * scale, (key, err / scale)


def test_ratio_nodes_far_apart_take_the_overflow_matrix(gpu_ctx):
    """Coupons that accrue from the short end over several years couple knots more than 16 apart: beyond the per-wave pair
    bands, into the launch's dense overflow matrix (global adds), which the projection scans only when it was used."""
    from adrates_amd.trades.compiler import TradeBatch
    vd = F.README_VALUE_DT
    rng = np.random.default_rng(4)
    n = 400
    ts = rng.uniform(0.01, 0.6, n)
    te = ts + rng.uniform(1.5, 7.0, n)
    tp = te + rng.choice([0.0, 0.008], size=n)           # half with payment lag, half ordinary coupons on long periods
    off = np.arange(n + 1, dtype=np.int64)
    batch = TradeBatch(np.zeros(n + 1, dtype=np.int64), off, np.zeros(0), np.zeros(0), tp, ts, te, te - ts,
                       np.round(rng.uniform(1e6, 5e7, n), -5), np.zeros(n), np.ones(n), np.where(rng.random(n) < 0.5, 1.0, -1.0))
    for interp in (InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES):
        curve = F.gbp_model(vd, interp).curves.GBP_OIS_SONIA
        host, dc = _device_curve(gpu_ctx, curve)
        dt = _native.DeviceTrades(gpu_ctx, batch)
        ref = port.price(interp.value, host.times, host.dfs, host.jac, host.hess, batch)
        only = _native.price(gpu_ctx, dc, dt, per_trade=False, aggregate=True)
        worst = assert_book(only, ref, f"far-apart ratio nodes, {interp.name}")
        full = _native.price(gpu_ctx, dc, dt, aggregate=True)
        assert_book(full, ref, "per-trade kernels")
        dt.close()
        print(f"far-apart ratio nodes, {interp.name}: worst error {worst:.2e}")
