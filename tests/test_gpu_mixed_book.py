"""BASELINE.json configs[3] and configs[4] at their stated sizes on one GPU.

configs[3]: 100 000 GBP/USD cross-currency swaps - every swap distinct (adrates_amd/trades/synthetic_xccy.py), compiled
from terms by the vectorised compiler, per-coupon discount factors from the device lookups - ladders to SONIA, SOFR and
basis pillars, checked per trade against the C oracle and, for a sample, against the torch-autodiff restatement of
Engine._compute_xccy.
configs[4], one rank's slice: 1 000 000 OIS + 100 000 cross-currency swaps priced in one step into ONE aggregate
buffer (what the ranks all-reduce), checked against the oracle's sums; the same book cut in two shards gives the same
aggregate, which is all the multi-GPU path adds."""
import numpy as np
import pytest

from adrates_amd import _native
from adrates_amd.distributed import shard_batch
from adrates_amd.market.position import xccy_engine as XE
from adrates_amd.market.position.engine import Engine
from adrates_amd.trades import synthetic, synthetic_xccy as SX
from adrates_amd.utils import InterpTypes, RequestTypes
from adrates_amd.utils.helpers import times_from_dates
from oracle import cavour_oracle as O
from oracle import port
from oracle import xccy_oracle as XO

from . import _fixtures as F
from ._sweeps import xccy_book_case
from .test_gpu_parity_batch import _device_curve

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.mark.parametrize("method", [InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_ZERO_RATES, InterpTypes.LINEAR_FWD_RATES])
def test_device_curve_lookups_match_the_oracle_interpolation(gpu_ctx, method):
    """adr_curve_df (the batched InterpolatorAd.simple_interpolate on the GPU) against the oracle's restatement:
    knots, near-knots inside and outside the 1e-10 snap distance, interior points, negative times, both ends."""
    curve = F.gbp_model(F.README_VALUE_DT, method).curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    rng = np.random.default_rng(5)
    x = host.times
    t = np.concatenate([x, x[1:] + 5e-11, x[1:] - 3e-9, x[1:] + 2e-9, rng.uniform(0.0, x[-1] + 3.0, 2000), [0.0, -0.3, 80.0]])
    want = O.simple_interpolate(t, x, host.dfs, method.value).numpy()
    got = _native.curve_df(gpu_ctx, dc, t)
    np.testing.assert_allclose(got, want, rtol=2e-15, atol=0)
    assert _native.curve_df(gpu_ctx, dc, float(x[40])) == host.dfs[np.flatnonzero(x == x[40])[0]]     # snaps to the FIRST duplicate
    assert _native.curve_df(gpu_ctx, dc, np.zeros(0)).shape == (0,)


@pytest.mark.slow
def test_cross_currency_book_100k_distinct_swaps(gpu_ctx):
    rows = xccy_book_case(gpu_ctx, 100_000)
    for r in rows:
        assert r["judged"] <= TOL and r["agg_gamma_rel"] <= TOL, r
    assert rows[1]["cash_flows"] > 2_000_000 and {r["pillars"] for r in rows} == {32, 18}


def test_terms_book_against_the_autodiff_restatement(gpu_ctx):
    """A book given by terms (no per-swap objects) through `price_xccy_batch`: the assembled PV and the three ladders
    of a sample of its swaps against oracle/xccy_oracle.py on the corresponding objects, with the parity metric of
    tests/_parity.py (relative to the ladder's own scale, floors per unit notional)."""
    from adrates_amd.trades.rates.xccy_basis_swap import XccyBasisSwap
    from adrates_amd.utils import Date
    from tests.test_gpu_xccy import _cache
    vd = F.README_VALUE_DT
    m = SX.build_market(vd, F.GBP_PX, F.USD_PX, F.TENORS)
    _native.set_default_context(gpu_ctx)
    terms, _ = SX.draw_terms(vd, 400, seed=8)
    out = XE.price_xccy_batch(Engine(m), terms, {RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA})
    gbp, usd, x = m.curves.GBP_OIS_SONIA, m.curves.USD_OIS_SOFR, m.curves.USD_GBP_BASIS
    for i in (0, 57, 133, 262, 399):
        swap = XccyBasisSwap(effective_dt=Date._from_serial(int(terms.effective_dt[i])), term_dt_or_tenor=terms.tenor[1][int(terms.tenor[0][i])],
                             domestic_notional=float(terms.domestic_notional[i]), foreign_notional=float(terms.foreign_notional[i]),
                             domestic_spread=float(terms.domestic_spread[i]), foreign_spread=float(terms.foreign_spread[i]),
                             domestic_freq_type=terms.domestic_freq_type, foreign_freq_type=terms.foreign_freq_type[1][int(terms.foreign_freq_type[0][i])],
                             domestic_dc_type=terms.domestic_dc_type, foreign_dc_type=terms.foreign_dc_type,
                             domestic_floating_index=terms.domestic_floating_index,
                             foreign_floating_index=terms.foreign_floating_index, domestic_currency=terms.domestic_currency,
                             foreign_currency=terms.foreign_currency)
        want = XO.xccy_analytics(swap, vd, _cache(gbp), gbp._interp_type.value, _cache(usd), usd._interp_type.value, x,
                                 times_from_dates)
        n = abs(swap._domestic_leg._notional)
        assert abs(out["pv"][i] - want["value"]) <= TOL * max(abs(want["value"]), 1e-4 * n)
        for key, floor in (("delta_dom", 1e-8), ("delta_for", 1e-8), ("delta_basis", 1e-8),
                           ("gamma_dom", 1e-12), ("gamma_for", 1e-12), ("gamma_basis", 1e-12)):
            a, b = np.asarray(out[key][i]), np.asarray(want[key])
            # the domestic ladders of these near-par legs are rounding noise: judged per unit notional (floor)
            scale = max(np.max(np.abs(b)), (1e-6 if key.endswith("dom") else floor) * n)
            assert np.max(np.abs(a - b)) <= TOL * scale, (i, key, np.max(np.abs(a - b)) / scale)


@pytest.mark.slow
def test_cross_currency_book_100k_swaps_two_launches_vs_three(gpu_ctx, monkeypatch):
    """BASELINE configs[3] at its stated size through `price_xccy_batch`: VALUE + DELTA requests take two launches (the foreign
    leg on two curves, adr_price_xccy_foreign); the three-batch assembly - swept against the C port above - gives the same
    ladders per swap and as book sums, and the aggregate-only request the same book sums again."""
    vd = F.README_VALUE_DT
    m = SX.build_market(vd, F.GBP_PX, F.USD_PX, F.TENORS)
    _native.set_default_context(gpu_ctx)
    terms, _ = SX.draw_terms(vd, 100_000)
    reqs = {RequestTypes.VALUE, RequestTypes.DELTA}
    calls = []
    real = _native.price_xccy_foreign
    monkeypatch.setattr(XE._native, "price_xccy_foreign", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    two = XE.price_xccy_batch(Engine(m), terms, reqs, per_trade=True, aggregate=True)
    book = XE.price_xccy_batch(Engine(m), terms, reqs, per_trade=False, aggregate=True)
    assert len(calls) == 2
    monkeypatch.setattr(XE, "FUSED_FOREIGN_LEG", False)
    three = XE.price_xccy_batch(Engine(m), terms, reqs, per_trade=True, aggregate=True)
    assert len(calls) == 2
    assert np.max(np.abs(two["pv"] - three["pv"])) <= TOL * np.max(np.abs(three["pv"]))
    for key in ("delta_dom", "delta_for", "delta_basis"):
        a, b = np.asarray(two[key]), np.asarray(three[key])
        scale = np.max(np.abs(b), axis=1) + 1e-8 * np.abs(np.asarray(terms.domestic_notional))
        assert np.max(np.max(np.abs(a - b), axis=1) / scale) <= TOL, key
        tot = np.abs(b).sum(0).max()
        assert np.max(np.abs(two["agg_" + key] - three["agg_" + key])) <= TOL * tot, key
        assert np.max(np.abs(book["agg_" + key] - three["agg_" + key])) <= TOL * tot, key
    assert abs(book["agg_pv"] - three["agg_pv"]) <= TOL * np.abs(three["pv"]).sum()


@pytest.mark.slow
def test_mixed_book_one_step_one_aggregate_buffer(gpu_ctx):
    """configs[4], the slice of one rank: four launches (OIS; domestic, foreign-rates and foreign-flows pieces of the
    cross-currency book) write their aggregate ladders into ONE device buffer - the buffer the ranks all-reduce."""
    import torch
    vd = F.README_VALUE_DT
    n_ois, n_x = 1_000_000, 100_000
    market = SX.build_market(vd, F.GBP_PX, F.USD_PX, F.TENORS)
    _native.set_default_context(gpu_ctx)
    engine = Engine(market)
    ois_curve = F.readme_model().curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, ois_curve)
    ois = synthetic.synthesize(vd, n_ois)
    parts, spot = SX.synthesize_book(engine, vd, n_x)
    dev = torch.device("cuda", 0)
    P = 32
    offsets, total = [0], 1 + P + P * P
    for b, cur in parts:
        offsets.append(total)
        total += 1 + cur.n_pillars + cur.n_pillars ** 2

    def one_step(ois_batch, x_parts):
        agg = torch.zeros(total, dtype=torch.float64, device=dev)
        stream = torch.cuda.Stream(dev)
        keep = []
        with torch.cuda.stream(stream):
            for (batch, cur), off in zip([(ois_batch, dc)] + list(x_parts), offsets):
                t = _native.DeviceTrades(gpu_ctx, batch)
                keep.append(t)
                _native.price_dev(gpu_ctx, cur, t, 7, 0, 0, 0, agg.data_ptr() + 8 * off, stream.cuda_stream)
        stream.synchronize()
        for t in keep:
            t.close()
        return agg.cpu().numpy()

    got = one_step(ois, parts)

    # oracle sums, chunked so that the per-trade gamma of a million trades never sits in memory at once
    def oracle_sum(tab, batch, chunk=100_000):
        Pn = tab[3].shape[1]
        acc = np.zeros(1 + Pn + Pn * Pn)
        for lo in range(0, batch.n_trades, chunk):
            r = port.price(*tab, batch.slice(lo, min(lo + chunk, batch.n_trades)))
            acc[0] += r["pv"].sum(); acc[1:1 + Pn] += r["delta"].sum(0); acc[1 + Pn:] += r["gamma"].sum(0).ravel()
        return acc

    dom_model, for_model, xccy, dom_cur, for_cur, x_dev = XE._curves(engine, SX.template_swaps(vd)[:1])
    jac, hess = np.asarray(xccy._jac_basis), np.asarray(xccy._hess_basis)
    if jac.shape[1] % 2:
        jac, hess = np.pad(jac, ((0, 0), (0, 1))), np.pad(hess, ((0, 0), (0, 1), (0, 1)))
    tabs = [(4, host.times, host.dfs, host.jac, host.hess),
            (dom_model._interp_type.value, dom_cur["host"].times, dom_cur["host"].dfs, dom_cur["host"].jac, dom_cur["host"].hess),
            (for_model._interp_type.value, for_cur["host"].times, for_cur["host"].dfs, for_cur["host"].jac, for_cur["host"].hess),
            (xccy._interp_type.value, np.asarray(xccy._times), np.asarray(xccy._dfs), jac, hess)]
    batches = [ois] + [b for b, _ in parts]
    for name, tab, batch, off in zip(("ois", "domestic", "foreign_rates", "foreign_flows"), tabs, batches, offsets):
        want = oracle_sum(tab, batch)
        Pn = tab[3].shape[1]
        have = got[off:off + want.size]
        gross = np.abs(batch.notional).sum()
        # a sum of a million signed terms: relative to the ladder's own size, with a floor per unit of gross notional
        for lo, hi, floor in ((0, 1, 1e-6), (1, 1 + Pn, 1e-10), (1 + Pn, want.size, 1e-14)):
            scale = max(np.max(np.abs(want[lo:hi])), floor * gross)
            assert np.max(np.abs(have[lo:hi] - want[lo:hi])) <= TOL * scale, (name, lo)
    # PV of the cross-currency book in domestic currency: dom + flows / spot (the rates piece carries no PV)
    assert np.isfinite(got[offsets[1]] + got[offsets[3]] / spot)

    # two shards of the same book, priced one after the other, add up to the same buffer
    halves = np.zeros(total)
    for rank in range(2):
        mine, _ = shard_batch(ois, rank, 2)
        x_share, _ = SX.synthesize_book(engine, vd, n_x, rank=rank, world_size=2)
        halves += one_step(mine, x_share)
    scale = np.maximum(np.abs(got), 1e-12 * np.abs(ois.notional).sum())
    assert np.max(np.abs(halves - got) / scale) <= 1e-10
