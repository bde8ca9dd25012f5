"""Curves with more than 32 pillars, and with an odd pillar count (the reference has no pillar limit,
cavour/market/position/engine.py:2388-2389).  33-64 pillars: the wide variants of the general kernel (one wavefront = 64
pillars, the whole ladder in one launch; include/adrates.h, ADR_MAX_PILLARS), or - when the wide tables do not fit the
LDS, or with ADR_CURVE_PILLAR_TILES at upload - the general kernel once per pair of 32-pillar tiles; a 17-pillar curve takes
the general kernel for gamma (the fast kernel stores matrices as 16-byte pairs) and the lite kernel for delta."""
import numpy as np
import pytest

from adrates_amd import _native
from adrates_amd.trades import synthetic
from adrates_amd.trades.compiler import OISTerms, compile_ois_terms
from adrates_amd.utils import (BusDayAdjustTypes, CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes, InterpTypes,
                               RequestTypes)
from oracle import port

from . import _fixtures as F
from ._parity import assert_batch_parity
from .test_gpu_parity_batch import _device_curve

pytestmark = pytest.mark.gpu

EXTRA = ["11Y", "13Y", "14Y", "16Y", "17Y", "18Y", "19Y", "35Y"]


def _years(tenor):
    n, unit = int(tenor[:-1]), tenor[-1]
    return n / {"D": 365.0, "W": 52.0, "M": 12.0, "Y": 1.0}[unit]


def forty_pillar_quotes():
    """The 32 README quotes plus eight more annual pillars, quoted off the neighbours (linear in maturity)."""
    base_t = np.array([_years(t) for t in F.TENORS])
    tenors = sorted(list(F.TENORS) + EXTRA, key=_years)
    px = [float(np.interp(_years(t), base_t, F.GBP_PX)) if t in EXTRA else F.GBP_PX[F.TENORS.index(t)] for t in tenors]
    return px, tenors


def many_pillar_quotes(P):
    """The 32 README quotes plus annual pillars up to 49Y until there are P of them (quoted off the neighbours); beyond 62:
    monthly pillars between 1Y and 2Y, then half-year pillars (30M, 42M, ...) - up to 123."""
    base_t = np.array([_years(t) for t in F.TENORS])
    extra = [f"{y}Y" for y in range(1, 50) if f"{y}Y" not in F.TENORS]
    tenors = sorted(list(F.TENORS) + extra, key=_years)[:P]
    if P > len(tenors):
        more = [f"{m}M" for m in range(13, 24) if m != 18] + [f"{m}M" for m in range(30, 600, 12)]
        tenors = sorted(tenors + more[:P - len(tenors)], key=_years)
    assert len(tenors) == P
    px = [float(np.interp(_years(t), base_t, F.GBP_PX)) if t not in F.TENORS else F.GBP_PX[F.TENORS.index(t)] for t in tenors]
    return px, tenors


def _mixed_batch(vd, n, seed):
    rng = np.random.default_rng(seed)
    terms = OISTerms(effective_dt=vd, tenor=[f"{int(m)}M" for m in rng.integers(1, 481, n)],
                     coupon=rng.uniform(0.01, 0.07, n), notional=np.round(rng.uniform(1e6, 5e7, n), -5),
                     pay_fixed=rng.random(n) < 0.5, fixed_freq_type=FrequencyTypes.ANNUAL, fixed_dc_type=DayCountTypes.ACT_365F,
                     floating_index=CurveTypes.GBP_OIS_SONIA, currency=CurrencyTypes.GBP,
                     float_freq_type=[[FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL][i] for i in rng.integers(0, 2, n)],
                     float_dc_type=DayCountTypes.ACT_365F, float_spread=np.where(rng.random(n) < 0.3, 0.0015, 0.0),
                     payment_lag=rng.choice([0, 0, 2], size=n), bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
    return compile_ois_terms(terms, vd)


@pytest.mark.parametrize("interp", [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_FWD_RATES])
def test_forty_pillar_curve_vs_c_oracle(gpu_ctx, interp):
    vd = F.README_VALUE_DT
    px, tenors = forty_pillar_quotes()
    curve = F.gbp_model(vd, interp, px=px, tenors=tenors).curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    assert dc.n_pillars == 40
    batch = _mixed_batch(vd, 3001, seed=12)
    dt = _native.DeviceTrades(gpu_ctx, batch)
    ref = port.price(interp.value, host.times, host.dfs, host.jac, host.hess, batch)
    got = _native.price(gpu_ctx, dc, dt, aggregate=True)
    assert got["delta"].shape == (3001, 40) and got["gamma"].shape == (3001, 40, 40)
    worst = assert_batch_parity(got, ref, batch.notional)
    asym = np.max(np.abs(got["gamma"] - np.swapaxes(got["gamma"], 1, 2)), axis=(1, 2))
    assert np.all(asym <= 1e-12 * np.max(np.abs(got["gamma"]), axis=(1, 2)) + 1e-14)     # mirrored tiles / blocks, rounding inside diagonal blocks
    assert np.any(got["gamma"][:, :32, 32:] != 0.0)                       # the off-diagonal tile pair carries weight
    assert np.allclose(got["agg_pv"], ref["pv"].sum(), rtol=1e-10, atol=1e-3)
    assert np.allclose(got["agg_delta"], ref["delta"].sum(0), rtol=1e-10, atol=1e-6)
    assert np.allclose(got["agg_gamma"], ref["gamma"].sum(0), rtol=1e-10, atol=1e-9)
    only_d = _native.price(gpu_ctx, dc, dt, want_gamma=False, aggregate=True)
    assert_batch_parity(only_d, dict(pv=ref["pv"], delta=ref["delta"]), batch.notional)
    assert np.allclose(only_d["agg_delta"], ref["delta"].sum(0), rtol=1e-10, atol=1e-6) and np.all(only_d["agg_gamma"] == 0.0)
    only_v = _native.price(gpu_ctx, dc, dt, want_delta=False, want_gamma=False, aggregate=True)   # one launch: tile (0, 0)
    assert_batch_parity(only_v, dict(pv=ref["pv"]), batch.notional)
    assert np.allclose(only_v["agg_pv"], ref["pv"].sum(), rtol=1e-10, atol=1e-3)
    assert np.all(only_v["agg_delta"] == 0.0) and np.all(only_v["agg_gamma"] == 0.0)
    dt.close()
    print(f"40 pillars, {interp.name}: worst error {worst:.2e}")


@pytest.mark.parametrize("P,interp", [(47, InterpTypes.LINEAR_ZERO_RATES), (47, InterpTypes.LINEAR_FWD_RATES),
                                      (64, InterpTypes.FLAT_FWD_RATES), (64, InterpTypes.LINEAR_FWD_RATES),
                                      (33, InterpTypes.LINEAR_ZERO_RATES)])
def test_wide_kernel_block_counts_vs_c_oracle(gpu_ctx, P, interp):
    """One, two and three 4x4 blocks per lane (33 / 47 / 64 pillars; 47 and 33 are odd: scalar stores), payment lag and
    spreads in the batch, every request mask, the aggregate."""
    vd = F.README_VALUE_DT
    px, tenors = many_pillar_quotes(P)
    curve = F.gbp_model(vd, interp, px=px, tenors=tenors).curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    assert dc.n_pillars == P
    batch = _mixed_batch(vd, 2003, seed=P)
    dt = _native.DeviceTrades(gpu_ctx, batch)
    ref = port.price(interp.value, host.times, host.dfs, host.jac, host.hess, batch)
    got = _native.price(gpu_ctx, dc, dt, aggregate=True)
    assert got["delta"].shape == (2003, P) and got["gamma"].shape == (2003, P, P)
    worst = assert_batch_parity(got, ref, batch.notional)
    asym = np.max(np.abs(got["gamma"] - np.swapaxes(got["gamma"], 1, 2)), axis=(1, 2))
    assert np.all(asym <= 1e-12 * np.max(np.abs(got["gamma"]), axis=(1, 2)) + 1e-14)
    assert np.allclose(got["agg_pv"], ref["pv"].sum(), rtol=1e-10, atol=1e-3)
    assert np.allclose(got["agg_delta"], ref["delta"].sum(0), rtol=1e-10, atol=1e-6)
    assert np.allclose(got["agg_gamma"], ref["gamma"].sum(0), rtol=1e-10, atol=1e-9)
    only_d = _native.price(gpu_ctx, dc, dt, want_gamma=False, aggregate=True)
    assert_batch_parity(only_d, dict(pv=ref["pv"], delta=ref["delta"]), batch.notional)
    assert np.allclose(only_d["agg_delta"], ref["delta"].sum(0), rtol=1e-10, atol=1e-6) and np.all(only_d["agg_gamma"] == 0.0)
    only_v = _native.price(gpu_ctx, dc, dt, want_delta=False, want_gamma=False, aggregate=True)
    assert_batch_parity(only_v, dict(pv=ref["pv"]), batch.notional)
    assert np.all(only_v["agg_delta"] == 0.0) and np.all(only_v["agg_gamma"] == 0.0)
    dt.close()
    # a batch without payment lag or weighted coupons takes the kernel's instantiation without the ratio-node path
    plain = synthetic.synthesize(vd, 1501, seed=P)
    dtp = _native.DeviceTrades(gpu_ctx, plain)
    refp = port.price(interp.value, host.times, host.dfs, host.jac, host.hess, plain)
    gotp = _native.price(gpu_ctx, dc, dtp, aggregate=True)
    assert_batch_parity(gotp, refp, plain.notional)
    assert np.allclose(gotp["agg_gamma"], refp["gamma"].sum(0), rtol=1e-10, atol=1e-9)
    dtp.close()
    print(f"{P} pillars, {interp.name}: worst error {worst:.2e}")


@pytest.mark.parametrize("P,interp", [(70, InterpTypes.FLAT_FWD_RATES), (96, InterpTypes.LINEAR_ZERO_RATES),
                                      (96, InterpTypes.LINEAR_FWD_RATES), (123, InterpTypes.FLAT_FWD_RATES)])
def test_more_than_64_pillars_on_tiles_vs_c_oracle(gpu_ctx, P, interp):
    """65 pillars and more (the reference has no limit, engine.py:2388-2389): three and four 32-pillar tiles, one launch of the
    general kernel per tile pair; payment lag and spreads in the batch, every request mask, the aggregate, and the
    aggregate-only request (knot space, one projection over all tiles)."""
    vd = F.README_VALUE_DT
    px, tenors = many_pillar_quotes(P)
    curve = F.gbp_model(vd, interp, px=px, tenors=tenors).curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    assert dc.n_pillars == P
    batch = _mixed_batch(vd, 403, seed=P)
    dt = _native.DeviceTrades(gpu_ctx, batch)
    ref = port.price(interp.value, host.times, host.dfs, host.jac, host.hess, batch)
    got = _native.price(gpu_ctx, dc, dt, aggregate=True)
    assert got["delta"].shape == (403, P) and got["gamma"].shape == (403, P, P)
    worst = assert_batch_parity(got, ref, batch.notional)
    assert np.any(got["delta"][:, P - 20:] != 0.0) and np.any(got["gamma"][:, P - 20:, :40] != 0.0)       # the last tile, an off-diagonal tile
    asym = np.max(np.abs(got["gamma"] - np.swapaxes(got["gamma"], 1, 2)), axis=(1, 2))
    assert np.all(asym <= 1e-12 * np.max(np.abs(got["gamma"]), axis=(1, 2)) + 1e-14)
    assert np.allclose(got["agg_pv"], ref["pv"].sum(), rtol=1e-10, atol=1e-3)
    assert np.allclose(got["agg_delta"], ref["delta"].sum(0), rtol=1e-10, atol=1e-6)
    assert np.allclose(got["agg_gamma"], ref["gamma"].sum(0), rtol=1e-10, atol=1e-9)
    only_d = _native.price(gpu_ctx, dc, dt, want_gamma=False, aggregate=True)
    assert_batch_parity(only_d, dict(pv=ref["pv"], delta=ref["delta"]), batch.notional)
    assert np.allclose(only_d["agg_delta"], ref["delta"].sum(0), rtol=1e-10, atol=1e-6) and np.all(only_d["agg_gamma"] == 0.0)
    only_v = _native.price(gpu_ctx, dc, dt, want_delta=False, want_gamma=False, aggregate=True)
    assert_batch_parity(only_v, dict(pv=ref["pv"]), batch.notional)
    book = _native.price(gpu_ctx, dc, dt, per_trade=False, aggregate=True)           # Portfolio.compute's request
    assert np.allclose(book["agg_pv"], ref["pv"].sum(), rtol=1e-10, atol=1e-3)
    assert np.allclose(book["agg_delta"], ref["delta"].sum(0), rtol=1e-10, atol=1e-6)
    assert np.allclose(book["agg_gamma"], ref["gamma"].sum(0), rtol=1e-10, atol=1e-9)
    dt.close()
    print(f"{P} pillars, {interp.name}: worst error {worst:.2e}")


def test_tiled_route_still_serves_wide_curves(gpu_ctx):
    """ADR_CURVE_PILLAR_TILES at upload (adr_curve_upload_ex): the general kernel once per pair of 32-pillar tiles - the route
    of curves whose wide tables do not fit the LDS.  Same numbers as the wide route, to rounding."""
    vd = F.README_VALUE_DT
    px, tenors = forty_pillar_quotes()
    curve = F.gbp_model(vd, px=px, tenors=tenors).curves.GBP_OIS_SONIA
    batch = _mixed_batch(vd, 1501, seed=5)
    dt = _native.DeviceTrades(gpu_ctx, batch)
    host, dc_wide = _device_curve(gpu_ctx, curve)
    wide = _native.price(gpu_ctx, dc_wide, dt, aggregate=True)
    _, dc_tiled = _device_curve(gpu_ctx, curve, flags=_native.DeviceCurve.PILLAR_TILES)
    tiled = _native.price(gpu_ctx, dc_tiled, dt, aggregate=True)
    ref = port.price(4, host.times, host.dfs, host.jac, host.hess, batch)
    assert_batch_parity(tiled, ref, batch.notional)
    assert_batch_parity(wide, ref, batch.notional)
    assert np.allclose(tiled["agg_gamma"], wide["agg_gamma"], rtol=1e-12, atol=1e-9)
    dt.close()


def test_forty_pillar_curve_through_the_public_api(gpu_ctx):
    vd = F.README_VALUE_DT
    px, tenors = forty_pillar_quotes()
    model = F.gbp_model(vd, px=px, tenors=tenors)
    swap = F.make_swap(vd, "33Y", 0.041, 2e7)
    res = swap.position(model).compute([RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA])
    assert len(res.risk.risk_ladder) == 40 and res.gamma.risk_ladder.shape == (40, 40)
    assert res.risk.ladder.data["35Y"] != 0.0 and res.risk.ladder.data["40Y"] == 0.0
    bumped = swap.position(model.scenario("GBP_OIS_SONIA", 0.01)).compute([RequestTypes.VALUE]).value.amount
    down = swap.position(model.scenario("GBP_OIS_SONIA", -0.01)).compute([RequestTypes.VALUE]).value.amount
    assert abs(res.risk.value.amount - (bumped - down) / 2.0) / abs(res.risk.value.amount) < 1e-4     # the reference's tolerance at 1 bp
    # every calibration swap of the 40-pillar curve reprices through the engine grid
    for tenor, p in zip(tenors, px):
        v = F.make_swap(vd, tenor, p / 100, 1e6).position(model).compute([RequestTypes.VALUE]).value.amount
        assert abs(v) <= 1e-5, (tenor, v)
    # more than 64 pillars: 32-pillar tiles (test_more_than_64_pillars_on_tiles_vs_c_oracle); the limit is ADR_MAX_PILLARS
    big_t = [f"{k}M" for k in range(1, 13)] + [f"{k}Y" for k in range(2, 60)]      # 70 pillars
    big = F.gbp_model(vd, px=[4.0 + 0.001 * i for i in range(len(big_t))], tenors=big_t)
    r70 = F.make_swap(vd, "45Y", 0.04).position(big).compute([RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA])
    assert len(r70.risk.risk_ladder) == 70 and r70.gamma.risk_ladder.shape == (70, 70) and r70.risk.ladder.data["45Y"] != 0.0
    K = _native.MAX_PILLARS + 2
    with pytest.raises(Exception, match="pillar"):
        _native.DeviceCurve(gpu_ctx, 4, np.arange(K, dtype=float), np.exp(-0.03 * np.arange(K)), np.zeros((K, K - 1)))


def test_seventeen_pillar_curve_all_requests(gpu_ctx):
    """Odd pillar count inside the library: gamma on the general kernel, delta on the lite kernel, same numbers."""
    vd = F.README_VALUE_DT
    px, tenors = list(F.GBP_PX[8:9] + F.GBP_PX[14:30]), list(F.TENORS[8:9] + F.TENORS[14:30])      # 6M, 1Y ... 30Y
    curve = F.gbp_model(px=px, tenors=tenors).curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    assert dc.n_pillars == 17
    batch = synthetic.synthesize(vd, 4001, seed=3)
    dt = _native.DeviceTrades(gpu_ctx, batch)
    ref = port.price(4, host.times, host.dfs, host.jac, host.hess, batch)
    got = _native.price(gpu_ctx, dc, dt, aggregate=True)
    assert got["gamma"].shape == (4001, 17, 17)
    assert_batch_parity(got, ref, batch.notional)
    assert np.allclose(got["agg_gamma"], ref["gamma"].sum(0), rtol=1e-10, atol=1e-9)
    only_d = _native.price(gpu_ctx, dc, dt, want_gamma=False)
    assert_batch_parity(only_d, dict(pv=ref["pv"], delta=ref["delta"]), batch.notional)
    dt.close()


def test_wide_route_aggregate_without_per_trade_gamma_and_two_streams(gpu_ctx):
    """`adr_price_dev` on a 40-pillar curve with the aggregate ladder alone (no per-trade gamma buffer: the kernel's
    staging / store path is skipped, the totals are the same), and two batches priced concurrently on two streams of one
    ctx without aggregates - bitwise the serial results."""
    import torch
    vd = F.README_VALUE_DT
    px, tenors = forty_pillar_quotes()
    curve = F.gbp_model(vd, px=px, tenors=tenors).curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    P = dc.n_pillars
    dev = torch.device("cuda", 0)
    batches = [_mixed_batch(vd, 4001, seed=21), _mixed_batch(vd, 3001, seed=22)]
    dts = [_native.DeviceTrades(gpu_ctx, b) for b in batches]
    full = [_native.price(gpu_ctx, dc, dt, aggregate=True) for dt in dts]
    # aggregate only
    n = batches[0].n_trades
    pv = torch.empty(n, dtype=torch.float64, device=dev); de = torch.empty((n, P), dtype=torch.float64, device=dev)
    ag = torch.zeros(1 + P + P * P, dtype=torch.float64, device=dev)
    _native.price_dev(gpu_ctx, dc, dts[0], 7, pv.data_ptr(), de.data_ptr(), 0, ag.data_ptr())
    gpu_ctx.sync()
    agg = ag.cpu().numpy()
    assert np.array_equal(agg[1 + P:].reshape(P, P), full[0]["agg_gamma"])       # same totals, same order of summation
    assert np.array_equal(agg[1:1 + P], full[0]["agg_delta"]) and agg[0] == full[0]["agg_pv"]
    assert np.array_equal(de.cpu().numpy(), full[0]["delta"])
    # two streams, no aggregates
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    outs = []
    for b, dt, st in zip(batches, dts, streams):
        m = b.n_trades
        bufs = (torch.empty(m, dtype=torch.float64, device=dev), torch.empty((m, P), dtype=torch.float64, device=dev),
                torch.empty((m, P, P), dtype=torch.float64, device=dev))
        outs.append(bufs)
    for rep in range(3):
        for dt, st, bufs in zip(dts, streams, outs):
            _native.price_dev(gpu_ctx, dc, dt, 7, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), 0, st.cuda_stream)
    torch.cuda.synchronize()
    for bufs, ref in zip(outs, full):
        assert np.array_equal(bufs[2].cpu().numpy(), ref["gamma"]) and np.array_equal(bufs[0].cpu().numpy(), ref["pv"])
    for dt in dts:
        dt.close()


@pytest.mark.parametrize("drop,interp", [(13, InterpTypes.LINEAR_ZERO_RATES), (31, InterpTypes.FLAT_FWD_RATES), (0, InterpTypes.LINEAR_FWD_RATES)])
def test_odd_pillar_count_on_the_fast_kernels(gpu_ctx, drop, interp):
    """31 pillars (one README pillar dropped: the first, a middle one, the last): the packed layout and the fast kernels take odd
    pillar counts - the matrices of odd-numbered trades start on 8-byte boundaries (16-byte stores there are legal on gfx950)
    and the last element of a matrix, whose pair of the flat array would reach into the next trade's matrix, is stored on its
    own.  Plain, payment-lag and 80-coupon trades; every neighbour of a matrix's last element is checked by the batch parity."""
    vd = F.README_VALUE_DT
    tenors = [t for i, t in enumerate(F.TENORS) if i != drop]
    px = [p for i, p in enumerate(F.GBP_PX) if i != drop]
    curve = F.gbp_model(vd, interp, px=px, tenors=tenors).curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    assert dc.n_pillars == 31
    assert _native.curve_layout_host(host.times, host.dfs, host.jac, host.hess)["packed_ok"] == 1
    for batch in (synthetic.synthesize(vd, 5003, seed=drop), _mixed_batch(vd, 3001, seed=drop + 1)):
        dt = _native.DeviceTrades(gpu_ctx, batch)
        ref = port.price(interp.value, host.times, host.dfs, host.jac, host.hess, batch)
        got = _native.price(gpu_ctx, dc, dt, aggregate=True)
        assert_batch_parity(got, ref, batch.notional)
        assert np.allclose(got["agg_gamma"], ref["gamma"].sum(0), rtol=1e-10, atol=1e-9)
        only_d = _native.price(gpu_ctx, dc, dt, want_gamma=False)
        assert_batch_parity(only_d, dict(pv=ref["pv"], delta=ref["delta"]), batch.notional)
        dt.close()
