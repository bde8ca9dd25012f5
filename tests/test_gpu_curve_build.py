"""Device curve builder (csrc/curve_build.hip) vs the host builder and the torch.func oracle, and the
scenario grid vs one-model-per-shock pricing (the reference's Model.scenario route)."""
import numpy as np
import pytest

from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.market.position.scenarios import ScenarioGrid, bump_ladder, finite_difference_delta
from adrates_amd.utils import InterpTypes, RequestTypes
from oracle import cavour_oracle as O

from . import _fixtures as F
from ._parity import REL_TOL, assert_parity, gpu_price, oracle_price

pytestmark = pytest.mark.gpu


def _scenario_rates(curve, n, seed=7):
    rng = np.random.default_rng(seed)
    base = np.array(curve.swap_rates, dtype=np.float64)
    shifts = rng.uniform(-25e-4, 25e-4, size=(n, base.size))      # up to +/- 25 bp per pillar
    shifts[0] = 0.0
    return base[None, :] + shifts


@pytest.mark.parametrize("model_fn,name", [(F.gbp_model, "GBP_OIS_SONIA"), (F.usd_model, "USD_OIS_SOFR")])
def test_device_bootstrap_equals_host_builder(gpu_ctx, model_fn, name):
    curve = getattr(model_fn().curves, name)
    base = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    rates = _scenario_rates(curve, 5)
    plan = _native.CurvePlan(gpu_ctx, curve._interp_type.value, base)
    cset = plan.build(rates)
    try:
        for i in range(len(cset)):
            host = build_engine_curve(list(rates[i]), curve.swap_times, curve.year_fracs)
            dfs, jac, hess = cset.download(i)
            # same IEEE operations in the same order (fp contraction is off in the kernel): bit for bit
            assert np.array_equal(dfs, host.dfs)
            assert np.array_equal(jac, host.jac)
            assert np.array_equal(hess, host.hess)
    finally:
        cset.close()
        plan.close()


def test_device_bootstrap_matches_oracle_autodiff(gpu_ctx):
    """The oracle differentiates the reference's scan with torch.func (jacrev / hessian)."""
    curve = F.gbp_model().curves.GBP_OIS_SONIA
    base = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    rates = _scenario_rates(curve, 2, seed=11)
    plan = _native.CurvePlan(gpu_ctx, curve._interp_type.value, base)
    cset = plan.build(rates)
    try:
        dfs, jac, hess = cset.download(1)
        cache = O.cached_curve(list(rates[1]), curve.swap_times, curve.year_fracs, derivatives=True)
        assert np.allclose(dfs, np.asarray(cache["dfs"]), rtol=1e-14, atol=0)
        assert np.allclose(jac, np.asarray(cache["jac"]), rtol=1e-12, atol=1e-14)
        assert np.allclose(hess, np.asarray(cache["hess"]), rtol=1e-11, atol=1e-12)
    finally:
        cset.close()
        plan.close()


@pytest.mark.parametrize("interp", [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES])
def test_scenario_curve_prices_like_uploaded_curve_and_oracle(gpu_ctx, interp):
    vd = F.README_VALUE_DT
    model = F.gbp_model(vd, interp)
    swaps = [F.make_swap(vd, "10Y", 0.045, 1e7), F.make_swap(vd, "87M", 0.04, 1e7, pay=False),
             F.make_swap(vd, "3M", 0.05, 2e6), F.make_swap(vd, "30Y", 0.039, 5e6),
             F.make_swap(vd, "3Y", 0.04, 1e6, payment_lag=2)]          # last one: general kernel
    shocks = [0.0, 0.10, {"5Y": -0.07, "10Y": 0.03}, -0.25]
    grid = ScenarioGrid(model, "GBP_OIS_SONIA", shocks, ctx=gpu_ctx)
    try:
        got = grid.price(swaps, [RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA])
        for i, shock in enumerate(shocks):
            shocked = model.scenario("GBP_OIS_SONIA", shock)         # the reference's route: a model per shock
            curve_i = shocked.curves.GBP_OIS_SONIA
            one = {k: got[k][i] for k in ("pv", "delta", "gamma")}
            # (a) oracle on the shocked model
            assert_parity(one, oracle_price(curve_i, swaps, vd), [s._notional for s in swaps])
            # (b) the same kernels on tables built on the host and uploaded
            ref = gpu_price(gpu_ctx, curve_i, swaps, vd)
            for k in ("pv", "delta", "gamma"):
                scale = np.maximum(np.abs(ref[k]).max(), 1e-12)
                assert np.max(np.abs(one[k] - ref[k])) <= 1e-13 * scale, (k, i)
    finally:
        grid.close()


def test_finite_difference_ladder_on_device(gpu_ctx):
    """tests/test_ois_request_types.py:171-207 in one batch: per-tenor central differences of PV over a
    65-curve grid agree with the analytic delta ladder."""
    vd = F.README_VALUE_DT
    model = F.gbp_model(vd)
    tenors = model._curve_params_dict["GBP_OIS_SONIA"]["tenor_list"]
    swaps = [F.make_swap(vd, "10Y", 0.045, 1e7), F.make_swap(vd, "87M", 0.04, 1e7, pay=False),
             F.make_swap(vd, "2Y", 0.05, 1e6)]
    grid = ScenarioGrid(model, "GBP_OIS_SONIA", bump_ladder(tenors, 1.0), with_gamma=False, ctx=gpu_ctx)
    try:
        assert len(grid) == 2 * len(tenors) + 1
        pv = grid.price(swaps, [RequestTypes.VALUE])["pv"]                 # [65, 3]
        fd = finite_difference_delta(pv, 1.0)                               # [3, 32]
        ad = grid.price(swaps, [RequestTypes.VALUE, RequestTypes.DELTA])["delta"][0]
        for t in range(len(swaps)):
            scale = np.abs(ad[t]).max()
            assert np.max(np.abs(fd[t] - ad[t])) <= 2e-5 * scale       # O(h^2) truncation of a 1 bp bump
    finally:
        grid.close()


def test_plan_rejects_bad_inputs(gpu_ctx):
    curve = F.gbp_model().curves.GBP_OIS_SONIA
    base = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs, with_hessian=False)
    plan = _native.CurvePlan(gpu_ctx, curve._interp_type.value, base)
    try:
        with pytest.raises(Exception):
            plan.build(np.zeros((2, 31)))                                  # wrong pillar count
        bad = np.array([curve.swap_rates], dtype=np.float64)
        bad[0, 3] = np.nan
        with pytest.raises(Exception):
            plan.build(bad)
        cset = plan.build(np.array([curve.swap_rates]))
        dfs, jac, hess = cset.download(0)
        assert hess is None and np.array_equal(dfs, base.dfs) and np.array_equal(jac, base.jac)
        cset.close()
    finally:
        plan.close()


def test_price_dev_calls_can_be_captured_into_a_hip_graph():
    """adr_price_dev neither allocates nor synchronises: a bump ladder of pricing calls captured on a stream into a
    HIP graph (torch.cuda.CUDAGraph) replays to the same bits as the eager launches."""
    import torch
    from adrates_amd import _native
    from adrates_amd.market.position.scenarios import ScenarioGrid, bump_ladder
    from adrates_amd.trades import synthetic
    from tests._fixtures import README_VALUE_DT, TENORS, readme_model
    grid = ScenarioGrid(readme_model(), "GBP_OIS_SONIA", bump_ladder(TENORS[:6], 1.0), with_gamma=True)
    n, S, P = 500, len(grid), 32
    trades = _native.DeviceTrades(grid._ctx, synthetic.synthesize(README_VALUE_DT, n, seed=5))
    dev = torch.device("cuda", 0)
    pv = torch.zeros((S, n), dtype=torch.float64, device=dev)
    delta = torch.zeros((S, n, P), dtype=torch.float64, device=dev)
    agg = torch.zeros((S, 1 + P + P * P), dtype=torch.float64, device=dev)
    stream = torch.cuda.Stream(dev)

    def launch_all():
        for i in range(S):
            _native.price_dev(grid._ctx, grid.device_curve(i), trades, 7, pv[i].data_ptr(), delta[i].data_ptr(), 0,
                              agg[i].data_ptr(), stream.cuda_stream)

    with torch.cuda.stream(stream):
        launch_all()
        stream.synchronize()
        eager = (pv.clone(), delta.clone(), agg.clone())
        pv.zero_(); delta.zero_(); agg.zero_()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            launch_all()
        graph.replay()
        stream.synchronize()
    assert torch.equal(pv, eager[0]) and torch.equal(delta, eager[1]) and torch.equal(agg, eager[2])
    assert float(pv.abs().max()) > 0.0
    trades.close(); grid.close()


@pytest.mark.parametrize("P", [40, 64])
def test_device_bootstrap_of_wide_curves(gpu_ctx, P):
    """33-64 pillars (the reference has no pillar limit, engine.py:2388-2389): the device builder's knot values and
    derivatives equal the host builder's bit for bit - at 64 pillars the PV01 gradients of the 1 242-knot grid go through a
    scratch buffer instead of LDS -, and a built curve (the wide layout's tables only) prices a mixed batch like the same
    rates uploaded from the host, and like the C oracle."""
    from oracle import port
    from .test_gpu_many_pillars import _mixed_batch, many_pillar_quotes
    from ._parity import assert_batch_parity
    vd = F.README_VALUE_DT
    px, tenors = many_pillar_quotes(P)
    curve = F.gbp_model(vd, px=px, tenors=tenors).curves.GBP_OIS_SONIA
    base = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    rates = _scenario_rates(curve, 3, seed=P)
    plan = _native.CurvePlan(gpu_ctx, curve._interp_type.value, base)
    cset = plan.build(rates)
    batch = _mixed_batch(vd, 1203, seed=P + 1)
    dt = _native.DeviceTrades(gpu_ctx, batch)
    try:
        for i in range(len(cset)):
            host = build_engine_curve(list(rates[i]), curve.swap_times, curve.year_fracs)
            dfs, jac, hess = cset.download(i)
            assert np.array_equal(dfs, host.dfs)
            assert np.array_equal(jac, host.jac)
            assert np.array_equal(hess, host.hess)
        host = build_engine_curve(list(rates[2]), curve.swap_times, curve.year_fracs)
        uploaded = _native.DeviceCurve(gpu_ctx, curve._interp_type.value, host.times, host.dfs, host.jac, host.hess)
        a = _native.price(gpu_ctx, cset[2], dt, aggregate=True)
        b = _native.price(gpu_ctx, uploaded, dt, aggregate=True)
        ref = port.price(curve._interp_type.value, host.times, host.dfs, host.jac, host.hess, batch)
        assert_batch_parity(a, ref, batch.notional)
        for key in ("pv", "delta", "gamma", "agg_gamma"):
            scale = np.max(np.abs(b[key])) + 1e-300
            assert np.max(np.abs(a[key] - b[key])) <= 1e-13 * scale, key      # tables differ by the libm log only
        only_d = _native.price(gpu_ctx, cset[1], dt, want_gamma=False)
        host1 = build_engine_curve(list(rates[1]), curve.swap_times, curve.year_fracs)
        ref1 = port.price(curve._interp_type.value, host1.times, host1.dfs, host1.jac, host1.hess, batch)
        assert_batch_parity(only_d, dict(pv=ref1["pv"], delta=ref1["delta"]), batch.notional)
    finally:
        dt.close()
        cset.close()
        plan.close()
