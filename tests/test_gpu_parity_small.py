"""HIP path vs the torch.func oracle on trades built through the public API (small cases)."""
import numpy as np
import pytest

from adrates_amd import _native
from adrates_amd.utils import DayCountTypes, FrequencyTypes, InterpTypes

from . import _fixtures as F
from ._parity import assert_parity, gpu_price, oracle_price

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("interp", [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_FWD_RATES])
def test_readme_curve_mixed_trades(gpu_ctx, interp):
    vd = F.README_VALUE_DT
    model = F.gbp_model(vd, interp)
    curve = model.curves.GBP_OIS_SONIA
    swaps = [F.make_swap(vd, "10Y", 0.045, 1e7),                    # README trade
             F.make_swap(vd, "87M", 0.04, 1e7, pay=False),          # off-grid, front stub
             F.make_swap(vd, "1W", 0.052014, 1e6),                  # notebook KAT
             F.make_swap(vd, "3M", 0.05, 2e6),
             F.make_swap(vd, "50Y", 0.039, 5e6, pay=False),
             F.make_swap(vd, "55Y", 0.039, 5e6),                    # beyond the last knot
             F.make_swap(vd, "7Y", 0.03, 1e6, spread=0.0025),       # float spread
             F.make_swap(vd, "5Y", 0.04, 1e6, fixed_freq=FrequencyTypes.SEMI_ANNUAL),
             F.make_swap(vd, "4Y", 0.04, 3e6, float_freq=FrequencyTypes.QUARTERLY, pay=False),
             F.make_swap(vd, "6Y", 0.04, 1e6, float_dc=DayCountTypes.THIRTY_E_360),   # OIS default float dc
             F.make_swap(vd, "3Y", 0.04, 1e6, payment_lag=2),       # te != tp: ratio terms
             F.make_swap(vd, "30M", 0.045, 4e6, payment_lag=1, spread=0.001, pay=False)]
    got = gpu_price(gpu_ctx, curve, swaps, vd, aggregate=True)
    refs = oracle_price(curve, swaps, vd)
    worst = assert_parity(got, refs, [s._notional for s in swaps])
    # aggregate = sum of the per-trade results
    assert np.allclose(got["agg_pv"], got["pv"].sum(), rtol=1e-13, atol=1e-6)
    assert np.allclose(got["agg_delta"], got["delta"].sum(0), rtol=1e-12, atol=1e-9)
    assert np.allclose(got["agg_gamma"], got["gamma"].sum(0), rtol=1e-12, atol=1e-12)
    print("worst", worst)


def test_cashflows_next_to_gpu_requests(gpu_ctx):
    """VALUE/DELTA come from the kernels (engine knot grid), CASHFLOWS from the curve's own nodes; the two
    curve constructions agree on the calibration instruments, so the totals are close but not identical."""
    from adrates_amd.utils import RequestTypes
    vd = F.README_VALUE_DT
    model = F.gbp_model(vd)
    swap = F.make_swap(vd, "10Y", 0.045, 1e7)
    res = swap.position(model).compute([RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.CASHFLOWS])
    assert res.gamma is None and len(res.cashflows) == 20 and len(res.risk.risk_ladder) == 32
    assert abs(res.cashflows.total_pv - res.value.amount) < 1e-4 * swap._notional


@pytest.mark.parametrize("shape", ["core21_epg12", "core16_epg7"])
def test_other_curve_shapes_use_other_kernel_variants(gpu_ctx, shape):
    """32-pillar curves whose long end is longer / shorter than the README curve's: a different core size picks a
    different instantiation of the fast kernel (12 or 7 packed entries per lane; hub layout with 10 or 5 core
    slots).  Same parity bar."""
    from adrates_amd import _native
    from adrates_amd.market.curves.curve_tables import build_engine_curve
    vd = F.README_VALUE_DT
    if shape == "core21_epg12":
        tenors = ["1W", "2W", "1M", "2M", "3M", "4M", "5M", "6M", "7M", "8M", "9M", "1Y"] \
            + [f"{y}Y" for y in range(2, 19)] + ["20Y", "25Y", "30Y"]
        px = [5.20 - 0.02 * i for i in range(12)] + [4.6 - 0.03 * i for i in range(20)]
        want = (21, 12)
    else:
        tenors = ["1W", "2W", "1M", "2M", "3M", "4M", "5M", "6M", "7M", "8M", "9M", "10M", "11M", "1Y", "15M",
                  "18M", "21M"] + [f"{y}Y" for y in range(2, 13)] + ["15Y", "20Y", "25Y", "30Y"]
        px = [5.20 - 0.02 * i for i in range(17)] + [4.6 - 0.03 * i for i in range(15)]
        want = (16, 7)
    model = F.gbp_model(vd, px=px, tenors=tenors)
    curve = model.curves.GBP_OIS_SONIA
    host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    info = _native.curve_layout_host(host.times, host.dfs, host.jac, host.hess)
    assert info["packed_ok"] == 1 and (info["core_pillars"], info["entries_per_lane"]) == want
    assert info["lds_bytes"] <= 160 * 1024
    swaps = [F.make_swap(vd, "10Y", 0.045, 1e7), F.make_swap(vd, "87M", 0.04, 1e7, pay=False),
             F.make_swap(vd, "3M", 0.05, 2e6), F.make_swap(vd, "29Y", 0.039, 5e6),
             F.make_swap(vd, "14M", 0.05, 3e6, pay=False), F.make_swap(vd, "17Y", 0.04, 1e6, spread=0.002),
             F.make_swap(vd, "5Y", 0.04, 1e6, fixed_freq=FrequencyTypes.SEMI_ANNUAL)]
    got = gpu_price(gpu_ctx, curve, swaps, vd, aggregate=True)
    assert_parity(got, oracle_price(curve, swaps, vd), [s._notional for s in swaps])
    assert np.allclose(got["agg_gamma"], got["gamma"].sum(0), rtol=1e-12, atol=1e-12)


def test_even_pillar_count_below_32_uses_the_fast_kernel(gpu_ctx):
    """A 20-pillar curve: [20][20] matrices stored as 16-byte pairs of the flat array, the rest of each
    1 KB band goes to the sink.  An odd pillar count (tests/test_gpu_parity_batch.py, 5 pillars) stays on the
    general kernel."""
    from adrates_amd import _native
    from adrates_amd.market.curves.curve_tables import build_engine_curve
    vd = F.README_VALUE_DT
    tenors = ["1W", "1M", "2M", "3M", "6M", "9M", "1Y", "18M"] + [f"{y}Y" for y in range(2, 11)] + ["15Y", "20Y", "30Y"]
    px = [5.20 - 0.02 * i for i in range(8)] + [4.7 - 0.04 * i for i in range(12)]
    curve = F.gbp_model(vd, px=px, tenors=tenors).curves.GBP_OIS_SONIA
    host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    info = _native.curve_layout_host(host.times, host.dfs, host.jac, host.hess)
    assert host.n_pillars == 20 and info["packed_ok"] == 1
    swaps = [F.make_swap(vd, "10Y", 0.045, 1e7), F.make_swap(vd, "87M", 0.04, 1e7, pay=False),
             F.make_swap(vd, "3M", 0.05, 2e6), F.make_swap(vd, "29Y", 0.039, 5e6),
             F.make_swap(vd, "14M", 0.05, 3e6, pay=False), F.make_swap(vd, "35Y", 0.04, 1e6),
             F.make_swap(vd, "4Y", 0.04, 1e6, payment_lag=2)]                 # general kernel, same batch
    got = gpu_price(gpu_ctx, curve, swaps, vd, aggregate=True)
    assert got["gamma"].shape == (7, 20, 20) and got["delta"].shape == (7, 20)
    assert_parity(got, oracle_price(curve, swaps, vd), [s._notional for s in swaps])
    assert np.allclose(got["agg_gamma"], got["gamma"].sum(0), rtol=1e-12, atol=1e-12)
    assert np.allclose(got["agg_delta"], got["delta"].sum(0), rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("interp", [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_FWD_RATES])
def test_seasoned_and_expired_trades(gpu_ctx, interp):
    """Trades that started before the value date: past payments are masked (float: payment time >= 0, fixed:
    > 0, engine.py:2437, :2695), the running coupon's accrual start lies left of the first knot, and a fully
    expired trade is worth nothing."""
    vd = F.README_VALUE_DT
    curve = F.gbp_model(vd, interp).curves.GBP_OIS_SONIA
    swaps = [F.make_swap(vd.add_months(-6), "5Y", 0.04, 1e7),
             F.make_swap(vd.add_months(-18), "3Y", 0.045, 5e6, pay=False, spread=0.001),
             F.make_swap(vd.add_years(-5), "12Y", 0.03, 2e6, float_freq=FrequencyTypes.SEMI_ANNUAL),
             F.make_swap(vd.add_months(-30), "10Y", 0.035, 1e6, payment_lag=2),     # general kernel
             F.make_swap(vd.add_years(-3), "2Y", 0.05, 1e6)]                        # expired
    got = gpu_price(gpu_ctx, curve, swaps, vd)
    assert_parity(got, oracle_price(curve, swaps, vd), [s._notional for s in swaps])
    assert got["pv"][4] == 0.0 and not got["delta"][4].any() and not got["gamma"][4].any()


def test_linear_fwd_rates_through_the_python_api_and_scenario_grid(gpu_ctx):
    """LINEAR_FWD_RATES (interpolator_ad.py:234-235, linear in the knot DFs): `position(model).compute()` and a
    scenario grid on such a curve - the general kernel with the extra rank-one Hessian term of a linear DF."""
    from adrates_amd.market.position.scenarios import ScenarioGrid
    from adrates_amd.utils import RequestTypes
    from adrates_amd.utils.helpers import times_from_dates
    from oracle import cavour_oracle as O
    vd = F.README_VALUE_DT
    model = F.gbp_model(vd, InterpTypes.LINEAR_FWD_RATES)
    curve = model.curves.GBP_OIS_SONIA
    swaps = [F.make_swap(vd, "87M", 0.04, 1e7, pay=False), F.make_swap(vd, "3Y", 0.04, 1e6, payment_lag=2)]
    cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    for swap in swaps:
        res = swap.position(model).compute([RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA])
        fx, fl = O.leg_inputs_from_swap(swap, vd, times_from_dates)
        want = O.ois_analytics(cache, curve._interp_type.value, fx, fl)
        n = swap._notional
        assert abs(res.value.amount - want["value"]) <= 1e-10 * n
        assert np.max(np.abs(res.risk.risk_ladder - want["delta"])) <= 1e-10 * n * 1e-4
        assert np.max(np.abs(res.gamma.risk_ladder - want["gamma"])) <= 1e-10 * n * 1e-6
    # the linear scheme really is a different function of the knots
    lzr = swaps[0].position(F.gbp_model(vd, InterpTypes.LINEAR_ZERO_RATES)).compute([RequestTypes.VALUE]).value.amount
    assert abs(lzr - swaps[0].position(model).compute([RequestTypes.VALUE]).value.amount) > 1.0
    grid = ScenarioGrid(model, "GBP_OIS_SONIA", [0.0, {"10Y": 0.01}], with_gamma=True)
    out = grid.price(swaps, [RequestTypes.VALUE, RequestTypes.DELTA])
    base = swaps[0].position(model).compute([RequestTypes.VALUE, RequestTypes.DELTA])
    assert abs(out["pv"][0, 0] - base.value.amount) <= 1e-10 * swaps[0]._notional
    bumped = swaps[0].position(model.scenario("GBP_OIS_SONIA", {"10Y": 0.01})).compute([RequestTypes.VALUE])
    assert abs(out["pv"][1, 0] - bumped.value.amount) <= 1e-10 * swaps[0]._notional
    grid.close()


def test_c_example_matches_the_python_path(gpu_ctx, tmp_path):
    """examples/c_abi_example.c (plain C through include/adrates.h) prints the same numbers `_native.price` returns
    for the same toy curve and trades, and its first PV equals the closed form N[(1 - d2) - c(d1 + d2)]."""
    import subprocess
    from adrates_amd.trades.compiler import TradeBatch
    from tests.test_capi_library import _build_c_example
    out = subprocess.run([_build_c_example(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    r1, r2 = 0.04, 0.045
    d1 = 1.0 / (1.0 + r1); d2 = (1.0 - r2 * d1) / (1.0 + r2)
    dd1 = -d1 * d1
    jac = np.array([[0.0, 0.0], [dd1, 0.0], [-r2 * dd1 / (1 + r2), (-d1 * (1 + r2) - (1 - r2 * d1)) / (1 + r2) ** 2]])
    hess = np.zeros((3, 2, 2))
    hess[1, 0, 0] = 2 * d1 ** 3
    hess[2] = [[-r2 * 2 * d1 ** 3 / (1 + r2), -dd1 / (1 + r2) ** 2], [-dd1 / (1 + r2) ** 2, 2 * (1 + d1) / (1 + r2) ** 3]]
    dc = _native.DeviceCurve(gpu_ctx, 1, np.array([0.0, 1.0, 2.0]), np.array([1.0, d1, d2]), jac, hess)
    batch = TradeBatch(np.array([0, 2, 3]), np.array([0, 2, 3]), np.array([1.0, 2.0, 1.0]),
                       np.array([0.042e7, 0.042e7, 0.039 * 5e6]), np.array([1.0, 2.0, 1.0]), np.array([0.0, 1.0, 0.0]),
                       np.array([1.0, 2.0, 1.0]), np.ones(3), np.array([1e7, 5e6]), np.zeros(2), np.array([-1.0, 1.0]),
                       np.array([1.0, -1.0]))
    want = _native.price(gpu_ctx, dc, _native.DeviceTrades(gpu_ctx, batch), aggregate=True)
    text = out.stdout.splitlines()
    for t in range(2):
        nums = [float(x) for x in text[t].replace("trade %d" % t, "").split() if x not in ("pv", "delta", "gamma")]
        assert nums[0] == want["pv"][t] and nums[1:3] == list(want["delta"][t]) and nums[3:7] == list(want["gamma"][t].reshape(-1))
    book = [float(x) for x in text[2].split() if x not in ("book", "pv", "delta")]
    assert book[0] == want["agg_pv"] and book[1:3] == list(want["agg_delta"])
    closed = float(text[3].split()[-1])
    assert abs(want["pv"][0] - closed) <= 1e-9 and abs(closed - 1e7 * ((1 - d2) - 0.042 * (d1 + d2))) <= 1e-9


@pytest.mark.parametrize("n_pillars", [4, 10])
def test_small_annual_curves(gpu_ctx, n_pillars):
    """Few-pillar curves: 4 annual pillars stay below the packed layout's minimum core (general kernel), 10 are the
    smallest curve the fast kernel takes."""
    vd = F.README_VALUE_DT
    tenors = [f"{i}Y" for i in range(1, n_pillars + 1)]
    model = F.gbp_model(vd, InterpTypes.FLAT_FWD_RATES, px=[4.0 + 0.05 * i for i in range(n_pillars)], tenors=tenors)
    curve = model.curves.GBP_OIS_SONIA
    from adrates_amd.market.curves.curve_tables import build_engine_curve
    host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    assert _native.curve_layout_host(host.times, host.dfs, host.jac, host.hess)["packed_ok"] == (1 if n_pillars == 10 else 0)
    swaps = [F.make_swap(vd, "1Y", 0.04, 1e6), F.make_swap(vd, "30M", 0.045, 5e6, pay=False),
             F.make_swap(vd, f"{n_pillars}Y", 0.043, 1e7), F.make_swap(vd, f"{n_pillars + 2}Y", 0.043, 2e6, pay=False),
             F.make_swap(vd, "3Y", 0.04, 1e6, payment_lag=2), F.make_swap(vd.add_months(-5), "2Y", 0.04, 3e6)]
    got = gpu_price(gpu_ctx, curve, swaps, vd, aggregate=True)
    refs = oracle_price(curve, swaps, vd)
    assert_parity(got, refs, [s._notional for s in swaps])
    assert np.allclose(got["agg_gamma"], got["gamma"].sum(0), rtol=1e-12, atol=1e-12)
