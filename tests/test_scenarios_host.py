"""Host logic of the scenario grid (no GPU): shocked quotes follow Model.scenario, ladders are well formed."""
import numpy as np

from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.market.position.scenarios import bump_ladder, finite_difference_delta, shocked_quotes

from . import _fixtures as F


def test_shocked_quotes_match_model_scenario():
    model = F.gbp_model()
    params = model._curve_params_dict["GBP_OIS_SONIA"]
    for shock in (0.1, -0.25, {"5Y": 0.07, "10Y": -0.03, "not-a-tenor": 1.0}):
        shocked = model.scenario("GBP_OIS_SONIA", shock)
        want = shocked._curve_params_dict["GBP_OIS_SONIA"]["px_list"]
        assert shocked_quotes(params["px_list"], params["tenor_list"], shock) == want
        # and the shocked curve's par rates are quote / 100, the division ScenarioGrid repeats
        assert [q / 100.0 for q in want] == list(shocked.curves.GBP_OIS_SONIA.swap_rates)


def test_shock_leaves_the_knot_grid_unchanged():
    model = F.gbp_model()
    base = model.curves.GBP_OIS_SONIA
    shocked = model.scenario("GBP_OIS_SONIA", {"2Y": 0.2, "30Y": -0.1}).curves.GBP_OIS_SONIA
    a = build_engine_curve(base.swap_rates, base.swap_times, base.year_fracs, with_hessian=False)
    b = build_engine_curve(shocked.swap_rates, shocked.swap_times, shocked.year_fracs, with_hessian=False)
    for name in ("times", "acc", "pillar", "prev_idx"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert not np.array_equal(a.dfs, b.dfs)


def test_bump_ladder_and_central_differences():
    tenors = ["1Y", "2Y", "5Y"]
    shocks = bump_ladder(tenors, 2.0)
    assert shocks[0] == 0.0 and len(shocks) == 7
    assert shocks[1] == {"1Y": 0.02} and shocks[2] == {"1Y": -0.02} and shocks[5] == {"5Y": 0.02}
    # values linear in the bumps: PV = sum_k w_k * shift_k for two trades
    w = np.array([[3.0, -1.0, 0.5], [0.0, 2.0, 4.0]])
    vals = np.zeros((7, 2))
    for k in range(3):
        vals[1 + 2 * k] = w[:, k] * 2.0
        vals[2 + 2 * k] = -w[:, k] * 2.0
    assert np.allclose(finite_difference_delta(vals, 2.0), w)
