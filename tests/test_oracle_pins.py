"""Pins the CPU oracle to everything the reference offers for this path:

* the notebook's stored outputs (notebooks/intro.ipynb cells 12, 25-44),
* the properties asserted by the reference's own tests (tests/test_refit_curves.py:152-231,
  tests/test_ois_request_types.py:429-524, 707-753, 841-905, 908-942),
* the committed golden vectors (regression pin of the oracle itself).
"""
import json
import os

import numpy as np
import pytest

from adrates_amd.utils import InterpTypes
from adrates_amd.utils.helpers import times_from_dates
from oracle import cavour_oracle as O

from . import _fixtures as F
from .golden import make_golden

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ois_golden.json")


@pytest.fixture(scope="module")
def readme():
    model = F.readme_model()
    curve = model.curves.GBP_OIS_SONIA
    cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    return model, curve, cache


def _analytics(curve, cache, swap, vd, gamma=True):
    fx, fl = O.leg_inputs_from_swap(swap, vd, times_from_dates)
    return O.ois_analytics(cache, curve._interp_type.value, fx, fl, want_gamma=gamma)


def test_notebook_known_answers(readme):
    """1W GBP OIS @5.2014 %, N = 1e6, 30-Apr-2024 (cells 23-44)."""
    _, curve, cache = readme
    r = _analytics(curve, cache, F.make_swap(F.README_VALUE_DT, "1W", 5.2014 / 100), F.README_VALUE_DT)
    assert r["value"] == 4.672529030358419e-11                     # cell 36, reproduced bit for bit
    assert r["delta"][1] == pytest.approx(1.9158970567491282, rel=1e-14)   # cell 40 ladder['1W']
    assert np.count_nonzero(r["delta"]) == 1
    assert f"{r['gamma'].sum():.6g}" == "-7.34132e-06"            # cell 44 repr
    assert np.count_nonzero(r["gamma"]) == 1 and r["gamma"][1, 1] < 0
    # closed forms for a single-period swap: N a/(1+ra) 1e-4 and -2N a^2/(1+ra)^2 1e-8
    a, rr = 7 / 365, 0.052014
    assert r["delta"][1] == pytest.approx(1e6 * a / (1 + rr * a) * 1e-4, rel=1e-14)
    assert r["gamma"][1, 1] == pytest.approx(-2e6 * a * a / (1 + rr * a) ** 2 * 1e-8, rel=1e-13)


def test_notebook_curve_table(readme):
    """Cell 12 prints the bootstrapped DFs to 4 decimals: DF(1Y) = 0.9520, DF(10Y) = 0.6723, ..."""
    _, curve, cache = readme
    t, d = cache["times"], cache["dfs"]
    first = lambda x: d[int(np.searchsorted(t, x))]                # first knot of a cluster = the pillar's own swap
    printed = {curve.swap_times[14]: 0.9520, curve.swap_times[16]: 0.9114, curve.swap_times[19]: 0.8134,
               curve.swap_times[24]: 0.6723, curve.swap_times[29]: 0.3041, curve.swap_times[31]: 0.1513}
    for tt, want in list(printed.items())[:5]:
        assert round(float(first(tt)), 4) == want
    assert cache["times"].shape == (264,) and len(np.unique(cache["times"])) == 66
    assert d[0] == 1.0 and np.all(cache["jac"][0] == 0.0) and np.all(cache["hess"][0] == 0.0)


@pytest.mark.parametrize("which", ["gbp_apr", "gbp_dec", "usd_dec", "gbp_ffr"])
def test_calibration_swaps_reprice_through_engine_grid(which):
    """tests/test_refit_curves.py:152-231: every calibration OIS values to |PV| <= 1e-5 (N = 1e6)."""
    if which == "gbp_apr":
        vd, model, name = F.README_VALUE_DT, F.readme_model(), "GBP_OIS_SONIA"
    elif which == "gbp_dec":
        vd, model, name = F.TEST_VALUE_DT, F.gbp_model(F.TEST_VALUE_DT), "GBP_OIS_SONIA"
    elif which == "gbp_ffr":
        vd, model, name = F.README_VALUE_DT, F.gbp_model(interp=InterpTypes.FLAT_FWD_RATES), "GBP_OIS_SONIA"
    else:
        vd, model, name = F.TEST_VALUE_DT, F.usd_model(), "USD_OIS_SOFR"
    curve = model.curves[name]
    cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs, derivatives=False)
    assert all(abs(a - b) < 1e-6 or k == 0.0 for k, a, b in cache["collisions"])
    worst = 0.0
    for s in curve._used_swaps:
        fx, fl = O.leg_inputs_from_swap(s, vd, times_from_dates)
        worst = max(worst, abs(O.ois_value(cache, curve._interp_type.value, fx, fl)))
    assert worst <= 1e-5, worst
    assert worst <= 5e-9          # in fact rounding noise only


def _scenario_value(model, swap, vd, shock):
    m2 = model.scenario("GBP_OIS_SONIA", shock)
    c2 = m2.curves.GBP_OIS_SONIA
    cache = O.cached_curve(c2.swap_rates, c2.swap_times, c2.year_fracs, derivatives=False)
    fx, fl = O.leg_inputs_from_swap(swap, vd, times_from_dates)
    return O.ois_value(cache, c2._interp_type.value, fx, fl)


def test_delta_matches_bump_and_reprice(readme):
    """tests/test_ois_request_types.py:429-524: AD total delta vs central differences through
    Model.scenario (rel 1e-4 at 1 bp) and per-tenor ladder entries (5 %)."""
    model, curve, cache = readme
    vd = F.README_VALUE_DT
    swap = F.make_swap(vd, "10Y", 0.045)
    r = _analytics(curve, cache, swap, vd, gamma=False)
    fd = (_scenario_value(model, swap, vd, 0.01) - _scenario_value(model, swap, vd, -0.01)) / 2.0
    assert abs(r["delta"].sum() - fd) / abs(fd) < 1e-4
    assert abs(r["delta"].sum() - fd) / abs(fd) < 1e-6      # central-difference truncation is ~2e-7
    for tenor in ("2Y", "5Y", "10Y"):
        i = F.TENORS.index(tenor)
        fd_i = (_scenario_value(model, swap, vd, {tenor: 0.01}) - _scenario_value(model, swap, vd, {tenor: -0.01})) / 2.0
        assert abs(r["delta"][i] - fd_i) <= 0.05 * max(abs(fd_i), 1e-9)
        assert abs(r["delta"][i] - fd_i) <= 1e-5 * max(abs(fd_i), 1.0)


def test_gamma_taylor_and_symmetry(readme):
    """tests/test_ois_request_types.py:577-641, 707-796: second-order Taylor expansion explains a
    +-100 bp parallel move to < 5 %, gamma is symmetric (rtol 1e-10) with non-zero off-diagonals."""
    model, curve, cache = readme
    vd = F.README_VALUE_DT
    swap = F.make_swap(vd, "10Y", 0.045)
    r = _analytics(curve, cache, swap, vd)
    g = r["gamma"]
    assert g.shape == (32, 32)
    assert np.allclose(g, g.T, rtol=1e-10, atol=1e-14)
    assert np.count_nonzero(g - np.diag(np.diag(g))) > 0
    for bp in (100.0, -100.0):
        actual = _scenario_value(model, swap, vd, bp * 0.01) - r["value"]
        first = r["delta"].sum() * bp
        second = first + 0.5 * g.sum() * bp * bp
        assert abs(actual - second) < 0.5 * abs(actual - first)
        assert abs(actual - second) / abs(actual) < 0.05


def test_pay_receive_antisymmetry_and_extremes(readme):
    """tests/test_ois_request_types.py:841-942: PAY + RECEIVE cancel to 1e-10; 3M and 50Y are finite."""
    _, curve, cache = readme
    vd = F.README_VALUE_DT
    a = _analytics(curve, cache, F.make_swap(vd, "5Y", 0.045, pay=True), vd)
    b = _analytics(curve, cache, F.make_swap(vd, "5Y", 0.045, pay=False), vd)
    assert abs(a["value"] + b["value"]) < 1e-10 and np.max(np.abs(a["delta"] + b["delta"])) < 1e-10
    assert np.max(np.abs(a["gamma"] + b["gamma"])) < 1e-10
    for tenor in ("3M", "50Y"):
        r = _analytics(curve, cache, F.make_swap(vd, tenor, 0.04), vd)
        assert np.isfinite(r["value"]) and np.all(np.isfinite(r["delta"])) and np.all(np.isfinite(r["gamma"]))
        assert abs(r["delta"].sum()) > 0


def test_discontinuity_at_pillars(readme):
    """SURVEY.md 'three facts' no. 3: duplicate knot times make D jump right after a pillar."""
    _, curve, cache = readme
    at = float(O.simple_interpolate(1.0, cache["times"], cache["dfs"], 4))
    after = float(O.simple_interpolate(1.0 + 1e-9, cache["times"], cache["dfs"], 4))
    assert round(at, 6) == 0.952024 and round(after, 6) == 0.962649


def test_oracle_reproduces_golden_vectors():
    """The committed vectors are what the oracle produces today (regenerate with make_golden.py)."""
    with open(GOLDEN) as f:
        golden = json.load(f)
    for case, want in zip(make_golden.CASES, golden["cases"]):
        vd, curve, swaps = make_golden.build(case)
        cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
        assert cache["times"].shape[0] == want["n_knots"]
        for s, row in zip(swaps, want["trades"]):
            r = _analytics(curve, cache, s, vd)
            assert r["value"] == pytest.approx(row["pv"], rel=1e-13, abs=1e-9)
            assert np.allclose(r["delta"], row["delta"], rtol=1e-12, atol=1e-12)
            assert np.allclose(r["gamma"], np.array(row["gamma"]), rtol=1e-12, atol=1e-16)


def test_notebook_non_ad_valuation_outputs_bit_for_bit():
    """notebooks/intro.ipynb cells 23-29: the 1W calibration swap through `OISCurve`'s own nodes and the legs'
    non-AD `value()` - par rate, PV, PV01 by `bump()` and PV under the 10Y-shocked
    scenario model, each equal to the printed repr.  (The notebook's OIS call predates the `floating_index` /
    `currency` arguments; they are the GBP ones here.)"""
    from adrates_amd.trades.rates.ois import OIS
    from adrates_amd.utils import (BusDayAdjustTypes, CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes,
                                   SwapTypes)
    vd = F.README_VALUE_DT
    model = F.readme_model()
    curve = model.curves.GBP_OIS_SONIA
    swap = OIS(effective_dt=vd.add_weekdays(0), term_dt_or_tenor="1W", fixed_leg_type=SwapTypes.PAY,
               fixed_coupon=5.2014 / 100, fixed_freq_type=FrequencyTypes.ANNUAL, fixed_dc_type=DayCountTypes.ACT_365F,
               floating_index=CurveTypes.GBP_OIS_SONIA, currency=CurrencyTypes.GBP,
               bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING, float_freq_type=FrequencyTypes.ANNUAL,
               float_dc_type=DayCountTypes.ACT_365F)
    assert swap.swap_rate(vd, curve) * 10000 == 5.201400000000243          # cell 23
    assert swap.value(vd, curve) == 4.672529030358419e-11                  # cell 25
    assert swap.pv01(vd, curve) == 1.9158970567491285                      # cell 27
    bumped = model.scenario("GBP_OIS_SONIA", {"10Y": 0.01})
    assert swap.value(vd, bumped.curves.GBP_OIS_SONIA) == 4.672529030358419e-11   # cell 29


@pytest.mark.parametrize("which", ["base", "bump_10Y_1bp"])
def test_notebook_curve_tables_all_rows(readme, which):
    """Cells 12 and 20 in full (tests/golden/notebook_curve_tables.json, extracted by make_notebook_table.py): the
    32-row table of the README curve and of its 10Y + 1 bp scenario.  Tenor, last accrual fraction and rate
    columns are the curve's inputs to the engine and must match to the printed digits.  The DF
    column of that notebook is one row late (row i shows pillar i - 1's value, row 0 the 1.0 of t = 0) and was
    produced when the curve's own bootstrap still used each swap's own rate for its intermediate coupons - which
    is what the ENGINE grid does today (row B): every printed value equals the engine-grid DF of its pillar, and
    the curve's own nodes agree up to the 10Y pillar, before the first tenor gap."""
    from adrates_amd.market.curves.curve_tables import build_engine_curve
    table = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "notebook_curve_tables.json")))[which]
    rows = np.array(table["rows"])
    assert rows.shape == (32, 4)
    model = readme[0] if which == "base" else readme[0].scenario("GBP_OIS_SONIA", {"10Y": 0.01})
    curve = model.curves.GBP_OIS_SONIA
    for i in range(32):
        assert rows[i, 0] == round(curve.swap_times[i], 4)
        assert rows[i, 1] == round(curve.year_fracs[i][-1], 4)
        assert rows[i, 2] == round(curve.swap_rates[i], 4)
    assert rows[0, 3] == 1.0
    for name, (t, d) in (("oracle", (lambda c: (c["times"], c["dfs"]))(O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs))),
                         ("product", (lambda h: (h.times, h.dfs))(build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)))):
        for i in range(1, 32):
            knot = int(np.searchsorted(t, curve.swap_times[i - 1]))        # first knot of the cluster: the pillar's own swap
            assert abs(t[knot] - curve.swap_times[i - 1]) < 1e-9
            assert round(float(d[knot]), 4) == pytest.approx(rows[i, 3], abs=1e-12), (name, i)
    for i in range(1, 26):
        assert round(float(curve._repr_dfs[i]), 4) == pytest.approx(rows[i, 3], abs=1e-12)
    if which == "bump_10Y_1bp":
        base = np.array(json.load(open(os.path.join(os.path.dirname(__file__), "golden", "notebook_curve_tables.json")))["base"]["rows"])
        assert np.flatnonzero(np.abs(base - rows).max(axis=1) > 0).tolist() == [24, 25]    # the 10Y rate row, its DF one row late
