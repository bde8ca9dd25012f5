"""A third evaluation that shares no differentiation with anything else (oracle/mp_oracle.py): the PV restated in
60-digit arithmetic, delta and gamma by central differences w.r.t. the par rates.  `cavour_oracle` (torch.func
autodiff through the scan and the legs) must agree to 1e-12 on off-grid, multi-coupon trades - the cases the
reference-held numbers (a single-period 1W swap) do not reach: interpolation between duplicate knots, multi-pillar
gamma off-diagonals, payment-lag ratio terms, all three interpolation schemes, extrapolation, seasoned trades."""
import numpy as np
import pytest

from adrates_amd.utils import DayCountTypes, FrequencyTypes, InterpTypes
from adrates_amd.utils.helpers import times_from_dates
from oracle import cavour_oracle as O
from oracle import mp_oracle as MP

from . import _fixtures as F

TOL = 1e-12

CASES = {
    # name: (interp, swap kwargs)
    "offgrid_87M_lzr": (InterpTypes.LINEAR_ZERO_RATES, dict(tenor="87M", coupon=0.04, notional=1e7)),
    "readme_10Y_ffr": (InterpTypes.FLAT_FWD_RATES, dict(tenor="10Y", coupon=0.045, notional=1e7, pay=False)),
    "lagged_semi_float_41M_lzr": (InterpTypes.LINEAR_ZERO_RATES,
                                  dict(tenor="41M", coupon=0.043, notional=2.5e6, float_freq=FrequencyTypes.SEMI_ANNUAL,
                                       payment_lag=2, spread=0.0015)),
    "linear_fwd_29M": (InterpTypes.LINEAR_FWD_RATES, dict(tenor="29M", coupon=0.05, notional=1e6, payment_lag=3)),
    "beyond_last_knot_55Y_lzr": (InterpTypes.LINEAR_ZERO_RATES, dict(tenor="55Y", coupon=0.039, notional=1e6)),
    "seasoned_7Y_ffr": (InterpTypes.FLAT_FWD_RATES, dict(tenor="7Y", coupon=0.041, notional=5e6, back_months=7)),
}


def _build(case):
    interp, kw = CASES[case]
    kw = dict(kw)
    vd = F.README_VALUE_DT
    model = F.gbp_model(vd, interp)
    curve = model.curves.GBP_OIS_SONIA
    eff = vd.add_months(-kw.pop("back_months", 0))
    swap = F.make_swap(eff, kw.pop("tenor"), kw.pop("coupon"), **kw)
    fx, fl = O.leg_inputs_from_swap(swap, vd, times_from_dates)
    cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    ad = O.ois_analytics(cache, interp.value, fx, fl)
    third = MP.MpTrade(curve.swap_rates, curve.swap_times, curve.year_fracs, interp.value, fx, fl)
    return swap, ad, third


def _err(got, want, n, floor):
    """The parity metric of tests/_parity.py: per-unit-notional error and the ladder-relative error."""
    got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
    unit = np.max(np.abs(got - want) / n / np.maximum(1.0, np.abs(want) / n))
    ladder = np.max(np.abs(got - want)) / max(np.max(np.abs(want)), floor * n)
    return max(unit, ladder)


@pytest.mark.parametrize("case", list(CASES))
def test_autodiff_oracle_agrees_with_high_precision_differences(case):
    swap, ad, third = _build(case)
    n = abs(swap._notional)
    assert _err(third.value(), ad["value"], n, 1e-4) <= TOL
    delta = third.delta()
    assert _err(delta, ad["delta"], n, 1e-8) <= TOL
    # gamma: every pair among the four largest-delta pillars (off-diagonals included), the largest entries of the
    # autodiff matrix wherever they are, and a few structurally zero pairs
    live = [int(p) for p in np.argsort(-np.abs(delta))[:4]]
    pairs = {(min(p, q), max(p, q)) for p in live for q in live}
    g = ad["gamma"]
    for flat in np.argsort(-np.abs(np.triu(g)).ravel())[:6]:
        pairs.add((int(flat // g.shape[1]), int(flat % g.shape[1])))
    dead = [int(p) for p in np.flatnonzero(delta == 0.0)[:2]]
    pairs |= {(min(p, live[0]), max(p, live[0])) for p in dead}
    got = third.gamma(sorted(pairs))
    want = np.array([g[p, q] for p, q in sorted(pairs)])
    have = np.array([got[pq] for pq in sorted(pairs)])
    scale = max(np.max(np.abs(g)), 1e-12 * n)
    assert np.max(np.abs(have - want)) / scale <= TOL, (case, np.max(np.abs(have - want)) / scale)
    assert np.any(want != 0.0) and len([1 for p, q in pairs if p != q and g[p, q] != 0.0]) >= 3
    for p in dead:
        assert got[(min(p, live[0]), max(p, live[0]))] == 0.0 and np.all(g[p] == 0.0)


def test_cross_currency_autodiff_oracle_agrees_with_high_precision_differences():
    """oracle/xccy_oracle.py - the XCCY curve's scan and `Engine._compute_xccy` differentiated with torch.func - against
    the same PV restated in 60-digit arithmetic as a function of the three quote vectors and differenced
    (oracle/mp_oracle.py::MpXccy): PV, entries of the three delta ladders, of the three gamma matrices and of the
    foreign-rate x basis cross term, on a seasoned semi-annual swap with payment lags."""
    from adrates_amd.market.curves.curve_tables import build_engine_curve
    from adrates_amd.utils import CurveTypes
    from oracle import xccy_oracle as XO
    from tests.test_gpu_xccy import VALUE_DT, _model, _swap
    from tests.test_xccy_curve import _basis_swaps
    m = _model()
    gbp, usd, x = m.curves.GBP_OIS_SONIA, m.curves.USD_OIS_SOFR, m.curves.USD_GBP_BASIS
    swap = _swap("6Y", 0.0041, lag=2, effective=VALUE_DT.add_months(-8), freq=FrequencyTypes.SEMI_ANNUAL, notional=5_000_000)
    cache = lambda c: (lambda h: dict(times=h.times, dfs=h.dfs, jac=h.jac, hess=h.hess))(build_engine_curve(c.swap_rates, c.swap_times, c.year_fracs))
    ad = XO.xccy_analytics(swap, VALUE_DT, cache(gbp), gbp._interp_type.value, cache(usd), usd._interp_type.value, x, times_from_dates)
    third = MP.MpXccy(swap, VALUE_DT, gbp, usd, x, _basis_swaps(), times_from_dates)
    n = abs(swap._domestic_leg._notional)
    # the XCCY knot DFs of the 60-digit scan are the curve's
    assert np.allclose([float(d) for d in third.xccy_dfs(third.spreads)], np.asarray(x._dfs), rtol=1e-13, atol=0)
    assert abs(third.value() - ad["value"]) <= TOL * max(abs(ad["value"]), 1e-4 * n)
    for which, key, pillars in (("dom", "delta_dom", (1, 4, 6)), ("for", "delta_for", (0, 3, 5, 6)), ("basis", "delta_basis", (2, 5, 6))):
        got = third.delta(which, pillars)
        scale = max(np.max(np.abs(ad[key])), 1e-8 * n)
        for p in pillars:
            assert abs(got[p] - ad[key][p]) <= TOL * scale, (key, p, got[p], ad[key][p])
    for (a, p, b, q), key in ((("for", 5, "for", 5), "gamma_for"), (("for", 3, "for", 5), "gamma_for"),
                               (("basis", 5, "basis", 5), "gamma_basis"), (("basis", 2, "basis", 6), "gamma_basis"),
                               (("dom", 4, "dom", 6), "gamma_dom"), (("for", 5, "basis", 5), "cross_for_basis"),
                               (("for", 3, "basis", 6), "cross_for_basis")):
        want = ad[key][p, q]
        scale = max(np.max(np.abs(ad[key])), 1e-12 * n)
        assert abs(third.second(a, p, b, q) - want) <= 1e-11 * scale, (key, p, q, third.second(a, p, b, q), want)
