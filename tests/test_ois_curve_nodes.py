"""OISCurve's own node set and the DiscountCurve queries: product vs the
oracle restatement node by node, and the property assertions of the reference's
tests/test_curve_bootstrap_validation.py (classes :60-262) re-run on this implementation."""
import numpy as np
import pytest

from adrates_amd.market.curves.discount_curve import DiscountCurve
from adrates_amd.utils import DayCountTypes, FrequencyTypes, InterpTypes
from adrates_amd.utils.error import LibError
from oracle import curve_nodes as ON

from . import _fixtures as F


@pytest.fixture(scope="module")
def gbp_curve():
    return F.gbp_model().curves.GBP_OIS_SONIA


@pytest.mark.parametrize("model_fn,name", [(F.gbp_model, "GBP_OIS_SONIA"), (F.usd_model, "USD_OIS_SOFR")])
def test_nodes_equal_recursive_restatement(model_fn, name):
    curve = getattr(model_fn().curves, name)
    times, dfs, repr_dfs = ON.build_nodes(list(curve.swap_rates), list(curve.swap_times), curve.year_fracs)
    assert np.array_equal(curve._times, times)
    assert np.allclose(curve._dfs, dfs, rtol=1e-14, atol=0)          # math.exp/log vs numpy: ulps, compounded
    assert np.allclose(curve._repr_dfs, repr_dfs, rtol=1e-14, atol=0)
    assert len(curve._repr_dfs) == len(curve.swap_rates) + 1
    # node set = t0 + pillars + the coupon dates no shorter swap ends on
    assert len(times) > len(curve.swap_rates) + 1 and np.all(np.diff(times) > 0)


def test_df_ad_is_linear_forward_interpolation(gbp_curve):
    c = gbp_curve
    for t in (0.0, 0.01, 0.5, 1.0, 5.0, 10.0, 17.3, 49.9, 60.0):
        assert c.df_ad(t) == pytest.approx(ON.linear_forward_df(t, c._times, c._dfs), rel=1e-15)
    ts = np.array([0.25, 3.0, 12.5])
    assert np.allclose(c.df_ad(ts), [c.df_ad(float(t)) for t in ts], rtol=1e-15)
    # README section 1: `curve.df_ad(5.0)` is a discount factor for a TIME
    assert 0.7 < c.df_ad(5.0) < 0.9 and c.df_ad(10.0) < c.df_ad(5.0)


@pytest.mark.parametrize("interp", [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES,
                                    InterpTypes.LINEAR_FWD_RATES])
def test_df_follows_the_node_interpolation_schemes(interp):
    c = F.gbp_model(interp=interp).curves.GBP_OIS_SONIA
    for t in (0.0, 0.001, 0.0027, 0.3, 1.0, 2.5, 9.99, 30.0, 50.01, 75.0):
        want = ON.uinterpolate(t, list(c._times), list(c._dfs), interp.value)
        assert c._df(t) == pytest.approx(want, rel=1e-15)
    # nodes are reproduced (LINEAR_FWD_RATES' first segment carries the reference's 1e-10 fudge terms)
    for k in (1, 5, 20, len(c._times) - 1):
        assert c._df(float(c._times[k])) == pytest.approx(c._dfs[k], rel=1e-9 if interp == InterpTypes.LINEAR_FWD_RATES else 1e-15)
    # df() converts dates with ACT/ACT ISDA unless told otherwise (discount_curve.py:300-305)
    d = F.README_VALUE_DT.add_tenor("7Y")
    from adrates_amd.utils.helpers import times_from_dates
    assert c.df(d) == c._df(times_from_dates(d, c._value_dt, DayCountTypes.ACT_ACT_ISDA))
    assert c.df(d, DayCountTypes.ACT_365F) == c._df(times_from_dates(d, c._value_dt, DayCountTypes.ACT_365F))
    with pytest.raises(LibError):
        c._df(-0.5)


def test_calibration_swaps_reprice_off_the_nodes(gbp_curve):
    """`_check_refits` (ois_curve.py:344-358) and OIS.value / pv01 / swap_rate (ois.py:209-320)."""
    gbp_curve._check_refits(1e-5)
    for swap in gbp_curve._used_swaps[::5]:
        v = swap.value(swap._effective_dt, gbp_curve)
        assert abs(v) / swap._notional < 1e-5
        par = swap.swap_rate(swap._effective_dt, gbp_curve)
        # the reference's pv01 carries a factor 100 (ois.py:281), so its swap_rate is the par rate / 100
        assert par * 100 == pytest.approx(swap._fixed_coupon, abs=2e-6)
        assert swap.pv01(swap._effective_dt, gbp_curve) > 0


# ---- the reference's validation classes (tests/test_curve_bootstrap_validation.py) ------------------------
def test_dfs_strictly_decreasing_and_in_range(gbp_curve):
    dfs = gbp_curve._dfs
    assert np.all(np.diff(dfs) < 0)                                  # :66-83
    assert np.all((dfs > 0) & (dfs <= 1.0)) and 0.99 < dfs[1] <= 1.0  # :103-114


def test_interpolated_dfs_monotone_and_smooth(gbp_curve):
    vd = gbp_curve._value_dt
    dates = [vd.add_months(m) for m in range(1, 361, 3)]
    dfs = np.array([gbp_curve.df(d) for d in dates])
    assert np.all(np.diff(dfs) <= 1e-10)                             # :85-98
    assert np.max(np.abs(np.diff(dfs)) / dfs[:-1]) < 0.05            # :202-214 (quarterly steps here)


def test_forward_and_zero_rates_reasonable(gbp_curve):
    vd = gbp_curve._value_dt
    dates = [vd.add_tenor(f"{y}Y") for y in (1, 2, 3, 5, 7, 10, 15, 20, 30)]
    fwds = np.array([gbp_curve.fwd(d) for d in dates])
    zeros = gbp_curve.zero_rate(dates, FrequencyTypes.CONTINUOUS, DayCountTypes.ACT_365F)
    assert np.all((fwds > -0.05) & (fwds < 0.20))                    # :120-131
    assert np.mean(fwds > 0) > 0.8                                   # :134-151
    assert np.all((zeros > -0.05) & (zeros < 0.20))                  # :158-167
    assert np.max(np.abs(np.diff(zeros))) < 0.02                     # :170-184
    assert gbp_curve.zero_rate(dates[3]) == pytest.approx(
        gbp_curve.zero_rate(dates, FrequencyTypes.CONTINUOUS, DayCountTypes.ACT_360)[3])


def test_extrapolation_beyond_last_pillar(gbp_curve):
    vd = gbp_curve._value_dt
    df_last = gbp_curve._dfs[-1]
    df_60 = gbp_curve.df(vd.add_tenor("60Y"))
    df_100 = gbp_curve.df(vd.add_tenor("100Y"))
    assert 0.0 < df_60 < df_last and 0.0 < df_100 < 0.5              # :239-262


def test_discount_curve_from_year_offsets_and_bump():
    vd = F.README_VALUE_DT
    dc = DiscountCurve(vd, [1.0, 2.0, 5.0], np.array([0.96, 0.92, 0.80]), InterpTypes.FLAT_FWD_RATES)
    assert dc._times[0] == 0.0 and dc._dfs[0] == 1.0 and len(dc._times) == 4
    up = dc.bump(0.001)
    assert np.allclose(up._dfs, dc._dfs * np.exp(-0.001 * dc._times))
    assert dc.swap_rate(vd, vd.add_tenor("5Y"))[0] == pytest.approx(0.045, abs=0.01)
    assert dc.fwd_rate(vd.add_tenor("1Y"), "1Y") > 0
    with pytest.raises(LibError):
        DiscountCurve(vd, [2.0, 1.0], np.array([0.9, 0.95]))
