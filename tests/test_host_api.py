"""Drop-in surface: Model / OIS / results objects behave like the reference's."""
import numpy as np
import pytest

from adrates_amd.market.portfolio.portfolio import Portfolio
from adrates_amd.models.models import Model
from adrates_amd.requests.results import AnalyticsResult, Delta, Gamma, Risk, Valuation
from adrates_amd.trades.compiler import compile_ois
from adrates_amd.trades.rates.ois import OIS
from adrates_amd.utils import (CurrencyTypes, CurveTypes, Date, DayCountTypes, FrequencyTypes, InstrumentTypes,
                               InterpTypes, LibError, SwapTypes)

from . import _fixtures as F


def test_model_build_curve_and_accessors():
    m = F.readme_model()
    c = m.curves.GBP_OIS_SONIA
    assert m.curves["GBP_OIS_SONIA"] is c
    with pytest.raises(AttributeError):
        m.curves.USD_OIS_SOFR
    with pytest.raises(KeyError):
        m.curves["USD_OIS_SOFR"]
    with pytest.raises(KeyError):
        Model(F.README_VALUE_DT).build_curve("NOT_A_CURVE", [5.0], ["1Y"])
    assert c._interp_type == InterpTypes.LINEAR_ZERO_RATES and len(c.swap_rates) == 32
    assert c.swap_rates[0] == 5.1998 / 100      # quotes are in percent


def test_scenario_shifts_quotes_in_percent_units():
    m = F.readme_model()
    up = m.scenario("GBP_OIS_SONIA", 0.01)
    assert up.curves.GBP_OIS_SONIA.swap_rates[5] == (F.GBP_PX[5] + 0.01) / 100
    t = m.scenario("GBP_OIS_SONIA", {"10Y": 1.0}, new_name="GBP_OIS_SONIA")
    r0, r1 = m.curves.GBP_OIS_SONIA.swap_rates, t.curves.GBP_OIS_SONIA.swap_rates
    assert r1[24] == (F.GBP_PX[24] + 1.0) / 100 and r1[:24] == r0[:24] and r1[25:] == r0[25:]
    assert t.curves.GBP_OIS_SONIA.swap_times == m.curves.GBP_OIS_SONIA.swap_times
    with pytest.raises(ValueError):
        m.scenario("USD_OIS_SOFR", 0.01)


def test_ois_legs_and_signs():
    vd = F.README_VALUE_DT
    s = F.make_swap(vd, "10Y", 0.045, 1e7)
    assert s.derivative_type == InstrumentTypes.OIS_SWAP
    assert s._fixed_leg._leg_type == SwapTypes.PAY and s._float_leg._leg_type == SwapTypes.RECEIVE
    assert len(s._fixed_leg._payments) == 10 and s._float_leg._principal == 0.0
    assert s._fixed_leg._payments[0] == s._fixed_leg._year_fracs[0] * 1e7 * 0.045
    b = compile_ois([s], vd)
    assert b.fix_sign[0] == -1.0 and b.flt_sign[0] == 1.0 and b.n_trades == 1
    assert np.array_equal(b.flt_te, b.flt_tp) and np.array_equal(b.flt_ts[1:], b.flt_tp[:-1]) and b.flt_ts[0] == 0.0
    # default float day count is 30E/360 (ois.py:113): times on the two legs then differ
    d = OIS(vd, "2Y", SwapTypes.RECEIVE, 0.04, FrequencyTypes.ANNUAL, DayCountTypes.ACT_365F,
            CurveTypes.GBP_OIS_SONIA, CurrencyTypes.GBP)
    bd = compile_ois([d], vd)
    assert bd.flt_tp[-1] == 2.0 and bd.fix_tp[-1] == 730 / 365 and bd.fix_sign[0] == 1.0
    with pytest.raises(LibError):
        OIS(vd, "2Y", "PAY", 0.04, FrequencyTypes.ANNUAL, DayCountTypes.ACT_365F,
            CurveTypes.GBP_OIS_SONIA, CurrencyTypes.GBP)                        # Argument Type Error
    with pytest.raises(LibError):
        F.make_swap(vd, Date(1, 1, 2020), 0.04)                                  # matures before it starts


def test_result_objects():
    ten = ["1W", "1W", "1Y"]
    d1 = Delta([1.0, 2.0, 3.0], ten, CurrencyTypes.GBP, CurveTypes.GBP_OIS_SONIA)
    d2 = Delta(np.array([0.5, 0.5, 0.5]), ten, CurrencyTypes.GBP, CurveTypes.GBP_OIS_SONIA)
    s = d1 + d2
    assert s.value.amount == 7.5 and s.value.currency == CurrencyTypes.GBP
    assert list(s.risk_ladder) == [1.5, 2.5, 3.5]
    assert s.ladder.data == {"1W": 2.5, "1Y": 3.5}          # colliding labels collapse, as in the reference
    assert "points=3" in repr(s)
    with pytest.raises(ValueError):
        d1 + Delta([1.0, 2.0, 3.0], ten, CurrencyTypes.USD, CurveTypes.GBP_OIS_SONIA)
    with pytest.raises(ValueError):
        Delta([1.0, 2.0], ten, CurrencyTypes.GBP, CurveTypes.GBP_OIS_SONIA)
    g = Gamma(np.eye(3), ten, CurrencyTypes.GBP, CurveTypes.GBP_OIS_SONIA)
    assert (g + g).value.amount == 6.0 and (g + g).risk_ladder.shape == (3, 3)
    assert g.to_dict["1Y"]["1Y"] == 1.0
    v = Valuation(10.0, CurrencyTypes.GBP)
    assert (v + v).amount == 20.0 and (v * 2).amount == 20.0 and (v / 4).amount == 2.5 and (v - v).amount == 0.0
    assert sum([v, v]).amount == 20.0 and repr(v) == "10.00 GBP"
    with pytest.raises(ValueError):
        v + Valuation(1.0, CurrencyTypes.USD)
    with pytest.raises(TypeError):
        Valuation(1.0, "GBP")
    r = AnalyticsResult(value=v, risk=d1, gamma=g)
    assert r.value is v and r.risk is d1 and r.gamma is g and r.cashflows is None
    risk = Risk([d1])
    assert risk.GBP_OIS_SONIA is d1 and risk(CurveTypes.GBP_OIS_SONIA) is d1
    with pytest.raises(ValueError):
        risk(CurveTypes.USD_OIS_SOFR)


def test_unsupported_instrument_raises_liberror():
    class Bond:
        derivative_type = InstrumentTypes.BOND
    from adrates_amd.market.position.engine import Engine
    with pytest.raises(LibError):
        Engine(F.readme_model()).compute(Bond(), [])
    assert Portfolio([]).compute([]).value is None
