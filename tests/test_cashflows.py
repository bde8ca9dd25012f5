"""CASHFLOWS request (cavour/market/position/engine.py:34-87, 191-213; results.py:946-1121): payment tables of
both legs valued off the curve's own node set.  Host-only: no GPU is touched when CASHFLOWS is the only request."""
import numpy as np
import pytest

from adrates_amd.requests.results import CashflowItem, Cashflows
from adrates_amd.utils import FrequencyTypes, RequestTypes

from . import _fixtures as F


@pytest.fixture(scope="module")
def model():
    return F.gbp_model(F.README_VALUE_DT)


def test_cashflow_tables_of_a_payer_swap(model):
    vd = F.README_VALUE_DT
    swap = F.make_swap(vd, "5Y", 0.045, 1e7, float_freq=FrequencyTypes.SEMI_ANNUAL)
    res = swap.position(model).compute([RequestTypes.CASHFLOWS])
    assert res.value is None and res.risk is None and res.gamma is None
    cfs = res.cashflows
    assert isinstance(cfs, Cashflows) and cfs.validate() and len(cfs) == 5 + 10
    assert len(cfs.fixed()) == 5 and len(cfs.floating()) == 10
    assert {cf.leg_type for cf in cfs.cashflows} == {"Fixed_Pay", "Float_Rec"}
    assert len(cfs.pay()) == 5 and len(cfs.receive()) == 10 and len(cfs.notional_exchange()) == 0
    curve = model.curves.GBP_OIS_SONIA
    for cf, pay_dt, alpha in zip(cfs.fixed().cashflows, swap._fixed_leg._payment_dts, swap._fixed_leg._year_fracs):
        assert cf.payment_date == pay_dt and cf.accrual_period == alpha and cf.notional == 1e7
        assert cf.amount == pytest.approx(-0.045 * alpha * 1e7, rel=1e-15)          # pay leg: negative
        assert cf.payment_fraction == pytest.approx(0.045 * alpha, rel=1e-15)
        assert cf.discount_factor == pytest.approx(curve.df(pay_dt, swap._fixed_leg._dc_type), rel=1e-15)
        assert cf.discounted_amount == pytest.approx(cf.amount * cf.discount_factor, rel=1e-15)
    for cf in cfs.floating().cashflows:
        assert cf.amount > 0 and 0 < cf.discount_factor < 1
    # totals: the swap's non-AD value off the same nodes
    assert cfs.total_pv == pytest.approx(swap.value(vd, curve), rel=1e-14)
    assert cfs.sum().amount == cfs.total_pv and cfs.sum().currency == swap._currency
    assert cfs.total_amount == pytest.approx(sum(cf.amount for cf in cfs.cashflows))
    d = cfs.to_dict()
    assert d["count"] == 15 and d["currency"] == "GBP" and d["cashflows"][0]["leg_type"] == "Fixed_Pay"
    assert cfs.df.shape == (15, 7) and "Cashflows(count=15" in repr(cfs)


def test_receiver_swap_flips_leg_labels_and_signs(model):
    vd = F.README_VALUE_DT
    swap = F.make_swap(vd, "3Y", 0.04, 2e6, pay=False)
    cfs = swap.position(model).compute([RequestTypes.CASHFLOWS]).cashflows
    assert {cf.leg_type for cf in cfs.cashflows} == {"Fixed_Rec", "Float_Pay"}
    assert all(cf.amount > 0 for cf in cfs.fixed().cashflows)
    assert all(cf.amount < 0 for cf in cfs.floating().cashflows)
    payer = F.make_swap(vd, "3Y", 0.04, 2e6, pay=True)
    other = payer.position(model).compute([RequestTypes.CASHFLOWS]).cashflows
    assert cfs.total_pv == pytest.approx(-other.total_pv, rel=1e-14)


def test_containers_validate():
    with pytest.raises(ValueError):
        Cashflows("not a list", None).validate()
    with pytest.raises(ValueError):
        Cashflows([1, 2], None).validate()
    item = CashflowItem("30-APR-2025", 1e6, 0.05, 1.0, 5e4, 0.95, 4.75e4, "Fixed_Rec")
    assert item.to_dict()["discounted_amount"] == 4.75e4
    assert len(Cashflows([], None)) == 0 and Cashflows([], None).df.empty
