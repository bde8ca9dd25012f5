"""HIP path vs the torch.func oracle on trades built through the public API (small cases)."""
import numpy as np
import pytest

from adrates_amd.utils import DayCountTypes, FrequencyTypes, InterpTypes

from . import _fixtures as F
from ._parity import assert_parity, gpu_price, oracle_price

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("interp", [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES])
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
