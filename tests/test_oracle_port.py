"""oracle/port.c (batched C restatement, knot-DF space) against the torch.func oracle."""
import numpy as np
import pytest

from adrates_amd.trades import synthetic
from adrates_amd.trades.compiler import compile_ois
from adrates_amd.utils import DayCountTypes, FrequencyTypes, InterpTypes
from oracle import cavour_oracle as O
from oracle import port

from . import _fixtures as F
from ._parity import oracle_price, trade_errors


@pytest.mark.parametrize("interp", [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_FWD_RATES])
def test_port_matches_ad_oracle(interp):
    vd = F.README_VALUE_DT
    curve = F.gbp_model(vd, interp).curves.GBP_OIS_SONIA
    cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    swaps = [F.make_swap(vd, "10Y", 0.045, 1e7), F.make_swap(vd, "55Y", 0.039, 5e6, pay=False),
             F.make_swap(vd, "5Y", 0.04, 1e6, fixed_freq=FrequencyTypes.SEMI_ANNUAL),
             F.make_swap(vd, "3Y", 0.04, 1e6, payment_lag=2, spread=0.002),
             F.make_swap(vd, "6Y", 0.04, 1e6, float_dc=DayCountTypes.THIRTY_E_360, pay=False),
             F.make_swap(vd, "1W", 0.052014), F.make_swap(vd, "1D", 0.05)]
    terms = synthetic.draw_terms(12, "offgrid", seed=5)
    swaps += synthetic.swaps_from_terms(vd, *terms)
    refs = oracle_price(curve, swaps, vd, cache)
    got = port.price(interp.value, cache["times"], cache["dfs"], cache["jac"], cache["hess"], compile_ois(swaps, vd))
    for i, (r, s) in enumerate(zip(refs, swaps)):
        e = trade_errors(got["pv"][i], got["delta"][i], got["gamma"][i], r["value"], r["delta"], r["gamma"],
                         s._notional)
        assert e < 1e-12, (i, e)


def test_port_threads_agree():
    vd = F.README_VALUE_DT
    curve = F.readme_model().curves.GBP_OIS_SONIA
    cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    b = synthetic.synthesize(vd, 3000, seed=3)
    one = port.price(4, cache["times"], cache["dfs"], cache["jac"], cache["hess"], b, n_threads=1)
    many = port.price(4, cache["times"], cache["dfs"], cache["jac"], cache["hess"], b, n_threads=4)
    for k in ("pv", "delta", "gamma"):
        assert np.array_equal(one[k], many[k])
