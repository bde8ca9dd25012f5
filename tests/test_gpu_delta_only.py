"""Requests without GAMMA (BASELINE.json configs[1]: PV + 32-pillar delta) run on the lite kernel
(adrates_amd/csrc/kernels_lite.hip): parity against the C oracle on portfolios that exercise all of its row
layouts (1, 2 and 3 rows of 15 coupons per trade), both log-linear schemes, curves with and without the packed
layout, odd pillar counts, and the routing of what it does not take (payment lag, long legs, LINEAR_FWD_RATES)."""
import numpy as np
import pytest

from adrates_amd import _native
from adrates_amd.trades import synthetic
from adrates_amd.trades.compiler import OISTerms, compile_ois, compile_ois_terms
from adrates_amd.utils import (BusDayAdjustTypes, CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes, InterpTypes)
from oracle import port

from . import _fixtures as F
from ._parity import assert_batch_parity
from .test_gpu_parity_batch import _device_curve

pytestmark = pytest.mark.gpu


def _check(ctx, dc, host, method, batch, label):
    dt = _native.DeviceTrades(ctx, batch)
    ref = port.price(method, host.times, host.dfs, host.jac, host.hess, batch, want_gamma=False)
    got = _native.price(ctx, dc, dt, want_gamma=False, aggregate=True)
    worst = assert_batch_parity(got, ref, batch.notional)
    assert np.allclose(got["agg_pv"], ref["pv"].sum(), rtol=1e-10, atol=1e-3)
    assert np.allclose(got["agg_delta"], ref["delta"].sum(0), rtol=1e-10, atol=1e-6)
    assert np.all(got["agg_gamma"] == 0.0)                   # nothing requested, nothing accumulated
    pv_only = _native.price(ctx, dc, dt, want_delta=False, want_gamma=False, aggregate=True)
    assert_batch_parity(pv_only, dict(pv=ref["pv"]), batch.notional)
    assert np.allclose(pv_only["agg_pv"], ref["pv"].sum(), rtol=1e-10, atol=1e-3) and np.all(pv_only["agg_delta"] == 0.0)
    dt.close()
    print(f"{label}: worst error {worst:.2e}")


@pytest.mark.parametrize("interp", [InterpTypes.LINEAR_ZERO_RATES, InterpTypes.FLAT_FWD_RATES, InterpTypes.LINEAR_FWD_RATES])
@pytest.mark.parametrize("kind", ["offgrid", "ongrid"])
def test_benchmark_portfolio_value_and_delta(gpu_ctx, interp, kind):
    """The benchmark portfolio: 1-30 annual coupons, i.e. one- and two-row trades; odd trade counts leave groups of
    the last wavefront empty.  LINEAR_FWD_RATES is not the lite kernel's (the fast kernels built for it, kernels_fast_lindf.hip): same checks."""
    vd = F.README_VALUE_DT
    curve = F.gbp_model(vd, interp).curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    for n in (20001, 3):
        _check(gpu_ctx, dc, host, interp.value, synthetic.synthesize(vd, n, kind=kind, seed=31), f"{interp.name}/{kind}/{n}")


def test_mixed_frequencies_lags_spreads_long_legs(gpu_ctx):
    """Quarterly / semi-annual floats against annual or quarterly fixed legs (unmerged fixed coupons, 3-row trades,
    33-128-coupon chains, 200-coupon legs), spreads, payment lags (general kernel), seasoned and forward starts."""
    vd = F.README_VALUE_DT
    curve = F.readme_model().curves.GBP_OIS_SONIA
    host, dc = _device_curve(gpu_ctx, curve)
    rng = np.random.default_rng(77)
    n = 12000
    starts = [vd, vd.add_months(-7), vd.add_years(-2), vd.add_months(5)]
    eff = [starts[i] for i in rng.choice(4, size=n, p=[0.55, 0.15, 0.1, 0.2])]
    lfreq = [[FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL, FrequencyTypes.QUARTERLY, FrequencyTypes.MONTHLY][i]
             for i in rng.choice(4, size=n, p=[0.4, 0.3, 0.25, 0.05])]
    ffreq = [[FrequencyTypes.ANNUAL, FrequencyTypes.QUARTERLY][i] for i in rng.choice(2, size=n, p=[0.7, 0.3])]
    terms = OISTerms(effective_dt=eff, tenor=[f"{int(m)}M" for m in rng.integers(1, 361, n)],
                     coupon=rng.uniform(0.0, 0.08, n), notional=np.round(rng.uniform(1e5, 9e7, n), -4),
                     pay_fixed=rng.random(n) < 0.5, fixed_freq_type=ffreq, fixed_dc_type=DayCountTypes.ACT_365F,
                     floating_index=CurveTypes.GBP_OIS_SONIA, currency=CurrencyTypes.GBP, float_freq_type=lfreq,
                     float_dc_type=[[DayCountTypes.ACT_365F, DayCountTypes.ACT_360][i] for i in rng.integers(0, 2, n)],
                     float_spread=np.where(rng.random(n) < 0.3, rng.uniform(-0.002, 0.004, n), 0.0),
                     payment_lag=rng.choice([0, 0, 0, 2], size=n), bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
    batch = compile_ois_terms(terms, vd)
    m = np.maximum(np.diff(batch.flt_off), np.diff(batch.fix_off))
    lag0 = np.asarray(terms.payment_lag) == 0
    for lo, hi in ((1, 15), (16, 30), (31, 32), (33, 128), (129, 400)):
        assert ((m >= lo) & (m <= hi) & lag0).any(), (lo, hi)
    _check(gpu_ctx, dc, host, 4, batch, "mixed")


def test_curves_without_packed_layout_and_odd_pillar_counts(gpu_ctx):
    """The lite kernel needs no packed layout: a 5-pillar toy curve and a 17-pillar curve (odd: per-element delta
    stores) price through it; trades beyond the last knot and on the value date included."""
    vd = F.README_VALUE_DT
    for px, tenors in (([5.19, 5.13, 5.04, 4.75, 4.24], ["1M", "3M", "6M", "1Y", "5Y"]),
                       (list(F.GBP_PX[8:9] + F.GBP_PX[14:30]), list(F.TENORS[8:9] + F.TENORS[14:30]))):     # 6M, 1Y ... 30Y
        curve = F.gbp_model(px=px, tenors=tenors).curves.GBP_OIS_SONIA
        host, dc = _device_curve(gpu_ctx, curve)
        assert dc.n_pillars == len(px)
        swaps = [F.make_swap(vd, t, 0.045, 1e6 * (i + 1), pay=bool(i % 2)) for i, t in
                 enumerate(("2M", "9M", "3Y", "5Y", "7Y", "19Y", "45Y", "1W", "30M"))]
        _check(gpu_ctx, dc, host, 4, compile_ois(swaps, vd), f"{len(px)} pillars")
