#!/usr/bin/env python3
"""The reference README's sections 1-3 (build a curve, price an OIS with VALUE / DELTA / GAMMA, aggregate a
portfolio) with `cavour.` replaced by `adrates_amd.`, followed by what this implementation adds on the same path:
a scenario grid bootstrapped and priced on the GPU, the CASHFLOWS request and the vectorised trade compiler.

Run on an MI355X after `python -c "import __graft_entry__ as g; g.build()"`:  python examples/quickstart.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from adrates_amd.market.curves.interpolator import InterpTypes
from adrates_amd.market.portfolio.portfolio import Portfolio
from adrates_amd.market.position.scenarios import ScenarioGrid, bump_ladder, finite_difference_delta
from adrates_amd.models.models import Model
from adrates_amd.trades.compiler import OISTerms, compile_ois_terms
from adrates_amd.trades.rates.ois import OIS
from adrates_amd.utils.calendar import BusDayAdjustTypes
from adrates_amd.utils.currency import CurrencyTypes
from adrates_amd.utils.date import Date
from adrates_amd.utils.day_count import DayCountTypes
from adrates_amd.utils.frequency import FrequencyTypes
from adrates_amd.utils.global_types import CurveTypes, RequestTypes, SwapTypes

# ---- 1. curve (README.md:60-101)
value_dt = Date(30, 4, 2024)
px_list = [5.1998, 5.2014, 5.2003, 5.2027, 5.2023, 5.19281, 5.1656, 5.1482, 5.1342, 5.1173, 5.1013, 5.0862, 5.0701,
           5.054, 5.0394, 4.8707, 4.75483, 4.532, 4.3628, 4.2428, 4.16225, 4.1132, 4.08505, 4.0762, 4.078, 4.0961,
           4.12195, 4.1315, 4.113, 4.07724, 3.984, 3.88]
tenor_list = ["1D", "1W", "2W", "1M", "2M", "3M", "4M", "5M", "6M", "7M", "8M", "9M", "10M", "11M", "12M", "18M", "2Y",
              "3Y", "4Y", "5Y", "6Y", "7Y", "8Y", "9Y", "10Y", "12Y", "15Y", "20Y", "25Y", "30Y", "40Y", "50Y"]
model = Model(value_dt)
model.build_curve(name="GBP_OIS_SONIA", px_list=px_list, tenor_list=tenor_list, spot_days=0,
                  swap_type=SwapTypes.PAY, fixed_dcc_type=DayCountTypes.ACT_365F,
                  fixed_freq_type=FrequencyTypes.ANNUAL, float_freq_type=FrequencyTypes.ANNUAL,
                  float_dc_type=DayCountTypes.ACT_365F, bus_day_type=BusDayAdjustTypes.MODIFIED_FOLLOWING,
                  interp_type=InterpTypes.LINEAR_ZERO_RATES)
curve = model.curves.GBP_OIS_SONIA
print(f"5Y discount factor: {curve.df_ad(5.0):.6f}")

# ---- 2. one swap: VALUE, DELTA, GAMMA (README.md:105-160)
swap = OIS(effective_dt=value_dt, term_dt_or_tenor="10Y", fixed_leg_type=SwapTypes.PAY, fixed_coupon=0.045,
           fixed_freq_type=FrequencyTypes.ANNUAL, fixed_dc_type=DayCountTypes.ACT_365F,
           floating_index=CurveTypes.GBP_OIS_SONIA, currency=CurrencyTypes.GBP, notional=10_000_000,
           bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING, float_freq_type=FrequencyTypes.ANNUAL,
           float_dc_type=DayCountTypes.ACT_365F)
res = swap.position(model).compute([RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA, RequestTypes.CASHFLOWS])
print(f"PV {res.value.amount:,.2f} {res.value.currency.name}   total delta {res.risk.value.amount:,.4f} per bp"
      f"   total gamma {res.gamma.value.amount:.6f} per bp^2")
print("10Y bucket delta:", dict(zip(res.risk.tenors, res.risk.risk_ladder))["10Y"])
print(res.cashflows, "| fixed leg PV", f"{res.cashflows.fixed().total_pv:,.2f}")

# ---- 3. portfolio (README.md:164-230)
others = [OIS(value_dt, t, SwapTypes.RECEIVE, c, FrequencyTypes.ANNUAL, DayCountTypes.ACT_365F, CurveTypes.GBP_OIS_SONIA,
              CurrencyTypes.GBP, notional=n, bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING,
              float_freq_type=FrequencyTypes.ANNUAL, float_dc_type=DayCountTypes.ACT_365F)
          for t, c, n in (("87M", 0.04, 1e7), ("3M", 0.05, 2e6), ("30Y", 0.039, 5e6))]
book = Portfolio([s.position(model) for s in [swap] + others]).compute([RequestTypes.VALUE, RequestTypes.DELTA,
                                                                        RequestTypes.GAMMA])
print(f"portfolio PV {book.value.amount:,.2f}, delta {book.risk.value.amount:,.4f}, gamma {book.gamma.value.amount:.6f}")

# ---- scenario grid: 65 shocked curves bootstrapped (with Jacobians) and priced on the GPU
grid = ScenarioGrid(model, "GBP_OIS_SONIA", bump_ladder(tenor_list, 1.0), with_gamma=False)
pv = grid.price([swap] + others, [RequestTypes.VALUE])["pv"]
fd = finite_difference_delta(pv, 1.0)
print("bump-and-reprice vs analytic 10Y delta of the first swap:", fd[0][24], "vs", res.risk.risk_ladder[24])
grid.close()

# ---- a million trades from their terms, without a million Python objects
n = 1_000_000
rng = np.random.default_rng(1)
t0 = time.perf_counter()
months = rng.integers(1, 361, n)
names = {m: f"{m}M" for m in range(1, 361)}
batch = compile_ois_terms(OISTerms(value_dt, [names[int(m)] for m in months], rng.uniform(0.01, 0.07, n),
                                   np.round(rng.uniform(1e6, 5e7, n), -5), rng.random(n) < 0.5, FrequencyTypes.ANNUAL,
                                   DayCountTypes.ACT_365F, CurveTypes.GBP_OIS_SONIA, CurrencyTypes.GBP,
                                   float_freq_type=FrequencyTypes.ANNUAL, float_dc_type=DayCountTypes.ACT_365F,
                                   bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING), value_dt)
print(f"compiled {batch.n_trades:,} trades in {time.perf_counter() - t0:.1f} s")
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
ctx = _native.default_context()
dc = _native.DeviceCurve(ctx, curve._interp_type.value, host.times, host.dfs, host.jac, host.hess)
dt = _native.DeviceTrades(ctx, batch)
t0 = time.perf_counter()
agg = _native.price(ctx, dc, dt, per_trade=False, aggregate=True)
print(f"portfolio ladder of {n:,} trades (PV + delta + gamma, aggregated on the device) in "
      f"{1e3 * (time.perf_counter() - t0):.1f} ms: PV {agg['agg_pv']:,.0f}, delta {agg['agg_delta'].sum():,.1f}")

# ---- cross-currency: a USD curve, the GBP/USD basis curve, a basis swap and an OIS under USD collateral
from adrates_amd.trades.rates.xccy_basis_swap import XccyBasisSwap
from adrates_amd.utils import CollateralType
usd_px = [5.35, 5.32, 5.31, 5.29, 5.27, 5.25, 5.23, 5.21, 5.19, 5.17, 5.15, 5.13, 5.11, 5.09, 5.07, 4.95, 4.85, 4.70,
          4.58, 4.48, 4.41, 4.36, 4.32, 4.29, 4.27, 4.28, 4.30, 4.32, 4.31, 4.29, 4.24, 4.18]
model.build_curve(name="USD_OIS_SOFR", px_list=usd_px, tenor_list=tenor_list, spot_days=0, swap_type=SwapTypes.PAY,
                  fixed_dcc_type=DayCountTypes.ACT_360, fixed_freq_type=FrequencyTypes.ANNUAL,
                  float_freq_type=FrequencyTypes.ANNUAL, float_dc_type=DayCountTypes.ACT_360,
                  bus_day_type=BusDayAdjustTypes.MODIFIED_FOLLOWING, interp_type=InterpTypes.FLAT_FWD_RATES)
basis_tenors = ["1Y", "2Y", "3Y", "5Y", "7Y", "10Y", "15Y", "20Y", "30Y"]
basis_bp = [25.0, 28.0, 30.0, 34.0, 36.0, 39.0, 42.0, 45.0, 48.0]
for name, dom, frn, spreads, fx in (("USD_GBP_BASIS", "GBP_OIS_SONIA", "USD_OIS_SOFR", basis_bp, 0.79),
                                    ("GBP_USD_XCCY", "USD_OIS_SOFR", "GBP_OIS_SONIA", [-b for b in basis_bp], 1 / 0.79)):
    model.build_xccy_curve(name=name, domestic_curve_name=dom, foreign_curve_name=frn, basis_spreads=spreads,
                           tenor_list=basis_tenors, spot_fx=fx,
                           domestic_dc_type=DayCountTypes.ACT_365F if dom.startswith("GBP") else DayCountTypes.ACT_360,
                           foreign_dc_type=DayCountTypes.ACT_360 if dom.startswith("GBP") else DayCountTypes.ACT_365F,
                           interp_type=InterpTypes.FLAT_FWD_RATES)
xccy = XccyBasisSwap(effective_dt=value_dt, term_dt_or_tenor="7Y", domestic_notional=7_900_000, foreign_notional=10_000_000,
                     domestic_spread=0.0, foreign_spread=0.0040, domestic_freq_type=FrequencyTypes.ANNUAL,
                     foreign_freq_type=FrequencyTypes.SEMI_ANNUAL, domestic_dc_type=DayCountTypes.ACT_365F,
                     foreign_dc_type=DayCountTypes.ACT_360, domestic_floating_index=CurveTypes.GBP_OIS_SONIA,
                     foreign_floating_index=CurveTypes.USD_OIS_SOFR, domestic_currency=CurrencyTypes.GBP,
                     foreign_currency=CurrencyTypes.USD)
x = xccy.position(model).compute([RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA])
print(f"7Y GBP/USD basis swap: PV {x.value.amount:,.2f} GBP; delta per bp - SONIA {x.risk.GBP_OIS_SONIA.value.amount:,.2f}, "
      f"SOFR {x.risk.USD_OIS_SOFR.value.amount:,.2f}, basis {x.risk.USD_GBP_BASIS.value.amount:,.2f}; "
      f"basis gamma {x.gamma.USD_GBP_BASIS.value.amount:.4f}")
cross = x.gamma.cross_gamma(CurveTypes.USD_OIS_SOFR, CurveTypes.USD_GBP_BASIS)     # d2 PV / d(SOFR quote) d(basis spread), per bp^2
print(f"foreign OIS x basis cross-gamma: {cross.risk_matrix.shape[0]} x {cross.risk_matrix.shape[1]} ladder, total "
      f"{cross.value.amount:.6f} {cross.value.currency.name}")
c = swap.position(model).compute([RequestTypes.VALUE, RequestTypes.DELTA], collateral_type=CollateralType.USD)
print(f"the 10Y OIS under USD collateral: PV {c.value.amount:,.2f} {c.value.currency.name} "
      f"(vs {res.value.amount / 0.79:,.2f} converted at spot), basis delta {c.risk.USD_GBP_BASIS.value.amount:,.2f} per bp")
