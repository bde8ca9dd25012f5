#!/usr/bin/env python3
"""Debug aid (ADR_LAG_DEBUG build): dump the date-walk records of one annual payment-lag trade."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades.compiler import OISTerms, compile_ois_terms
from adrates_amd.utils import BusDayAdjustTypes, CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes, InterpTypes
from adrates_amd.trades.market_data import README_VALUE_DT as vd, gbp_model
m = int(sys.argv[1]); n = 2
curve = gbp_model(vd, InterpTypes.LINEAR_ZERO_RATES).curves.GBP_OIS_SONIA
host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
ctx = _native.Context(0)
dc = _native.DeviceCurve(ctx, 4, host.times, host.dfs, host.jac, host.hess)
terms = OISTerms(effective_dt=vd, tenor=[f"{m}M"] * n, coupon=np.full(n, 0.04), notional=np.full(n, 1e7), pay_fixed=np.array([True] * n),
                 fixed_freq_type=FrequencyTypes.ANNUAL, fixed_dc_type=DayCountTypes.ACT_365F, floating_index=CurveTypes.GBP_OIS_SONIA,
                 currency=CurrencyTypes.GBP, float_freq_type=FrequencyTypes.ANNUAL, float_dc_type=DayCountTypes.ACT_365F,
                 float_spread=np.zeros(n), payment_lag=2, bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
batch = compile_ois_terms(terms, vd)
got = _native.price(ctx, dc, _native.DeviceTrades(ctx, batch))
g = got["gamma"][0].reshape(-1)
np.set_printoptions(linewidth=250, precision=6, suppress=False)
print("ts", batch.flt_ts[:20]); print("te", batch.flt_te[:20]); print("tp", batch.flt_tp[:20])
print(" i      w_r        x_a        x_b        w_p        p_a       p_b   ca  cb flags      vacc        ua         ub        v_d        v_p       dacc     next_w_r")
for i in range(18):
    r = g[16 * i:16 * i + 16]
    print(f"{i:2d} " + " ".join(f"{v:10.4g}" for v in r))

g1 = got["gamma"][1].reshape(-1)
print(" l   pair_s  pair_e  pair_p tile_prev uses_vacc next_uses regular s_null prev_ok prev_te ts_q e_ok ratio_on om prev_dpair q")
for l in range(24, 32):
    r = g1[16 * l:16 * l + 16]
    print(f"{l:2d} " + " ".join(f"{v:11.6g}" for v in r))
