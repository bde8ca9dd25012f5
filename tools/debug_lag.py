#!/usr/bin/env python3
"""Debug aid: payment-lag batch on the GPU vs oracle/port.c, failures broken down by trade shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve
from adrates_amd.trades.compiler import OISTerms, compile_ois_terms
from adrates_amd.utils import BusDayAdjustTypes, CurrencyTypes, CurveTypes, DayCountTypes, FrequencyTypes, InterpTypes
from adrates_amd.trades.market_data import README_VALUE_DT as vd, gbp_model
from oracle import port
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
only = sys.argv[2] if len(sys.argv) > 2 else "mixed"
curve = gbp_model(vd, InterpTypes.LINEAR_ZERO_RATES).curves.GBP_OIS_SONIA
host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
ctx = _native.Context(0)
dc = _native.DeviceCurve(ctx, 4, host.times, host.dfs, host.jac, host.hess)
rng = np.random.default_rng(12)
months = rng.integers(1, 361, n)
lag = rng.choice([0, 1, 2, 5], size=n, p=[0.2, 0.3, 0.4, 0.1])
fidx = rng.choice(3, size=n, p=[0.7, 0.2, 0.1])
if only == "annual":
    fidx[:] = 0; lag[:] = 2
if only.startswith("m") and only != "mixed":
    fidx[:] = 0; lag[:] = 2; months[:] = int(only[1:])
ffreq = [[FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL, FrequencyTypes.QUARTERLY][i] for i in fidx]
spread = np.where(rng.random(n) < 0.3, 0.0015, 0.0)
eff = vd
if only.startswith("f"):
    mm_, fw_ = only[1:].split("_")
    fidx[:] = 0; lag[:] = 2; months[:] = int(mm_); eff = vd.add_months(int(fw_))
terms = OISTerms(effective_dt=eff, tenor=[f"{int(m)}M" for m in months], coupon=rng.uniform(0.01, 0.07, n),
                 notional=np.round(rng.uniform(1e6, 5e7, n), -5), pay_fixed=rng.random(n) < 0.5,
                 fixed_freq_type=FrequencyTypes.ANNUAL, fixed_dc_type=DayCountTypes.ACT_365F,
                 floating_index=CurveTypes.GBP_OIS_SONIA, currency=CurrencyTypes.GBP, float_freq_type=ffreq,
                 float_dc_type=DayCountTypes.ACT_365F, float_spread=spread,
                 payment_lag=lag, bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING)
batch = compile_ois_terms(terms, vd)
n_flt = np.diff(batch.flt_off)
got = _native.price(ctx, dc, _native.DeviceTrades(ctx, batch), aggregate=True)
ref = port.price(4, host.times, host.dfs, host.jac, host.hess, batch)
N = np.abs(batch.notional)
def err(key, floor):
    a = got[key].reshape(n, -1); b = ref[key].reshape(n, -1)
    return np.max(np.abs(a - b), axis=1) / np.maximum(np.max(np.abs(b), axis=1), floor * N)
e_pv, e_d, e_g = err("pv", 1e-4), err("delta", 1e-8), err("gamma", 1e-12)
bad = (e_pv > 1e-10) | (e_d > 1e-10) | (e_g > 1e-10)
print("trades", n, "bad", int(bad.sum()), "pv", int((e_pv > 1e-10).sum()), "delta", int((e_d > 1e-10).sum()), "gamma", int((e_g > 1e-10).sum()))
for f in range(3):
    for lg in (0, 1, 2, 5):
        sel = (fidx == f) & (lag == lg)
        if sel.any():
            print(f"freq {f} lag {lg}: {int(sel.sum()):5d} trades, bad {int((bad & sel).sum()):5d}; n_flt>32 bad {int((bad & sel & (n_flt > 32)).sum())} of {int((sel & (n_flt > 32)).sum())}; n_flt 17-32 bad {int((bad & sel & (n_flt > 16) & (n_flt <= 32)).sum())} of {int((sel & (n_flt > 16) & (n_flt <= 32)).sum())}; <=16 bad {int((bad & sel & (n_flt <= 16)).sum())} of {int((sel & (n_flt <= 16)).sum())}")
print("bad idx", np.flatnonzero(bad)[:40].tolist())
print("spread idx", np.flatnonzero(spread > 0)[:40].tolist())
idx = np.flatnonzero(bad)[:2]
for t in idx:
    d = (got["delta"][t] - ref["delta"][t]); g = got["gamma"][t] - ref["gamma"][t]
    print(f"trade {t}: months {months[t]} n_flt {n_flt[t]} freq {fidx[t]} lag {lag[t]} spread {spread[t]} e_pv {e_pv[t]:.1e} e_d {e_d[t]:.1e} e_g {e_g[t]:.1e}")
    print("   delta diff nz pillars", np.flatnonzero(np.abs(d) > 1e-9 * N[t] * 1e-8)[:12], " ref nz", np.flatnonzero(ref["delta"][t])[:20])
    print("   delta got", np.round(got["delta"][t][14:28], 3)); print("   delta ref", np.round(ref["delta"][t][14:28], 3))
    gi = np.argwhere(np.abs(g) > 1e-10 * N[t] * 1e-8)
    print("   gamma diff entries", len(gi), gi[:8].tolist())
ag = got["agg_gamma"] - ref["gamma"].sum(0)
print("agg gamma max abs diff", np.abs(ag).max(), "rel", np.abs(ag).max() / np.abs(ref["gamma"].sum(0)).max())
