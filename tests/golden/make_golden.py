#!/usr/bin/env python3
"""Generates tests/golden/ois_golden.json with the CPU oracle (oracle/cavour_oracle.py).

The reference itself cannot be imported in the build container (no jax/numba/xbbg; SURVEY.md section
8(c)), so these vectors come from the oracle, which is pinned to the reference's notebook outputs and
test properties by tests/test_oracle_pins.py.  Inputs are the reference's own market data
(README.md:69-78, tests/test_ois_request_types.py:36-130) and the trades its README / tests / notebook
use; run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from adrates_amd.utils import DayCountTypes, CurveTypes, CurrencyTypes, FrequencyTypes, InterpTypes  # noqa: E402
from adrates_amd.utils.helpers import times_from_dates  # noqa: E402
from oracle import cavour_oracle as O  # noqa: E402
from tests import _fixtures as F  # noqa: E402

CASES = [
    # (case id, curve, value date, interp, trades[(tenor, coupon, notional, pay, extra kwargs)])
    ("readme_gbp_lzr", "gbp", (30, 4, 2024), "LINEAR_ZERO_RATES", [
        ("1W", 0.052014, 1e6, True, {}),            # notebooks/intro.ipynb cells 23-44
        ("10Y", 0.045, 1e7, True, {}),              # README.md:111-131
        ("87M", 0.04, 1e7, False, {}),              # off-grid, front stub
        ("3M", 0.05, 1e6, True, {}),
        ("50Y", 0.0388, 1e6, True, {}),
        ("55Y", 0.039, 5e6, False, {}),             # beyond the last knot
        ("30M", 0.045, 4e6, False, {"payment_lag": 1, "spread": 0.001}),
        ("5Y", 0.04, 1e6, True, {"fixed_freq": "SEMI_ANNUAL"}),
    ]),
    ("readme_gbp_ffr", "gbp", (30, 4, 2024), "FLAT_FWD_RATES", [
        ("10Y", 0.045, 1e7, True, {}),
        ("87M", 0.04, 1e7, False, {}),
        ("55Y", 0.039, 5e6, False, {}),
    ]),
    ("tests_gbp_lzr", "gbp", (17, 12, 2024), "LINEAR_ZERO_RATES", [   # tests/test_ois_request_types.py trades
        ("10Y", 0.045, 1e6, True, {}),
        ("15Y", 0.04, 1e6, True, {}),
        ("5Y", 0.045, 1e6, False, {}),
        ("3M", 0.05, 1e6, True, {}),
        ("50Y", 0.04, 1e6, True, {}),
    ]),
    ("tests_usd_lzr", "usd", (17, 12, 2024), "LINEAR_ZERO_RATES", [   # ACT/360 curve, :87-130
        ("2Y", 0.045, 1e6, True, {}),
        ("10Y", 0.043, 1e6, False, {}),
    ]),
]


def build(case):
    cid, ccy, (d, m, y), interp, trades = case
    from adrates_amd.utils import Date
    vd = Date(d, m, y)
    it = InterpTypes[interp]
    if ccy == "gbp":
        model, name = F.gbp_model(vd, it), "GBP_OIS_SONIA"
        kw = dict(dc=DayCountTypes.ACT_365F, index=CurveTypes.GBP_OIS_SONIA, ccy=CurrencyTypes.GBP)
    else:
        model, name = F.usd_model(vd, it), "USD_OIS_SOFR"
        kw = dict(dc=DayCountTypes.ACT_360, index=CurveTypes.USD_OIS_SOFR, ccy=CurrencyTypes.USD)
    curve = model.curves[name]
    swaps = []
    for tenor, cpn, notional, pay, extra in trades:
        extra = dict(extra)
        if "fixed_freq" in extra:
            extra["fixed_freq"] = FrequencyTypes[extra["fixed_freq"]]
        swaps.append(F.make_swap(vd, tenor, cpn, notional, pay=pay, **kw, **extra))
    return vd, curve, swaps


def main():
    out = {"_about": "expected PV / delta / gamma from oracle/cavour_oracle.py; see make_golden.py", "cases": []}
    for case in CASES:
        vd, curve, swaps = build(case)
        cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
        rows = []
        for spec, s in zip(case[4], swaps):
            fx, fl = O.leg_inputs_from_swap(s, vd, times_from_dates)
            r = O.ois_analytics(cache, curve._interp_type.value, fx, fl)
            rows.append({"tenor": spec[0], "coupon": spec[1], "notional": spec[2], "pay_fixed": spec[3],
                         "extra": spec[4], "pv": r["value"], "delta": r["delta"].tolist(),
                         "gamma": r["gamma"].tolist()})
        out["cases"].append({"id": case[0], "curve": case[1], "value_dt": list(case[2]), "interp": case[3],
                             "n_knots": int(cache["times"].shape[0]),
                             "df_1y": float(cache["dfs"][int(np.searchsorted(cache["times"], 1.0))]),
                             "trades": rows})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ois_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
