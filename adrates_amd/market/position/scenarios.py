"""Scenario batches: many shocked versions of one curve, bootstrapped and priced on the GPU.

The reference shocks a curve by building a whole new `Model` per shock
(`Model.scenario`, cavour/models/models.py:507-557) and re-running the JAX scan,
`jacrev` and `hessian` for each (cavour/market/position/engine.py:2246-2412);
its finite-difference helpers do that twice per bumped tenor
(tests/test_ois_request_types.py:137-207).  A shock moves the par rates only -
schedules, hence the knot grid, stay - so here all shocked curves of a grid are
bootstrapped together by the device builder (csrc/curve_build.hip,
adr_curve_set_build) and every scenario is priced with the ordinary kernels.

Shocks use `Model.scenario`'s convention: a float shifts every quote, a dict
``{tenor: shift}`` only the named ones; shifts are in the quotes' units (percent).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Union

import numpy as np

from ... import _native
from ...trades.compiler import compile_ois
from ...utils.error import LibError
from ...utils.global_types import RequestTypes
from .engine import _SUPPORTED_INTERP
from ..curves.curve_tables import build_engine_curve

Shock = Union[float, Dict[str, float]]


def shocked_quotes(base_px: Sequence[float], tenors: Sequence[str], shock: Shock) -> List[float]:
    """The quote list `Model.scenario` would build the shocked curve from (models.py:539-546)."""
    if isinstance(shock, dict):
        return [base_px[i] + shock.get(t, 0.0) for i, t in enumerate(tenors)]
    return [px + shock for px in base_px]


class ScenarioGrid:
    """``len(shocks)`` shocked versions of ``model.curves[curve_name]`` resident on the GPU."""

    def __init__(self, model, curve_name: str, shocks: Iterable[Shock], with_gamma: bool = True, ctx=None):
        if curve_name not in model._curve_params_dict:
            raise ValueError(f"No stored parameters found for curve '{curve_name}'")
        params = model._curve_params_dict[curve_name]
        self.model, self.curve_name = model, curve_name
        self.curve = getattr(model.curves, curve_name)
        method = self.curve._interp_type.value
        if method not in _SUPPORTED_INTERP:
            raise LibError("Invalid interpolation scheme.")
        self.shocks = list(shocks)
        # OISCurve stores quote / 100 as the swap's fixed coupon (models.py build_curve); same division here
        self.rates = np.array([[px / 100.0 for px in shocked_quotes(params["px_list"], params["tenor_list"], s)]
                               for s in self.shocks], dtype=np.float64).reshape(len(self.shocks), -1)
        base = build_engine_curve(self.curve.swap_rates, self.curve.swap_times, self.curve.year_fracs,
                                  with_hessian=with_gamma)
        self._ctx = ctx or _native.default_context()
        self._plan = _native.CurvePlan(self._ctx, method, base)
        self._set = self._plan.build(self.rates)
        self.base = base

    def __len__(self):
        return len(self.shocks)

    def device_curve(self, i: int):
        return self._set[i]

    def download(self, i: int):
        """Scenario ``i``'s ``dfs, jac, hess`` - the reference's cache dict for the shocked curve."""
        return self._set.download(i)

    def price(self, derivatives, reqs=(RequestTypes.VALUE,), aggregate: bool = False):
        """Price the trades under every scenario.

        Returns a dict of arrays with a leading scenario axis: ``pv [S, n]``, ``delta [S, n, P]``,
        ``gamma [S, n, P, P]`` (per request), or ``agg_*`` sums over the trades when ``aggregate``."""
        reqs = set(reqs)
        batch = compile_ois(list(derivatives), self.curve._value_dt)
        trades = _native.DeviceTrades(self._ctx, batch)
        outs = []
        try:
            for i in range(len(self)):
                outs.append(_native.price(self._ctx, self._set[i], trades,
                                          want_value=RequestTypes.VALUE in reqs,
                                          want_delta=RequestTypes.DELTA in reqs,
                                          want_gamma=RequestTypes.GAMMA in reqs,
                                          per_trade=not aggregate, aggregate=aggregate))
        finally:
            trades.close()
        keys = outs[0].keys() if outs else ()
        return {k: np.stack([np.asarray(o[k]) for o in outs]) for k in keys}

    def close(self):
        self._set.close()
        self._plan.close()


def bump_ladder(tenors: Sequence[str], bump_bp: float = 1.0) -> List[Shock]:
    """Shocks for central finite differences: base, then +/- ``bump_bp`` on each tenor in turn
    (tests/test_ois_request_types.py:171-207 does these one `Model.scenario` at a time)."""
    h = bump_bp * 0.01            # quotes are in percent: 1 bp = 0.01
    shocks: List[Shock] = [0.0]
    for t in tenors:
        shocks.append({t: h})
        shocks.append({t: -h})
    return shocks


def finite_difference_delta(grid_values: np.ndarray, bump_bp: float = 1.0) -> np.ndarray:
    """Per-tenor central differences from values priced on a `bump_ladder` grid: [n, P] per 1 bp."""
    up, down = grid_values[1::2], grid_values[2::2]
    return ((up - down) / (2.0 * bump_bp)).T
