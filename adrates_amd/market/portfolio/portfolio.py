"""Portfolio of positions (cavour/market/portfolio/portfolio.py:8-66).

The reference prices position by position in a Python loop and adds the result
objects; here positions that share a curve go to the GPU as one batch and the
sums are formed on the device (per-block partials, fixed-order final sum).
Mixed portfolios fall back to combining the per-curve aggregates with the same
``+`` the reference uses, so mismatched curves/currencies raise as they do there.
"""
from typing import Iterable, List

from ...requests.results import AnalyticsResult
from ...utils.global_types import InstrumentTypes, RequestTypes
from ..position.engine import Engine, price_batch, wrap_result
from ..position.position import Position


class Portfolio:
    def __init__(self, positions: Iterable[Position] | None = None) -> None:
        self._positions: List[Position] = list(positions or [])

    def add_position(self, position: Position) -> None:
        self._positions.append(position)

    def positions(self) -> List[Position]:
        return list(self._positions)

    def compute(self, request_list: Iterable[RequestTypes]) -> AnalyticsResult:
        """Aggregate VALUE / DELTA / GAMMA over all positions."""
        reqs = set(request_list)
        groups = {}   # (model id, curve name, currency) -> positions, in first-seen order
        singles = []  # everything that is not an OIS: priced one by one and added with `+`, as the reference does
        for pos in self._positions:
            d = pos.derivative
            if d.derivative_type != InstrumentTypes.OIS_SWAP:
                singles.append(pos)
                continue
            key = (id(pos.model), d._floating_index, d._currency)
            groups.setdefault(key, []).append(pos)

        total_val = total_delta = total_gamma = None
        for pos in singles:
            # cross-currency swaps return `Risk` containers, which have no `+` in the reference either
            # (requests/results.py:839-942): two of them with DELTA / GAMMA raise TypeError there and here;
            # books of XCCY swaps go through xccy_engine.price_xccy_batch(aggregate=True) instead
            res = pos.compute(request_list)
            if RequestTypes.VALUE in reqs:
                total_val = res.value if total_val is None else total_val + res.value
            if RequestTypes.DELTA in reqs:
                total_delta = res.risk if total_delta is None else total_delta + res.risk
            if RequestTypes.GAMMA in reqs:
                total_gamma = res.gamma if total_gamma is None else total_gamma + res.gamma
        for (_, curve_type, currency), members in groups.items():
            model = members[0].model
            ir_model = getattr(model.curves, curve_type.name)
            engine = members[0]._engine
            res = price_batch(engine, ir_model, [p.derivative for p in members], reqs,
                              per_trade=False, aggregate=True)
            part = wrap_result(res, 0, reqs, res["tenors"], currency, curve_type, aggregate=True)
            if RequestTypes.VALUE in reqs:
                total_val = part.value if total_val is None else total_val + part.value
            if RequestTypes.DELTA in reqs:
                total_delta = part.risk if total_delta is None else total_delta + part.risk
            if RequestTypes.GAMMA in reqs:
                total_gamma = part.gamma if total_gamma is None else total_gamma + part.gamma
        return AnalyticsResult(value=total_val, risk=total_delta, gamma=total_gamma)
