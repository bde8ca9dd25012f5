"""A derivative bound to a market model (cavour/market/position/position.py:25-80)."""
from .engine import Engine


class Position:
    def __init__(self, derivative, model):
        self.derivative = derivative
        self.model = model
        self._engine = Engine(model)

    def compute(self, request_list, collateral_type=None):
        """VALUE / DELTA / GAMMA of this position as an `AnalyticsResult`."""
        return self._engine.compute(self.derivative, request_list, collateral_type)
