"""Second-order forward-mode numbers ("jets"): a value with its gradient and Hessian w.r.t. n inputs.

Used by the host-side curve bootstraps that have to return first and second derivatives of every node -
where the reference differentiates a `lax.scan` with `jacrev` / `jacfwd` (cavour/trades/rates/xccy_curve.py:
529-703), the same recurrence is simply evaluated on jets.  Only the operations those recurrences need:
+, -, *, /, exp.  Hessians are kept symmetric.
"""
from __future__ import annotations

import numpy as np


class Jet:
    __slots__ = ("v", "g", "h")

    def __init__(self, v: float, g: np.ndarray, h: np.ndarray):
        self.v, self.g, self.h = float(v), g, h

    # ---- construction
    @staticmethod
    def const(v: float, n: int) -> "Jet":
        return Jet(v, np.zeros(n), np.zeros((n, n)))

    @staticmethod
    def variable(v: float, i: int, n: int) -> "Jet":
        g = np.zeros(n)
        g[i] = 1.0
        return Jet(v, g, np.zeros((n, n)))

    def _lift(self, other) -> "Jet":
        return other if isinstance(other, Jet) else Jet.const(other, self.g.shape[0])

    # ---- arithmetic
    def __add__(self, other):
        o = self._lift(other)
        return Jet(self.v + o.v, self.g + o.g, self.h + o.h)

    __radd__ = __add__

    def __neg__(self):
        return Jet(-self.v, -self.g, -self.h)

    def __sub__(self, other):
        o = self._lift(other)
        return Jet(self.v - o.v, self.g - o.g, self.h - o.h)

    def __rsub__(self, other):
        return self._lift(other) - self

    def __mul__(self, other):
        if not isinstance(other, Jet):
            return Jet(self.v * other, self.g * other, self.h * other)
        cross = np.outer(self.g, other.g)
        return Jet(self.v * other.v, self.v * other.g + other.v * self.g,
                   self.v * other.h + other.v * self.h + cross + cross.T)

    __rmul__ = __mul__

    def reciprocal(self):
        r = 1.0 / self.v
        return Jet(r, -r * r * self.g, 2.0 * r ** 3 * np.outer(self.g, self.g) - r * r * self.h)

    def __truediv__(self, other):
        if not isinstance(other, Jet):
            return self * (1.0 / other)
        return self * other.reciprocal()

    def __rtruediv__(self, other):
        return self.reciprocal() * other

    def exp(self):
        e = float(np.exp(self.v))
        return Jet(e, e * self.g, e * (self.h + np.outer(self.g, self.g)))

    def __repr__(self):
        return f"Jet({self.v!r}, n={self.g.shape[0]})"
