"""Result containers returned by ``Position.compute`` / ``Portfolio.compute``.

Shapes and arithmetic follow cavour/requests/results.py: `Valuation` :37-164,
`Value` :168-180, `Ladder` :184-225, `Delta` :228-380, `Gamma` :383-605,
`Risk` :839-942, `AnalyticsResult` :1124-1202.  Ladders are numpy float64
arrays (the reference holds jax arrays); export/plot helpers of the reference
are presentation code outside the hot-path scope, only the pandas views that
tests and notebooks touch (`.df`) are kept, importing pandas lazily.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from typing import Any, Dict, Iterable, List, Optional, Tuple, Union

import numpy as np

from ..utils.currency import CurrencyTypes
from ..utils.global_types import CurveTypes


@dataclass(frozen=True)
class Valuation:
    """A monetary amount with its currency; ``+ - * /`` defined for matching
    currencies."""
    amount: float
    currency: CurrencyTypes = CurrencyTypes.NONE

    def __post_init__(self):
        if not isinstance(self.currency, CurrencyTypes):
            raise TypeError(f"currency must be a CurrencyTypes enum, got {type(self.currency)}")

    def __repr__(self) -> str:
        return f"{self.amount:.2f} {self.currency.name}"

    def _same_ccy(self, other, verb):
        if self.currency is not other.currency:
            raise ValueError(f"Cannot {verb} {self.currency.name} and {other.currency.name}")

    def __add__(self, other: Any) -> "Valuation":
        if not isinstance(other, Valuation):
            return NotImplemented
        self._same_ccy(other, "add")
        return Valuation(self.amount + other.amount, self.currency)

    def __radd__(self, other: Any) -> "Valuation":
        if other == 0:  # lets sum() start from 0
            return self
        return self.__add__(other)

    def __sub__(self, other: Any) -> "Valuation":
        if not isinstance(other, Valuation):
            return NotImplemented
        self._same_ccy(other, "subtract")
        return Valuation(self.amount - other.amount, self.currency)

    def __mul__(self, factor: float) -> "Valuation":
        return Valuation(self.amount * factor, self.currency)

    __rmul__ = __mul__

    def __truediv__(self, divisor: float) -> "Valuation":
        return Valuation(self.amount / divisor, self.currency)

    def to_dict(self) -> Dict[str, Any]:
        return {"amount": float(self.amount), "currency": self.currency.name}

    def to_json(self, indent: Optional[int] = 2) -> str:
        return json.dumps(self.to_dict(), indent=indent)

    @property
    def df(self):
        import pandas as pd
        return pd.DataFrame([self.to_dict()])


@dataclass(frozen=True)
class Value:
    """Amount + currency without arithmetic (the ``.value`` of a ladder)."""
    amount: float
    currency: CurrencyTypes = CurrencyTypes.NONE


class Ladder:
    """tenor -> sensitivity mapping with a DataFrame view."""

    def __init__(self, data: Dict[str, float], curve_name: str):
        self.data = data
        self._curve_name = curve_name

    @property
    def df(self):
        import pandas as pd
        out = pd.DataFrame.from_dict(self.data, orient="index",
                                     columns=[f"{self._curve_name}_Risk"])
        out.index.name = "Tenor"
        return out

    def to_dict(self) -> Dict[str, float]:
        return dict(self.data)

    def __repr__(self):
        return f"Ladder(curve={self._curve_name}, points={len(self.data)}, curve_data={self.data})"


def _as_array(x):
    return np.asarray(x, dtype=np.float64)


class _LadderBase:
    """Checks and arithmetic shared by `Delta` and `Gamma`."""

    def _validate(self):
        object.__setattr__(self, "risk_ladder", _as_array(self.risk_ladder))
        n = self.risk_ladder.shape[-1] if self.risk_ladder.ndim else 0
        if n != len(self.tenors):
            raise ValueError(f"Expected {n} tenors, got {len(self.tenors)}")
        if not isinstance(self.currency, CurrencyTypes):
            raise TypeError(f"currency must be CurrencyTypes, got {type(self.currency)}")
        if not isinstance(self.curve_type, CurveTypes):
            raise TypeError(f"curve_type must be CurveTypes, got {type(self.curve_type)}")

    @property
    def value(self) -> Value:
        return Value(amount=float(np.sum(self.risk_ladder)), currency=self.currency)

    def __repr__(self):
        return (f"{type(self).__name__}({self.curve_type.name}: {self.value.amount:.6g} "
                f"{self.currency.name}, points={len(self.tenors)})")

    def _add(self, other):
        if not isinstance(other, type(self)):
            return NotImplemented
        if (self.curve_type != other.curve_type or self.currency != other.currency
                or self.tenors != other.tenors):
            raise ValueError(f"Cannot add {type(self).__name__} with mismatched "
                             "curve_type, currency, or tenors")
        return type(self)(risk_ladder=self.risk_ladder + other.risk_ladder, tenors=self.tenors,
                          currency=self.currency, curve_type=self.curve_type)


@dataclass(frozen=True, repr=False)
class Delta(_LadderBase):
    """Per-pillar first-order sensitivity, value per 1 bp move of each par rate."""
    risk_ladder: np.ndarray
    tenors: List[str]
    currency: CurrencyTypes
    curve_type: CurveTypes

    def __post_init__(self):
        self._validate()

    @property
    def ladder(self) -> Ladder:
        # dict(zip(...)) collapses pillars whose labels collide (1D and 1W are
        # both "1W"), as in the reference (results.py:288-291).
        return Ladder(dict(zip(self.tenors, self.risk_ladder.tolist())), self.curve_type.name)

    def __add__(self, other: Any) -> "Delta":
        return self._add(other)

    __radd__ = __add__

    def to_dict(self) -> Dict[str, Any]:
        return {"risk_ladder": self.risk_ladder.tolist(), "tenors": self.tenors,
                "currency": self.currency.name, "curve_type": self.curve_type.name,
                "total": float(np.sum(self.risk_ladder))}

    def to_json(self, indent: Optional[int] = 2) -> str:
        return json.dumps(self.to_dict(), indent=indent)

    @property
    def df(self):
        return self.ladder.df


@dataclass(frozen=True, repr=False)
class Gamma(_LadderBase):
    """Pillar x pillar second-order sensitivity, value per bp^2."""
    risk_ladder: np.ndarray
    tenors: List[str]
    currency: CurrencyTypes
    curve_type: CurveTypes

    def __post_init__(self):
        self._validate()

    @property
    def to_dict(self) -> dict:
        g = np.asarray(self.risk_ladder)
        if g.ndim != 2:
            raise ValueError("Gamma risk_ladder must be 2D to access matrix")
        return {rt: {ct: float(g[i, j]) for j, ct in enumerate(self.tenors)}
                for i, rt in enumerate(self.tenors)}

    def __add__(self, other: Any) -> "Gamma":
        return self._add(other)

    __radd__ = __add__

    def to_json(self, indent: Optional[int] = 2) -> str:
        return json.dumps({"matrix": self.to_dict, "tenors": self.tenors,
                           "currency": self.currency.name, "curve_type": self.curve_type.name,
                           "total": float(np.sum(self.risk_ladder))}, indent=indent)

    @property
    def df(self):
        import pandas as pd
        g = np.asarray(self.risk_ladder)
        if g.ndim == 1:
            g = np.diag(g)
        return pd.DataFrame(g, index=self.tenors, columns=self.tenors)


@dataclass(frozen=True)
class CrossGamma:
    """Cross-curve second-order sensitivity (cavour/requests/results.py:608-836): ``risk_matrix[i, j]`` =
    d2 PV / d(curve 1 rate i) d(curve 2 rate j), per bp^2."""
    risk_matrix: Any                    # [N1, N2]
    tenors_curve1: Any
    tenors_curve2: Any
    curve_type_1: Any
    curve_type_2: Any
    currency: Any
    # what the matrix holds when it is not the reference's own block (None = as the reference defines it); the
    # cross-currency engine labels its foreign OIS x basis matrix "direct" (xccy_engine.CROSS_GAMMA_MODE)
    definition: Any = None

    def __post_init__(self):
        arr = np.asarray(self.risk_matrix, dtype=np.float64)
        object.__setattr__(self, "risk_matrix", arr)
        if arr.ndim != 2:
            raise ValueError(f"CrossGamma risk_matrix must be 2D, got {arr.ndim}D")
        n1, n2 = arr.shape
        if n1 != len(self.tenors_curve1):
            raise ValueError(f"Expected {n1} tenors for curve 1, got {len(self.tenors_curve1)}")
        if n2 != len(self.tenors_curve2):
            raise ValueError(f"Expected {n2} tenors for curve 2, got {len(self.tenors_curve2)}")

    @property
    def value(self):
        return Value(amount=float(np.sum(self.risk_matrix)), currency=self.currency)

    @property
    def to_dict(self) -> dict:
        return {t1: {t2: float(self.risk_matrix[i, j]) for j, t2 in enumerate(self.tenors_curve2)}
                for i, t1 in enumerate(self.tenors_curve1)}


class Risk:
    """Several per-curve ladders addressed by curve name or `CurveTypes`."""

    def __init__(self, ladders: Iterable[Union[Delta, Gamma]], cross_gammas=None):
        self._by_curve: Dict[str, Union[Delta, Gamma]] = {}
        self._cross_gammas: Dict[Tuple[str, str], Any] = {}
        for ladder in ladders:
            name = ladder.curve_type.name
            if name in self._by_curve:
                raise ValueError(f"Duplicate curve {name}")
            self._by_curve[name] = ladder
            setattr(self, name, ladder)
        for cg in (cross_gammas or []):
            key = (cg.curve_type_1.name, cg.curve_type_2.name)
            if key in self._cross_gammas:
                raise ValueError(f"Duplicate cross-gamma for {key}")
            self._cross_gammas[key] = cg

    def __call__(self, curve_type: CurveTypes):
        try:
            return self._by_curve[curve_type.name]
        except KeyError:
            raise ValueError(f"No risk data for curve: {curve_type.name}")

    def cross_gamma(self, curve_type_1: CurveTypes, curve_type_2: CurveTypes):
        return self._cross_gammas.get((curve_type_1.name, curve_type_2.name))

    def has_cross_gamma(self, curve_type_1: CurveTypes, curve_type_2: CurveTypes) -> bool:
        return (curve_type_1.name, curve_type_2.name) in self._cross_gammas

    @property
    def all_cross_gammas(self):
        return dict(self._cross_gammas)

    def __repr__(self):
        parts = [f"{n}={o.value.amount:.6g} {o.value.currency.name}" for n, o in self._by_curve.items()]
        return f"Risk({', '.join(parts)})"


@dataclass(frozen=True)
class CashflowItem:
    """One payment of a leg as last valued (cavour/requests/results.py:946-995): ``amount`` and
    ``discounted_amount`` are signed from the holder's side (pay legs negative), ``payment_fraction`` is
    the unsigned leg amount per unit notional, ``discount_factor`` is relative to the valuation date."""
    payment_date: Any
    notional: float
    payment_fraction: float
    accrual_period: float
    amount: float
    discount_factor: float
    discounted_amount: float
    leg_type: str          # "Fixed_Pay", "Fixed_Rec", "Float_Pay", "Float_Rec", "Notional_..."

    def to_dict(self) -> Dict[str, Any]:
        d = {k: float(getattr(self, k)) for k in ("notional", "payment_fraction", "accrual_period", "amount",
                                                  "discount_factor", "discounted_amount")}
        return {"payment_date": str(self.payment_date), **d, "leg_type": self.leg_type}


class Cashflows:
    """The cash flows of a trade with filters and totals (cavour/requests/results.py:998-1121)."""

    def __init__(self, cashflows: List[CashflowItem], currency: CurrencyTypes):
        self.cashflows = cashflows
        self.currency = currency

    def validate(self) -> bool:
        if not isinstance(self.cashflows, list):
            raise ValueError("cashflows must be a list")
        if not all(isinstance(cf, CashflowItem) for cf in self.cashflows):
            raise ValueError("All items must be CashflowItem instances")
        return True

    @property
    def total_amount(self) -> float:
        return sum(cf.amount for cf in self.cashflows)

    @property
    def total_pv(self) -> float:
        return sum(cf.discounted_amount for cf in self.cashflows)

    def _where(self, tag: str) -> "Cashflows":
        return Cashflows([cf for cf in self.cashflows if tag in cf.leg_type], self.currency)

    def fixed(self) -> "Cashflows":
        return self._where("Fixed")

    def floating(self) -> "Cashflows":
        return self._where("Float")

    def pay(self) -> "Cashflows":
        return self._where("Pay")

    def receive(self) -> "Cashflows":
        return self._where("Rec")

    def notional_exchange(self) -> "Cashflows":
        return self._where("Notional")

    def sum(self) -> Valuation:
        return Valuation(amount=self.total_pv, currency=self.currency)

    def to_dict(self) -> Dict[str, Any]:
        return {"currency": self.currency.name, "cashflows": [cf.to_dict() for cf in self.cashflows],
                "total_amount": float(self.total_amount), "total_pv": float(self.total_pv),
                "count": len(self.cashflows)}

    def to_json(self, indent: Optional[int] = 2) -> str:
        return json.dumps(self.to_dict(), indent=indent)

    @property
    def df(self):
        import pandas as pd
        if not self.cashflows:
            return pd.DataFrame()
        return pd.DataFrame([cf.to_dict() for cf in self.cashflows]).set_index("payment_date")

    def __len__(self) -> int:
        return len(self.cashflows)

    def __repr__(self) -> str:
        return f"Cashflows(count={len(self.cashflows)}, total_pv={self.total_pv:,.2f} {self.currency.name})"


class AnalyticsResult:
    """What ``compute`` returns: ``.value`` (Valuation), ``.risk`` (a `Delta`
    for a natural-currency OIS, cavour/market/position/engine.py:215), ``.gamma``
    and ``.cashflows``; anything not requested is ``None``."""

    def __init__(self, value: Optional[Valuation] = None, risk=None,
                 gamma: Optional[Gamma] = None, cashflows=None):
        self._value = value
        self._risk = risk
        self._gamma = gamma
        self._cashflows = cashflows

    @property
    def value(self):
        return self._value

    @property
    def risk(self):
        return self._risk

    @property
    def gamma(self):
        return self._gamma

    @property
    def cashflows(self):
        return self._cashflows

    def __repr__(self):
        parts = []
        if self._value is not None:
            parts.append(f"value={self._value!r}")
        if self._risk is not None:
            parts.append(f"risk={self._risk!r}")
        if self._gamma is not None:
            parts.append(f"gamma={self._gamma!r}")
        if self._cashflows is not None:
            parts.append(f"cashflows={self._cashflows!r}")
        return f"AnalyticsResult({', '.join(parts)})"
