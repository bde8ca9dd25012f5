"""Enumerations used by the OIS valuation path (cavour/utils/global_types.py:45-99).

Only the members are mirrored; numeric values are identical to the reference so
``InterpTypes.X.value`` can be handed to the C-ABI unchanged (1 = flat forward,
4 = linear zero rates; cavour/market/curves/interpolator_ad.py:227-235).
"""
from enum import Enum

from .currency import CurrencyTypes


class SwapTypes(Enum):
    PAY = 1
    RECEIVE = 2


class InstrumentTypes(Enum):
    SWAP_FIXED_LEG = 1
    SWAP_FLOAT_LEG = 2
    OIS_SWAP = 3
    XCCY_SWAP = 4
    ZCIS = 5
    SWAP_INFLATION_LEG = 6
    BOND = 7
    FRN = 8
    YOY_INFLATION_SWAP = 9
    SWAP_YOY_INFLATION_LEG = 10


class RequestTypes(Enum):
    VALUE = 1
    DELTA = 2
    GAMMA = 3
    SPEED = 4
    CASHFLOWS = 5


class InterpTypes(Enum):
    FLAT_FWD_RATES = 1
    LINEAR_FWD_RATES = 2
    LINEAR_ZERO_RATES = 4
    FINCUBIC_ZERO_RATES = 7
    NATCUBIC_LOG_DISCOUNT = 8
    NATCUBIC_ZERO_RATES = 9
    PCHIP_ZERO_RATES = 10
    PCHIP_LOG_DISCOUNT = 11


class CurveTypes(Enum):
    GBP_OIS_SONIA = 1
    USD_OIS_SOFR = 2
    EUR_OIS_ESTR = 3
    USD_GBP_BASIS = 4
    GBP_RPI_INFLATION = 5
    GBP_CPI_INFLATION = 6
    USD_CPI_INFLATION = 7
    EUR_HICP_INFLATION = 8


class CollateralType(Enum):
    USD = 1
    GBP = 2
    EUR = 3
    JPY = 4
    CHF = 5
    AUD = 6
    CAD = 7
    USD_TIPS = 10
    EUR_OATS = 11
    EUR_BUNDS = 12
    GBP_GILTS = 13
    JGB = 14
    UNCOLLATERALIZED = 99


_COLLATERAL_CCY = {
    CollateralType.USD: CurrencyTypes.USD,
    CollateralType.GBP: CurrencyTypes.GBP,
    CollateralType.EUR: CurrencyTypes.EUR,
    CollateralType.JPY: CurrencyTypes.JPY,
    CollateralType.CHF: CurrencyTypes.CHF,
    CollateralType.AUD: CurrencyTypes.AUD,
    CollateralType.CAD: CurrencyTypes.CAD,
    CollateralType.USD_TIPS: CurrencyTypes.USD,
    CollateralType.EUR_OATS: CurrencyTypes.EUR,
    CollateralType.EUR_BUNDS: CurrencyTypes.EUR,
    CollateralType.GBP_GILTS: CurrencyTypes.GBP,
    CollateralType.JGB: CurrencyTypes.JPY,
}


def collateral_to_currency(collateral_type: CollateralType) -> CurrencyTypes:
    """Currency a collateral type settles in (cavour/utils/global_types.py:154-189)."""
    try:
        return _COLLATERAL_CCY[collateral_type]
    except KeyError:
        raise ValueError(f"Cannot convert {collateral_type} to currency.")
