"""Cross-currency basis swap: two floating legs with notional exchange at both ends.

Host-side trade object of SURVEY.md section 8(f) row 1 (cavour/trades/rates/xccy_basis_swap.py:67-304): the
domestic leg is received, the foreign leg - which carries the basis spread - is paid; both exchange notional
at the effective date (-N) and at maturity (+N).  `value()` discounts the foreign leg on the XCCY curve (domestic
collateral, the default) or the domestic leg on the inverted XCCY curve (foreign collateral) and converts with
`spot_fx` as the reference does: ``domestic + foreign / spot_fx`` resp. ``domestic * spot_fx + foreign``.
"""
from ...utils.calendar import BusDayAdjustTypes, Calendar, CalendarTypes, DateGenRuleTypes
from ...utils.currency import CurrencyTypes
from ...utils.date import Date
from ...utils.day_count import DayCountTypes
from ...utils.error import LibError
from ...utils.frequency import FrequencyTypes
from ...utils.global_types import CurveTypes, InstrumentTypes, SwapTypes, collateral_to_currency
from ...utils.helpers import check_argument_types
from .swap_float_leg import SwapFloatLeg


class XccyBasisSwap:
    def __init__(self,
                 effective_dt: Date,
                 term_dt_or_tenor: (Date, str),
                 domestic_notional: float,
                 foreign_notional: float,
                 domestic_spread: float,
                 foreign_spread: float,
                 domestic_freq_type: FrequencyTypes,
                 foreign_freq_type: FrequencyTypes,
                 domestic_dc_type: DayCountTypes,
                 foreign_dc_type: DayCountTypes,
                 domestic_floating_index: CurveTypes,
                 foreign_floating_index: CurveTypes,
                 domestic_currency: CurrencyTypes,
                 foreign_currency: CurrencyTypes,
                 domestic_payment_lag: int = 0,
                 foreign_payment_lag: int = 0,
                 domestic_cal_type: CalendarTypes = CalendarTypes.WEEKEND,
                 foreign_cal_type: CalendarTypes = CalendarTypes.WEEKEND,
                 domestic_bd_type: BusDayAdjustTypes = BusDayAdjustTypes.FOLLOWING,
                 foreign_bd_type: BusDayAdjustTypes = BusDayAdjustTypes.FOLLOWING,
                 domestic_dg_type: DateGenRuleTypes = DateGenRuleTypes.BACKWARD,
                 foreign_dg_type: DateGenRuleTypes = DateGenRuleTypes.BACKWARD,
                 domestic_end_of_month: bool = False,
                 foreign_end_of_month: bool = False):
        check_argument_types(self.__init__, locals())
        self.derivative_type = InstrumentTypes.XCCY_SWAP
        self._termination_dt = (term_dt_or_tenor if isinstance(term_dt_or_tenor, Date)
                                else effective_dt.add_tenor(term_dt_or_tenor))
        # the domestic calendar adjusts the maturity (xccy_basis_swap.py:143-145)
        self._maturity_dt = Calendar(domestic_cal_type).adjust(self._termination_dt, domestic_bd_type)
        if effective_dt > self._maturity_dt:
            raise LibError("Start date after maturity date")
        self._effective_dt = effective_dt
        self._domestic_notional = domestic_notional
        self._foreign_notional = foreign_notional
        self._domestic_currency = domestic_currency
        self._foreign_currency = foreign_currency
        self._domestic_floating_index = domestic_floating_index
        self._foreign_floating_index = foreign_floating_index
        self._domestic_leg = SwapFloatLeg(effective_dt, self._termination_dt, SwapTypes.RECEIVE, domestic_spread,
                                          domestic_freq_type, domestic_dc_type, domestic_floating_index,
                                          domestic_currency, domestic_notional, 0.0, domestic_payment_lag,
                                          domestic_cal_type, domestic_bd_type, domestic_dg_type,
                                          domestic_end_of_month, True)
        self._foreign_leg = SwapFloatLeg(effective_dt, self._termination_dt, SwapTypes.PAY, foreign_spread,
                                         foreign_freq_type, foreign_dc_type, foreign_floating_index,
                                         foreign_currency, foreign_notional, 0.0, foreign_payment_lag,
                                         foreign_cal_type, foreign_bd_type, foreign_dg_type,
                                         foreign_end_of_month, True)
        self._domestic_spread = domestic_spread
        self._foreign_spread = foreign_spread
        self._adjusted_domestic_dts = self._domestic_leg._payment_dts
        self._adjusted_foreign_dts = self._foreign_leg._payment_dts

    def value(self, value_dt: Date, domestic_discount_curve, foreign_discount_curve, xccy_discount_curve=None,
              xccy_discount_curve_inverted=None, spot_fx: float = None, collateral_type=None,
              first_fixing_rate_domestic: float = None, first_fixing_rate_foreign: float = None):
        """PV in the collateral currency (xccy_basis_swap.py:209-304)."""
        collateral_ccy = (self._domestic_currency if collateral_type is None
                          else collateral_to_currency(collateral_type))
        if collateral_ccy == self._domestic_currency:
            dom_disc, for_disc = domestic_discount_curve, xccy_discount_curve
            if for_disc is None:
                raise ValueError(f"xccy_discount_curve required for domestic collateral "
                                 f"({self._domestic_currency.name})")
        elif collateral_ccy == self._foreign_currency:
            dom_disc, for_disc = xccy_discount_curve_inverted, foreign_discount_curve
            if dom_disc is None:
                raise ValueError(f"xccy_discount_curve_inverted required for foreign collateral "
                                 f"({self._foreign_currency.name})")
        else:
            raise ValueError(f"Third-party collateral not yet supported: {collateral_type}. Only "
                             f"{self._domestic_currency.name} or {self._foreign_currency.name} collateral allowed.")
        dom = self._domestic_leg.value(value_dt, dom_disc, domestic_discount_curve, first_fixing_rate_domestic)
        frn = self._foreign_leg.value(value_dt, for_disc, foreign_discount_curve, first_fixing_rate_foreign)
        if collateral_ccy == self._domestic_currency:
            return dom + frn / spot_fx
        return dom * spot_fx + frn

    def position(self, model):
        from ...market.position.position import Position
        return Position(self, model)

    def __repr__(self):
        return (f"XccyBasisSwap({self._effective_dt} -> {self._maturity_dt}, "
                f"{self._domestic_currency.name} {self._domestic_notional:,.0f} vs "
                f"{self._foreign_currency.name} {self._foreign_notional:,.0f}, "
                f"foreign spread {self._foreign_spread:.6f})")
