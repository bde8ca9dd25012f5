"""Overnight index swap: a fixed leg against a floating leg of opposite sign.

Boundary object of the drop-in path; constructor
signature and defaults follow cavour/trades/rates/ois.py:100-195 and
``position(model)`` follows :199-205.  The non-AD `value/pv01/swap_rate` of the
reference price on `OISCurve`'s own node set (a "next" row, SURVEY.md section
8(f) rank 3) and are not part of this path.
"""
from ...utils.calendar import BusDayAdjustTypes, Calendar, CalendarTypes, DateGenRuleTypes
from ...utils.currency import CurrencyTypes
from ...utils.date import Date
from ...utils.day_count import DayCountTypes
from ...utils.error import LibError
from ...utils.frequency import FrequencyTypes
from ...utils.global_types import CurveTypes, InstrumentTypes, SwapTypes
from ...utils.global_vars import ONE_MILLION
from ...utils.helpers import check_argument_types, label_to_string
from .swap_fixed_leg import SwapFixedLeg
from .swap_float_leg import SwapFloatLeg


class OIS:
    def __init__(self,
                 effective_dt: Date,
                 term_dt_or_tenor: (Date, str),
                 fixed_leg_type: SwapTypes,
                 fixed_coupon: float,
                 fixed_freq_type: FrequencyTypes,
                 fixed_dc_type: DayCountTypes,
                 floating_index: CurveTypes,
                 currency: CurrencyTypes,
                 notional: float = ONE_MILLION,
                 payment_lag: int = 0,
                 float_spread: float = 0.0,
                 float_freq_type: FrequencyTypes = FrequencyTypes.ANNUAL,
                 float_dc_type: DayCountTypes = DayCountTypes.THIRTY_E_360,
                 cal_type: CalendarTypes = CalendarTypes.WEEKEND,
                 bd_type: BusDayAdjustTypes = BusDayAdjustTypes.FOLLOWING,
                 dg_type: DateGenRuleTypes = DateGenRuleTypes.BACKWARD):
        check_argument_types(self.__init__, locals())
        self.derivative_type = InstrumentTypes.OIS_SWAP

        if isinstance(term_dt_or_tenor, Date):
            self._termination_dt = term_dt_or_tenor
        else:
            self._termination_dt = effective_dt.add_tenor(term_dt_or_tenor)
        self._maturity_dt = Calendar(cal_type).adjust(self._termination_dt, bd_type)
        if effective_dt > self._maturity_dt:
            raise LibError("Start date after maturity date")

        self._effective_dt = effective_dt
        self._floating_index = floating_index
        self._currency = currency

        float_leg_type = SwapTypes.RECEIVE if fixed_leg_type == SwapTypes.PAY else SwapTypes.PAY
        principal = 0.0  # no exchange of par in an OIS

        self._fixed_leg = SwapFixedLeg(effective_dt, self._termination_dt, fixed_leg_type,
                                       fixed_coupon, fixed_freq_type, fixed_dc_type,
                                       floating_index, currency, notional, principal,
                                       payment_lag, cal_type, bd_type, dg_type, False)
        self._float_leg = SwapFloatLeg(effective_dt, self._termination_dt, float_leg_type,
                                       float_spread, float_freq_type, float_dc_type,
                                       floating_index, currency, notional, principal,
                                       payment_lag, cal_type, bd_type, dg_type, False, False)

        # shortcuts the curve builder reads (cavour/trades/rates/ois.py:190-194)
        self._adjusted_fixed_dts = self._fixed_leg._adjusted_fixed_dts
        self._fixed_coupon = self._fixed_leg._cpn
        self._fixed_year_fracs = self._fixed_leg._year_fracs
        self._start_dt = self._fixed_leg._effective_dt
        self._notional = notional

    def position(self, model):
        from ...market.position.position import Position
        return Position(self, model)

    def value(self, value_dt: Date, ois_curve=None, discount_curve=None, xccy_discount_curve=None,
              spot_fx: float = None, collateral_type=None, first_fixing_rate=None):
        """Swap PV off the curves' own nodes, non-AD (cavour/trades/rates/ois.py:209-273).  With no
        ``discount_curve`` and no ``collateral_type`` the OIS curve discounts (single curve); a collateral
        currency other than the swap's needs ``xccy_discount_curve`` and ``spot_fx`` and returns PV / spot_fx."""
        from ...utils.global_types import collateral_to_currency
        if discount_curve is None and collateral_type is None:
            discount_curve = ois_curve
        cross = False
        if collateral_type is not None:
            cross = collateral_to_currency(collateral_type) != self._currency
            if cross:
                if xccy_discount_curve is None or spot_fx is None:
                    raise ValueError(f"xccy_discount_curve and spot_fx required for {self._currency.name} swap "
                                     f"with {collateral_to_currency(collateral_type).name} collateral")
                discount_curve = xccy_discount_curve
            else:
                discount_curve = ois_curve
        value = (self._fixed_leg.value(value_dt, discount_curve)
                 + self._float_leg.value(value_dt, discount_curve, ois_curve, first_fixing_rate))
        if cross and spot_fx is not None:
            value = value / spot_fx
        return value

    def pv01(self, value_dt, discount_curve):
        """Value of a 1 bp coupon on the fixed leg, always positive (ois.py:277-285)."""
        pv = self._fixed_leg.value(value_dt, discount_curve)
        return abs(pv / self._fixed_leg._cpn / self._fixed_leg._notional * 100)

    def ir01(self, value_dt, discount_curve):
        """PV change per 1 bp parallel shift of the curve's zero rates (ois.py:289-300)."""
        down = self.value(value_dt, discount_curve.bump(-0.001))
        up = self.value(value_dt, discount_curve.bump(0.001))
        return (up - down) / 10 / 2

    def swap_rate(self, value_dt, ois_curve, first_fixing_rate=None):
        """Fixed coupon that makes the swap worth zero (ois.py:304-320)."""
        pv01 = self.pv01(value_dt, ois_curve)
        float_leg_value = self._float_leg.value(value_dt, ois_curve, ois_curve, first_fixing_rate)
        return float_leg_value / pv01 / self._fixed_leg._notional

    def __repr__(self):
        s = label_to_string("OBJECT TYPE", type(self).__name__)
        s += self._fixed_leg.__repr__() + "\n" + self._float_leg.__repr__()
        return s
