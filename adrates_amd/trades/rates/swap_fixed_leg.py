"""Fixed leg of a swap: schedule, accrual fractions and coupon amounts.

Host-side input producer for the valuation kernels (SURVEY.md section 8(a) row
D).  Attribute names are the ones the reference's engine reads
(cavour/market/position/engine.py:2519-2527): ``_payment_dts``, ``_payments``,
``_year_fracs``, ``_principal``, ``_notional``, ``_leg_type``, ``_dc_type``,
``_currency``, ``_floating_index``.  Construction follows
cavour/trades/rates/swap_fixed_leg.py:63-196.
"""
from ...utils.calendar import BusDayAdjustTypes, Calendar, CalendarTypes, DateGenRuleTypes
from ...utils.currency import CurrencyTypes
from ...utils.date import Date
from ...utils.day_count import DayCount, DayCountTypes
from ...utils.error import LibError
from ...utils.frequency import FrequencyTypes
from ...utils.global_types import CurveTypes, InstrumentTypes, SwapTypes
from ...utils.global_vars import ONE_MILLION
from ...utils.helpers import check_argument_types, label_to_string
from ...utils.schedule import Schedule


class SwapFixedLeg:
    def __init__(self,
                 effective_dt: Date,
                 end_dt: (Date, str),
                 leg_type: SwapTypes,
                 coupon: (float),
                 freq_type: FrequencyTypes,
                 dc_type: DayCountTypes,
                 floating_index: CurveTypes,
                 currency: CurrencyTypes,
                 notional: float = ONE_MILLION,
                 principal: float = 0.0,
                 payment_lag: int = 0,
                 cal_type: CalendarTypes = CalendarTypes.WEEKEND,
                 bd_type: BusDayAdjustTypes = BusDayAdjustTypes.FOLLOWING,
                 dg_type: DateGenRuleTypes = DateGenRuleTypes.BACKWARD,
                 end_of_month: bool = False):
        self.intrument_type = InstrumentTypes.SWAP_FIXED_LEG  # (sic) reference spelling
        check_argument_types(self.__init__, locals())

        self._termination_dt = end_dt if type(end_dt) == Date else effective_dt.add_tenor(end_dt)
        self._maturity_dt = Calendar(cal_type).adjust(self._termination_dt, bd_type)
        if effective_dt > self._maturity_dt:
            raise LibError("Effective date after maturity date")

        self._effective_dt = effective_dt
        self._end_dt = end_dt
        self._leg_type = leg_type
        self._freq_type = freq_type
        self._payment_lag = payment_lag
        self._notional = notional
        self._principal = principal
        self._cpn = coupon
        self._floating_index = floating_index
        self._currency = currency
        self._dc_type = dc_type
        self._cal_type = cal_type
        self._bd_type = bd_type
        self._dg_type = dg_type
        self._end_of_month = end_of_month
        self.generate_payments()

    def generate_payments(self):
        """Coupon dates, accrual fractions and amounts for the whole life of
        the leg: ``payment_j = year_frac_j * notional * coupon``
        (cavour/trades/rates/swap_fixed_leg.py:131-196)."""
        dts = Schedule(self._effective_dt, self._termination_dt, self._freq_type,
                       self._cal_type, self._bd_type, self._dg_type,
                       end_of_month=self._end_of_month)._adjusted_dts
        if len(dts) < 2:
            raise LibError("Schedule has none or only one date")

        counter = DayCount(self._dc_type)
        calendar = Calendar(self._cal_type)

        self._start_accrued_dts = list(dts[:-1])
        self._end_accrued_dts = list(dts[1:])
        self._payment_dts = []
        self._payment_dts_ad = []
        self._payments = []
        self._year_fracs = []
        self._accrued_days = []
        self._rates = []
        for start, end in zip(dts[:-1], dts[1:]):
            pay = end if self._payment_lag == 0 else calendar.add_business_days(end, self._payment_lag)
            self._payment_dts.append(pay)
            self._payment_dts_ad.append(counter.year_frac(self._effective_dt, end)[0])
            alpha, days, _ = counter.year_frac(start, end)
            self._year_fracs.append(alpha)
            self._accrued_days.append(days)
            self._rates.append(self._cpn)
            self._payments.append(alpha * self._notional * self._cpn)
        self._adjusted_fixed_dts = list(self._payment_dts)

    def value(self, value_dt: Date, discount_curve):
        """Leg PV off a curve's own nodes (`curve.df`), non-AD (cavour/trades/rates/swap_fixed_leg.py:200-246):
        payments strictly after ``value_dt``, each discounted with ``df(payment) / df(value_dt)`` on the leg's
        day count; the per-payment tables of the last valuation are kept for inspection."""
        self._payment_dfs, self._payment_pvs, self._cumulative_pvs = [], [], []
        df_value = discount_curve.df(value_dt, self._dc_type)
        leg_pv, df_pmnt, pmnt_dt = 0.0, 0.0, None
        for pmnt_dt, amount in zip(self._payment_dts, self._payments):
            if pmnt_dt > value_dt:
                df_pmnt = discount_curve.df(pmnt_dt, self._dc_type) / df_value
                leg_pv += amount * df_pmnt
                self._payment_dfs.append(df_pmnt)
                self._payment_pvs.append(amount * df_pmnt)
                self._cumulative_pvs.append(leg_pv)
            else:
                self._payment_dfs.append(0.0)
                self._payment_pvs.append(0.0)
                self._cumulative_pvs.append(0.0)
        if pmnt_dt is not None and pmnt_dt > value_dt:
            principal_pv = self._principal * df_pmnt * self._notional
            self._payment_pvs[-1] += principal_pv
            leg_pv += principal_pv
            self._cumulative_pvs[-1] = leg_pv
        return -leg_pv if self._leg_type == SwapTypes.PAY else leg_pv

    def __repr__(self):
        s = label_to_string("OBJECT TYPE", type(self).__name__)
        s += label_to_string("START DATE", self._effective_dt)
        s += label_to_string("TERMINATION DATE", self._termination_dt)
        s += label_to_string("MATURITY DATE", self._maturity_dt)
        s += label_to_string("NOTIONAL", self._notional)
        s += label_to_string("PRINCIPAL", self._principal)
        s += label_to_string("LEG TYPE", self._leg_type)
        s += label_to_string("COUPON", self._cpn)
        s += label_to_string("FREQUENCY", self._freq_type)
        s += label_to_string("DAY COUNT", self._dc_type)
        s += label_to_string("CALENDAR", self._cal_type)
        s += label_to_string("BUS DAY ADJUST", self._bd_type)
        s += label_to_string("DATE GEN TYPE", self._dg_type, "")
        return s
