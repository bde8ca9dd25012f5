"""Floating leg of a swap: schedule and accrual fractions.

Host-side input producer for the valuation kernels (SURVEY.md section 8(a) row
D); the engine reads ``_payment_dts``, ``_start_accrued_dts``,
``_end_accrued_dts``, ``_year_fracs``, ``_spread``, ``_notional``,
``_notional_array``, ``_principal`` (cavour/market/position/engine.py:2858-2877).
Construction follows cavour/trades/rates/swap_float_leg.py:65-186; note that the
``principal`` argument is accepted but the stored principal is always 0
(:106).
"""
from ...utils.calendar import BusDayAdjustTypes, Calendar, CalendarTypes, DateGenRuleTypes
from ...utils.currency import CurrencyTypes
from ...utils.date import Date
from ...utils.day_count import DayCount, DayCountTypes
from ...utils.error import LibError
from ...utils.frequency import FrequencyTypes
from ...utils.global_types import CurveTypes, SwapTypes
from ...utils.global_vars import ONE_MILLION
from ...utils.helpers import check_argument_types, label_to_string
from ...utils.schedule import Schedule


class SwapFloatLeg:
    def __init__(self,
                 effective_dt: Date,
                 end_dt: (Date, str),
                 leg_type: SwapTypes,
                 spread: (float),
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
                 end_of_month: bool = False,
                 notional_exchange: bool = False):
        check_argument_types(self.__init__, locals())

        self._termination_dt = end_dt if type(end_dt) == Date else effective_dt.add_tenor(end_dt)
        self._maturity_dt = Calendar(cal_type).adjust(self._termination_dt, bd_type)
        if effective_dt > self._maturity_dt:
            raise LibError("Start date after maturity date")

        self._effective_dt = effective_dt
        self._end_dt = end_dt
        self._leg_type = leg_type
        self._freq_type = freq_type
        self._payment_lag = payment_lag
        self._principal = 0.0
        self._notional = notional
        self._notional_array = []
        self._spread = spread
        self._floating_index = floating_index
        self._currency = currency
        self._notional_exchange = notional_exchange
        self._dc_type = dc_type
        self._cal_type = cal_type
        self._bd_type = bd_type
        self._dg_type = dg_type
        self._end_of_month = end_of_month
        self._payments = []
        self.generate_payment_dts()

    def generate_payment_dts(self):
        """Accrual start/end dates, payment dates and accrual fractions
        (cavour/trades/rates/swap_float_leg.py:130-186)."""
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
        self._payment_dts_float = []
        self._payment_dts_ad = []
        self._year_fracs = []
        self._accrued_days = []
        running = 0
        for start, end in zip(dts[:-1], dts[1:]):
            pay = end if self._payment_lag == 0 else calendar.add_business_days(end, self._payment_lag)
            self._payment_dts.append(pay)
            self._payment_dts_ad.append(counter.year_frac(self._effective_dt, end)[0])
            alpha, days, _ = counter.year_frac(start, end)
            running += alpha
            self._payment_dts_float.append(running)
            self._year_fracs.append(alpha)
            self._accrued_days.append(days)

    def value(self, value_dt: Date, discount_curve, index_curve=None, first_fixing_rate: float = None):
        """Leg PV off the curves' own nodes, non-AD (cavour/trades/rates/swap_float_leg.py:190-352): forward
        rates from ``index_curve.df`` at the accrual dates over the INDEX curve's day-count fraction, coupons
        on the leg's accrual fraction, discounting as for the fixed leg; optional notional exchange
        (-N at the effective date, +N at maturity).  Unlike the reference, the exchange does not mutate the
        leg's schedule arrays (the reference inserts a zero-accrual flow into `_payment_dts` on first use)."""
        if discount_curve is None:
            raise LibError("Discount curve is None")
        if index_curve is None:
            index_curve = discount_curve
        self._rates, self._payments, self._payment_dfs, self._payment_pvs, self._cumulative_pvs = [], [], [], [], []
        df_value = discount_curve.df(value_dt, self._dc_type)
        n = len(self._payment_dts)
        notionals = list(self._notional_array) if len(self._notional_array) else [self._notional] * n
        if len(notionals) < n:
            notionals = [self._notional] * (n - len(notionals)) + notionals
        notionals = notionals[:n]
        index_counter = DayCount(index_curve._dc_type)
        leg_pv, df_pmnt, pmnt_dt, first_done = 0.0, 0.0, None, False
        for i, pmnt_dt in enumerate(self._payment_dts):
            if pmnt_dt > value_dt:
                start, end = self._start_accrued_dts[i], self._end_accrued_dts[i]
                index_alpha = index_counter.year_frac(start, end)[0]
                if not first_done and first_fixing_rate is not None:
                    fwd_rate, first_done = first_fixing_rate, True
                else:
                    fwd_rate = (index_curve.df(start, self._dc_type) / index_curve.df(end, self._dc_type) - 1.0) / index_alpha
                amount = (fwd_rate + self._spread) * self._year_fracs[i] * notionals[i]
                df_pmnt = discount_curve.df(pmnt_dt, self._dc_type) / df_value
                leg_pv += amount * df_pmnt
                self._rates.append(fwd_rate)
                self._payments.append(amount)
                self._payment_dfs.append(df_pmnt)
                self._payment_pvs.append(amount * df_pmnt)
            else:
                self._rates.append(0.0)
                self._payments.append(0.0)
                self._payment_dfs.append(0.0)
                self._payment_pvs.append(0.0)
            self._cumulative_pvs.append(leg_pv)
        if pmnt_dt is not None and pmnt_dt > value_dt:
            principal_pv = self._principal * df_pmnt * notionals[-1]
            self._payment_pvs[-1] += principal_pv
            leg_pv += principal_pv
            self._cumulative_pvs[-1] = leg_pv
        if self._notional_exchange:
            if self._effective_dt >= value_dt:
                leg_pv += float(-self._notional * (discount_curve.df(self._effective_dt, self._dc_type) / df_value))
            if self._maturity_dt >= value_dt and len(self._payments) > 0:
                leg_pv += float(self._notional * (discount_curve.df(self._maturity_dt, self._dc_type) / df_value))
        return -leg_pv if self._leg_type == SwapTypes.PAY else leg_pv

    def __repr__(self):
        s = label_to_string("OBJECT TYPE", type(self).__name__)
        s += label_to_string("START DATE", self._effective_dt)
        s += label_to_string("TERMINATION DATE", self._termination_dt)
        s += label_to_string("MATURITY DATE", self._maturity_dt)
        s += label_to_string("NOTIONAL", self._notional)
        s += label_to_string("LEG TYPE", self._leg_type)
        s += label_to_string("SPREAD", self._spread)
        s += label_to_string("FREQUENCY", self._freq_type)
        s += label_to_string("DAY COUNT", self._dc_type)
        s += label_to_string("CALENDAR", self._cal_type)
        s += label_to_string("BUS DAY ADJUST", self._bd_type)
        s += label_to_string("DATE GEN TYPE", self._dg_type, "")
        return s
