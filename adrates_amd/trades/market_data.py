"""Market inputs and builders shared by the benchmark, the tools, the examples and the tests (inputs only; no
expected values).

The two 32-pillar quote sets are the market inputs the reference's own tests and README use
(tests/test_ois_request_types.py:36-51, 88-103; README.md:69-78)."""

from ..models.models import Model
from .rates.ois import OIS
from ..utils import (BusDayAdjustTypes, CurrencyTypes, CurveTypes, Date, DayCountTypes, FrequencyTypes, InterpTypes,
                     SwapTypes)

GBP_PX = [5.1998, 5.2014, 5.2003, 5.2027, 5.2023, 5.19281,
          5.1656, 5.1482, 5.1342, 5.1173, 5.1013, 5.0862,
          5.0701, 5.054, 5.0394, 4.8707, 4.75483, 4.532,
          4.3628, 4.2428, 4.16225, 4.1132, 4.08505, 4.0762,
          4.078, 4.0961, 4.12195, 4.1315, 4.113, 4.07724, 3.984, 3.88]
USD_PX = [5.3500, 5.3200, 5.3100, 5.2900, 5.2700, 5.2500,
          5.2300, 5.2100, 5.1900, 5.1700, 5.1500, 5.1300,
          5.1100, 5.0900, 5.0700, 4.9500, 4.8500, 4.7000,
          4.5800, 4.4800, 4.4100, 4.3600, 4.3200, 4.2900,
          4.2700, 4.2800, 4.3000, 4.3200, 4.3100, 4.2900, 4.2400, 4.1800]
TENORS = ["1D", "1W", "2W", "1M", "2M", "3M", "4M", "5M", "6M",
          "7M", "8M", "9M", "10M", "11M", "1Y", "18M", "2Y",
          "3Y", "4Y", "5Y", "6Y", "7Y", "8Y", "9Y", "10Y",
          "12Y", "15Y", "20Y", "25Y", "30Y", "40Y", "50Y"]

README_VALUE_DT = Date(30, 4, 2024)
TEST_VALUE_DT = Date(17, 12, 2024)


def gbp_model(value_dt=README_VALUE_DT, interp=InterpTypes.LINEAR_ZERO_RATES, px=None, tenors=None,
              freq=FrequencyTypes.ANNUAL):
    m = Model(value_dt)
    m.build_curve(name="GBP_OIS_SONIA", px_list=list(px or GBP_PX), tenor_list=list(tenors or TENORS),
                  spot_days=0, swap_type=SwapTypes.PAY, fixed_dcc_type=DayCountTypes.ACT_365F,
                  fixed_freq_type=freq, float_freq_type=freq, float_dc_type=DayCountTypes.ACT_365F,
                  bus_day_type=BusDayAdjustTypes.MODIFIED_FOLLOWING, interp_type=interp)
    return m


def readme_model():
    return gbp_model()


def usd_model(value_dt=TEST_VALUE_DT, interp=InterpTypes.LINEAR_ZERO_RATES):
    m = Model(value_dt)
    m.build_curve(name="USD_OIS_SOFR", px_list=list(USD_PX), tenor_list=list(TENORS), spot_days=0,
                  swap_type=SwapTypes.PAY, fixed_dcc_type=DayCountTypes.ACT_360,
                  fixed_freq_type=FrequencyTypes.ANNUAL, float_freq_type=FrequencyTypes.ANNUAL,
                  float_dc_type=DayCountTypes.ACT_360, bus_day_type=BusDayAdjustTypes.MODIFIED_FOLLOWING,
                  interp_type=interp)
    return m


def make_swap(effective_dt, tenor, coupon, notional=1e6, pay=True, dc=DayCountTypes.ACT_365F,
              index=CurveTypes.GBP_OIS_SONIA, ccy=CurrencyTypes.GBP, fixed_freq=FrequencyTypes.ANNUAL,
              float_freq=FrequencyTypes.ANNUAL, float_dc=None, payment_lag=0, spread=0.0):
    return OIS(effective_dt=effective_dt, term_dt_or_tenor=tenor,
               fixed_leg_type=SwapTypes.PAY if pay else SwapTypes.RECEIVE, fixed_coupon=coupon,
               fixed_freq_type=fixed_freq, fixed_dc_type=dc, floating_index=index, currency=ccy,
               notional=notional, payment_lag=payment_lag, float_spread=spread,
               bd_type=BusDayAdjustTypes.MODIFIED_FOLLOWING, float_freq_type=float_freq,
               float_dc_type=float_dc or dc)
