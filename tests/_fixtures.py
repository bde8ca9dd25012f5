"""Shared market data and builders for the tests: re-exported from the package (adrates_amd/trades/market_data.py),
where the benchmark and the tools take them from as well."""
from adrates_amd.trades.market_data import *  # noqa: F401,F403
from adrates_amd.trades.market_data import (GBP_PX, README_VALUE_DT, TENORS, TEST_VALUE_DT, USD_PX, gbp_model, make_swap,  # noqa: F401
                                            readme_model, usd_model)
