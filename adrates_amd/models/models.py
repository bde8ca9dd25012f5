"""Market model: named curves, their build parameters and scenario copies.

Mirrors the parts of cavour/models/models.py the OIS path uses: `CurveAccessor`
:23-49, `Model.build_curve` :142-228, `Model.build_xccy_curve` :267-391,
`Model.scenario` :507-557, `Model.curves` :559-572.  The Bloomberg-backed `prebuilt_*`
builders are outside the built scope.
"""
from dataclasses import dataclass, field
from typing import Dict, List

from ..trades.rates.ois import OIS
from ..trades.rates.ois_curve import OISCurve
from ..utils.calendar import BusDayAdjustTypes
from ..utils.currency import CurrencyTypes
from ..utils.date import Date
from ..utils.day_count import DayCountTypes
from ..utils.frequency import FrequencyTypes
from ..utils.global_types import CurveTypes, InterpTypes, SwapTypes


class CurveAccessor:
    """``model.curves.NAME`` / ``model.curves["NAME"]``."""

    def __init__(self, curves: Dict[str, OISCurve]):
        self._curves = curves

    def __getattr__(self, item):
        try:
            return self._curves[item]
        except KeyError:
            raise AttributeError(f"No such curve: {item}")

    def __getitem__(self, item):
        return self._curves[item]


@dataclass
class Model:
    value_dt: Date
    _curves_dict: Dict[str, OISCurve] = field(default_factory=dict)
    _curve_params_dict: Dict[str, dict] = field(default_factory=dict)
    _fx_params_dict: Dict[str, dict] = field(default_factory=dict)

    def build_curve(self,
                    name: str,
                    px_list: List[float],
                    tenor_list: List[str],
                    spot_days: int = 0,
                    swap_type=SwapTypes.PAY,
                    fixed_dcc_type=DayCountTypes.ACT_360,
                    fixed_freq_type=FrequencyTypes.ANNUAL,
                    float_freq_type=FrequencyTypes.ANNUAL,
                    float_dc_type=DayCountTypes.ACT_360,
                    bus_day_type=BusDayAdjustTypes.MODIFIED_FOLLOWING,
                    interp_type=InterpTypes.LINEAR_ZERO_RATES,
                    payment_lag: int = 0):
        """Register an OIS curve built from par swap quotes given in percent."""
        settle_dt = self.value_dt.add_weekdays(spot_days)
        curve_type = CurveTypes[name]
        currency = CurrencyTypes[name.split("_")[0]]
        swaps = [OIS(effective_dt=settle_dt, term_dt_or_tenor=tenor, fixed_leg_type=swap_type,
                     fixed_coupon=px / 100, fixed_freq_type=fixed_freq_type,
                     fixed_dc_type=fixed_dcc_type, floating_index=curve_type, currency=currency,
                     bd_type=bus_day_type, float_freq_type=float_freq_type,
                     float_dc_type=float_dc_type, payment_lag=payment_lag)
                 for tenor, px in zip(tenor_list, px_list)]
        self._curves_dict[name] = OISCurve(value_dt=self.value_dt, ois_swaps=swaps,
                                           interp_type=interp_type, check_refit=True)
        # payment_lag is not remembered, as in the reference (models.py:217-228)
        self._curve_params_dict[name] = {
            "tenor_list": tenor_list, "px_list": px_list, "spot_days": spot_days,
            "swap_type": swap_type, "fixed_dcc_type": fixed_dcc_type,
            "fixed_freq_type": fixed_freq_type, "float_freq_type": float_freq_type,
            "float_dc_type": float_dc_type, "bus_day_type": bus_day_type,
            "interp_type": interp_type,
        }

    def build_fx(self, currency_pairs, pxs) -> None:
        for pair, price in zip(currency_pairs, pxs):
            try:
                base, quote = CurrencyTypes[pair[:3]], CurrencyTypes[pair[3:]]
            except KeyError:
                raise ValueError(f"Invalid currency code in pair: {pair}")
            self._fx_params_dict[pair] = {"base": base, "quote": quote,
                                          "ticker": f"{pair} Curncy", "price": float(price)}

    def build_xccy_curve(self, name: str, domestic_curve_name: str, foreign_curve_name: str,
                         basis_spreads: List[float], tenor_list: List[str], spot_fx: float,
                         domestic_notional: float = 100_000_000,
                         domestic_freq_type=FrequencyTypes.ANNUAL, foreign_freq_type=FrequencyTypes.ANNUAL,
                         domestic_dc_type=DayCountTypes.ACT_360, foreign_dc_type=DayCountTypes.ACT_365F,
                         bus_day_type=BusDayAdjustTypes.MODIFIED_FOLLOWING,
                         interp_type=InterpTypes.FLAT_FWD_RATES, use_ad: bool = True):
        """Register a cross-currency curve bootstrapped from basis swaps quoted in bp on the foreign leg
        (cavour/models/models.py:267-391).  The calibration swaps exchange ``domestic_notional`` against
        ``domestic_notional / spot_fx``; the curve itself is built with ``1 / spot_fx`` (:369).  ``bus_day_type``
        is accepted and, as in the reference, not passed on to the swaps."""
        for curve in (domestic_curve_name, foreign_curve_name):
            if curve not in self._curves_dict:
                kind = "Domestic" if curve == domestic_curve_name else "Foreign"
                raise ValueError(f"{kind} curve '{curve}' not found in model. "
                                 f"Build it first using build_curve() or prebuilt_curve().")
        from ..trades.rates.xccy_basis_swap import XccyBasisSwap
        from ..trades.rates.xccy_curve import XccyCurve
        dom_ccy = CurrencyTypes[domestic_curve_name.split("_")[0]]
        for_ccy = CurrencyTypes[foreign_curve_name.split("_")[0]]
        swaps = [XccyBasisSwap(effective_dt=self.value_dt, term_dt_or_tenor=tenor,
                               domestic_notional=domestic_notional, foreign_notional=domestic_notional / spot_fx,
                               domestic_spread=0.0, foreign_spread=spread_bps / 10000.0,
                               domestic_freq_type=domestic_freq_type, foreign_freq_type=foreign_freq_type,
                               domestic_dc_type=domestic_dc_type, foreign_dc_type=foreign_dc_type,
                               domestic_floating_index=CurveTypes[domestic_curve_name],
                               foreign_floating_index=CurveTypes[foreign_curve_name],
                               domestic_currency=dom_ccy, foreign_currency=for_ccy)
                 for tenor, spread_bps in zip(tenor_list, basis_spreads)]
        self._curves_dict[name] = XccyCurve(value_dt=self.value_dt, basis_swaps=swaps,
                                            domestic_curve=self._curves_dict[domestic_curve_name],
                                            foreign_curve=self._curves_dict[foreign_curve_name],
                                            spot_fx=1 / spot_fx, interp_type=interp_type, use_ad=use_ad)
        self._curve_params_dict[name] = {
            "domestic_curve_name": domestic_curve_name, "foreign_curve_name": foreign_curve_name,
            "basis_spreads": basis_spreads, "tenor_list": tenor_list, "spot_fx": spot_fx,
            "domestic_notional": domestic_notional, "domestic_freq_type": domestic_freq_type,
            "foreign_freq_type": foreign_freq_type, "domestic_dc_type": domestic_dc_type,
            "foreign_dc_type": foreign_dc_type, "bus_day_type": bus_day_type, "interp_type": interp_type,
            "use_ad": use_ad,
        }

    def scenario(self, curve_name: str, shock, new_name=None):
        """New model whose ``curve_name`` quotes are shifted by ``shock`` (percent
        units, added to the quotes): a float shifts every pillar, a dict
        ``{tenor: shift}`` only the named ones."""
        if curve_name not in self._curve_params_dict:
            raise ValueError(f"No stored parameters found for curve '{curve_name}'")
        params = self._curve_params_dict[curve_name]
        base_px, tenors = params["px_list"], params["tenor_list"]
        if isinstance(shock, dict):
            shocked = [base_px[i] + shock.get(t, 0.0) for i, t in enumerate(tenors)]
        else:
            shocked = [px + shock for px in base_px]
        new_model = Model(value_dt=self.value_dt)
        new_model.build_curve(name=new_name or curve_name, px_list=shocked,
                              **{k: v for k, v in params.items() if k != "px_list"})
        return new_model

    @property
    def curves(self):
        return CurveAccessor(self._curves_dict)
