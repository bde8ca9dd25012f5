"""Valuation engine: dispatch + the GPU pricing call.

Mirrors the OIS branch of cavour/market/position/engine.py
(`Engine.compute` :89-124, `_compute_ois` :126-151, `_compute_ois_natural`
:153-215): fetch the index curve from the model, price fixed + float legs and
return ``AnalyticsResult(value, risk=Delta, gamma=Gamma)``.  Where the
reference runs JAX per leg on the CPU, this engine

1. builds the engine knot grid with closed-form derivative recurrences once per
   curve (market/curves/curve_tables.py) and uploads it (adr_curve_upload);
2. compiles the trade(s) to arrays (trades/compiler.py, adr_trades_upload);
3. runs the hand-written HIP kernels (adr_price).

There is no CPU pricing path: without the HIP library and a GPU ``compute``
raises `LibError`.
"""
from __future__ import annotations

from typing import Any, Dict

import numpy as np

from ... import _native
from ...requests.results import AnalyticsResult, CashflowItem, Cashflows, Delta, Gamma, Valuation
from ...trades.compiler import compile_ois
from ...utils.error import LibError
from ...utils.global_types import InstrumentTypes, InterpTypes, RequestTypes, SwapTypes, collateral_to_currency
from ...utils.helpers import to_tenor
from ..curves.curve_tables import build_engine_curve

_SUPPORTED_INTERP = (InterpTypes.FLAT_FWD_RATES.value, InterpTypes.LINEAR_FWD_RATES.value, InterpTypes.LINEAR_ZERO_RATES.value)


class Engine:
    def __init__(self, model):
        self.model = model
        # The reference keys this cache by tuple(swap_times) alone (engine.py:2362-2412, 2510), so two curves
        # with the same pillar times - a GBP and a USD curve built on one day count - collide inside one Engine
        # and a cross-currency swap would read its domestic tables for the foreign leg.  The key here also
        # carries the rates and the interpolation scheme.  The device tables themselves are cached on the
        # curve object, so Positions sharing a model do not re-bootstrap (position.py:55 creates one Engine
        # per Position).
        self._curve_cache: Dict[Any, Dict[str, Any]] = {}

    # ------------------------------------------------------------------ curves
    def _device_curve(self, ir_model):
        key = (tuple(ir_model.swap_times), tuple(ir_model.swap_rates), ir_model._interp_type.value)
        hit = self._curve_cache.get(key)
        if hit is not None:
            return hit
        # object-level cache, valid for exactly these quotes, this scheme and this context (a curve whose
        # rates or scheme were changed after the first pricing, or a second device in the process, rebuilds)
        ctx = _native.default_context()
        shared = getattr(ir_model, "_adr_device_curve", None)
        if shared is not None and (shared.get("key") != key or shared["ctx"] is not ctx):
            shared = None
        if shared is None:
            method = ir_model._interp_type.value
            if method not in _SUPPORTED_INTERP:
                raise LibError("Invalid interpolation scheme.")   # interpolator_ad.py:237
            host = build_engine_curve(ir_model.swap_rates, ir_model.swap_times, ir_model.year_fracs)
            dev = _native.DeviceCurve(ctx, method, host.times, host.dfs, host.jac, host.hess)
            shared = dict(key=key, ctx=ctx, host=host, dev=dev, tenors=to_tenor(list(ir_model.swap_times)))
            ir_model._adr_device_curve = shared
        self._curve_cache[key] = shared
        return shared

    # ---------------------------------------------------------------- dispatch
    def compute(self, derivative, request_list, collateral_type=None):
        reqs = set(request_list)
        dtype = derivative.derivative_type
        if dtype == InstrumentTypes.OIS_SWAP:
            return self._compute_ois(derivative, reqs, collateral_type)
        if dtype == InstrumentTypes.XCCY_SWAP:
            from .xccy_engine import compute_xccy
            return compute_xccy(self, derivative, reqs)
        raise LibError(f"{dtype} not yet implemented")

    def _compute_ois(self, derivative, reqs, collateral_type=None):
        collateral_ccy = (derivative._currency if collateral_type is None
                          else collateral_to_currency(collateral_type))
        if collateral_ccy == derivative._currency:
            return self._compute_ois_natural(derivative, reqs)
        from .xccy_engine import compute_ois_xccy_collateral
        return compute_ois_xccy_collateral(self, derivative, reqs, collateral_ccy)

    def _compute_ois_natural(self, derivative, reqs):
        ir_model = getattr(self.model.curves, derivative._floating_index.name)
        out = AnalyticsResult()
        if reqs & {RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA}:
            res = price_batch(self, ir_model, [derivative], reqs, per_trade=True, aggregate=False)
            out = wrap_result(res, 0, reqs, ir_model_tenors=res["tenors"], currency=derivative._currency,
                              curve_type=derivative._floating_index)
        if RequestTypes.CASHFLOWS in reqs:
            out = AnalyticsResult(value=out.value, risk=out.risk, gamma=out.gamma,
                                  cashflows=self._ois_cashflows(derivative, ir_model))
        return out

    # --------------------------------------------------------------- cash flows
    @staticmethod
    def _extract_leg_cashflows(leg, leg_type_str: str) -> list:
        """Cash-flow items of a leg that has just been valued (engine.py:34-87)."""
        if not getattr(leg, "_payment_dfs", None):
            return []
        sign = -1.0 if "Pay" in leg_type_str else 1.0
        items = []
        for i, pay_dt in enumerate(leg._payment_dts):
            notionals = getattr(leg, "_notional_array", None)
            notional = float(notionals[i]) if notionals and i < len(notionals) else float(leg._notional)
            amount = float(leg._payments[i])
            items.append(CashflowItem(payment_date=pay_dt, notional=notional,
                                      payment_fraction=amount / notional if notional != 0 else 0.0,
                                      accrual_period=float(leg._year_fracs[i]), amount=sign * amount,
                                      discount_factor=float(leg._payment_dfs[i]),
                                      discounted_amount=sign * float(leg._payment_pvs[i]),
                                      leg_type=leg_type_str))
        return items

    def _ois_cashflows(self, derivative, ir_model):
        """CASHFLOWS request (engine.py:191-213): the legs are valued off the curve's OWN node set
        (`OISCurve.df`, not the engine's knot grid) and their payment tables are reported."""
        derivative._fixed_leg.value(ir_model._value_dt, ir_model)
        derivative._float_leg.value(ir_model._value_dt, ir_model, ir_model)
        pay_fixed = derivative._fixed_leg._leg_type == SwapTypes.PAY
        items = self._extract_leg_cashflows(derivative._fixed_leg, "Fixed_Pay" if pay_fixed else "Fixed_Rec")
        items += self._extract_leg_cashflows(derivative._float_leg, "Float_Rec" if pay_fixed else "Float_Pay")
        return Cashflows(items, derivative._currency)


def price_batch(engine: Engine, ir_model, derivatives, reqs, per_trade=True, aggregate=False):
    """Compile, upload and price a list of OIS trades on ``ir_model``'s curve."""
    cur = engine._device_curve(ir_model)
    batch = compile_ois(derivatives, ir_model._value_dt)
    dev_trades = _native.DeviceTrades(cur["ctx"], batch)
    try:
        out = _native.price(cur["ctx"], cur["dev"], dev_trades,
                            want_value=RequestTypes.VALUE in reqs,
                            want_delta=RequestTypes.DELTA in reqs,
                            want_gamma=RequestTypes.GAMMA in reqs,
                            per_trade=per_trade, aggregate=aggregate)
    finally:
        dev_trades.close()
    out["tenors"] = cur["tenors"]
    return out


def wrap_result(res, idx, reqs, ir_model_tenors, currency, curve_type, aggregate=False):
    """numpy outputs -> Valuation / Delta / Gamma (engine.py:2546, 2556-2561, 2569-2574)."""
    value = delta = gamma = None
    if RequestTypes.VALUE in reqs:
        amount = res["agg_pv"] if aggregate else res["pv"][idx]
        value = Valuation(amount=float(amount), currency=currency)
    if RequestTypes.DELTA in reqs:
        ladder = res["agg_delta"] if aggregate else res["delta"][idx]
        delta = Delta(risk_ladder=np.array(ladder, dtype=np.float64), tenors=ir_model_tenors,
                      currency=currency, curve_type=curve_type)
    if RequestTypes.GAMMA in reqs:
        mat = res["agg_gamma"] if aggregate else res["gamma"][idx]
        gamma = Gamma(risk_ladder=np.array(mat, dtype=np.float64), tenors=ir_model_tenors,
                      currency=currency, curve_type=curve_type)
    return AnalyticsResult(value=value, risk=delta, gamma=gamma)
