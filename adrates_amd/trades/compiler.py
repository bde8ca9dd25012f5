"""Trade compiler: OIS objects -> struct-of-arrays batch for the GPU.

Extracts, per trade, exactly the arrays the reference's engine pulls out of the
leg objects before calling its pure pricing functions
(cavour/market/position/engine.py:2519-2527 fixed leg, :2858-2877 float leg):
payment / accrual-start / accrual-end times as year fractions from the value date
in the *leg's* day count, fixed payment amounts, float accrual fractions, spread,
notional and leg signs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable

import numpy as np

from ..utils.day_count import DayCount
from ..utils.error import LibError
from ..utils.global_types import InstrumentTypes, SwapTypes


@dataclass
class TradeBatch:
    """CSR batch of OIS trades (argument list of adr_trades_upload)."""
    fix_off: np.ndarray   # [n+1] int64
    flt_off: np.ndarray   # [n+1] int64
    fix_tp: np.ndarray
    fix_pay: np.ndarray
    flt_tp: np.ndarray
    flt_ts: np.ndarray
    flt_te: np.ndarray
    flt_alpha: np.ndarray
    notional: np.ndarray  # [n]
    spread: np.ndarray
    fix_sign: np.ndarray
    flt_sign: np.ndarray

    @property
    def n_trades(self) -> int:
        return int(self.notional.shape[0])

    def slice(self, lo: int, hi: int) -> "TradeBatch":
        """Trades lo..hi-1 as an independent batch (offsets rebased)."""
        f0, f1 = int(self.fix_off[lo]), int(self.fix_off[hi])
        l0, l1 = int(self.flt_off[lo]), int(self.flt_off[hi])
        return TradeBatch(self.fix_off[lo:hi + 1] - f0, self.flt_off[lo:hi + 1] - l0,
                          self.fix_tp[f0:f1], self.fix_pay[f0:f1], self.flt_tp[l0:l1], self.flt_ts[l0:l1],
                          self.flt_te[l0:l1], self.flt_alpha[l0:l1], self.notional[lo:hi], self.spread[lo:hi],
                          self.fix_sign[lo:hi], self.flt_sign[lo:hi])


def _times(dts, value_dt, dc_type):
    counter = DayCount(dc_type)
    return [counter.year_frac(value_dt, d)[0] for d in dts]


def compile_ois(swaps: Iterable, value_dt) -> TradeBatch:
    fix_off, flt_off = [0], [0]
    fix_tp, fix_pay, flt_tp, flt_ts, flt_te, flt_al = [], [], [], [], [], []
    notional, spread, fix_sign, flt_sign = [], [], [], []
    for s in swaps:
        if getattr(s, "derivative_type", None) != InstrumentTypes.OIS_SWAP:
            raise LibError(f"{getattr(s, 'derivative_type', type(s))} not yet implemented")
        fl, xl = s._fixed_leg, s._float_leg
        if fl._principal != 0.0 or xl._principal != 0.0 or xl._notional_array:
            raise LibError("OIS legs with principal or notional schedules are outside the built path")
        fix_tp += _times(fl._payment_dts, value_dt, fl._dc_type)
        fix_pay += list(fl._payments)
        flt_tp += _times(xl._payment_dts, value_dt, xl._dc_type)
        flt_ts += _times(xl._start_accrued_dts, value_dt, xl._dc_type)
        flt_te += _times(xl._end_accrued_dts, value_dt, xl._dc_type)
        flt_al += list(xl._year_fracs)
        fix_off.append(len(fix_tp))
        flt_off.append(len(flt_tp))
        notional.append(xl._notional)
        spread.append(xl._spread)
        fix_sign.append(+1.0 if fl._leg_type == SwapTypes.RECEIVE else -1.0)
        flt_sign.append(+1.0 if xl._leg_type == SwapTypes.RECEIVE else -1.0)
    f64 = lambda a: np.array(a, dtype=np.float64)
    return TradeBatch(np.array(fix_off, dtype=np.int64), np.array(flt_off, dtype=np.int64),
                      f64(fix_tp), f64(fix_pay), f64(flt_tp), f64(flt_ts), f64(flt_te), f64(flt_al),
                      f64(notional), f64(spread), f64(fix_sign), f64(flt_sign))
