"""Trade compiler: OIS objects -> struct-of-arrays batch for the GPU.

Extracts, per trade, exactly the arrays the reference's engine pulls out of the
leg objects before calling its pure pricing functions
(cavour/market/position/engine.py:2519-2527 fixed leg, :2858-2877 float leg):
payment / accrual-start / accrual-end times as year fractions from the value date
in the *leg's* day count, fixed payment amounts, float accrual fractions, spread,
notional and leg signs.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterable, Optional, Sequence, Union

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
    flt_weight: Optional[np.ndarray] = None   # per float coupon: multiplies the notional (adr_trades_upload_weighted)

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
                          self.fix_sign[lo:hi], self.flt_sign[lo:hi],
                          None if self.flt_weight is None else self.flt_weight[l0:l1])


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


# --------------------------------------------------------------------------------------------------------
# Vectorised path: trades given by their economic terms, no per-trade objects
# (SURVEY.md section 8(f) row 4).  Every Python `Date` / `Schedule` / `DayCount` call of the object path
# (cavour/utils/schedule.py:163-270, swap_fixed_leg.py:131-196, swap_float_leg.py:130-186) depends only on
# the schedule-defining terms, so one template swap is built per DISTINCT combination of those terms and
# the per-trade arrays are gathered from the templates with NumPy.  The result is bit-identical to
# `compile_ois` on the corresponding `OIS` objects.
# --------------------------------------------------------------------------------------------------------
@dataclass
class OISTerms:
    """Economic terms of n OIS trades.  Scalars broadcast; sequences must have length n.

    ``effective_dt``: a `Date`, a sequence of `Date`, or integer Excel serials (`int(Date.excel_dt())`);
    ``tenor``: tenor strings ("18M", "10Y"); the remaining fields are the `OIS` constructor arguments of
    the same names (adrates_amd/trades/rates/ois.py)."""
    effective_dt: object
    tenor: object
    coupon: object
    notional: object
    pay_fixed: object
    fixed_freq_type: object
    fixed_dc_type: object
    floating_index: object
    currency: object
    float_freq_type: object = None        # None: same as the fixed leg
    float_dc_type: object = None
    float_spread: object = 0.0
    payment_lag: object = 0
    bd_type: object = None                # None: the OIS constructor's default
    cal_type: object = None
    dg_type: object = None


def _column(value, n, kind):
    """Broadcast a scalar or validate a sequence; enums/dates become small integers for `numpy.unique`."""
    coded = kind == "code" and isinstance(value, tuple) and len(value) == 2 and isinstance(value[0], np.ndarray)
    seq = isinstance(value, (list, tuple, np.ndarray)) and not coded
    if seq and len(value) != n:
        raise LibError("OISTerms: every per-trade sequence must have one entry per trade")
    if kind == "float":
        return np.broadcast_to(np.asarray(value, dtype=np.float64), (n,)).copy() if not seq else np.asarray(value, dtype=np.float64)
    if kind == "int":
        return np.broadcast_to(np.asarray(value, dtype=np.int64), (n,)).copy() if not seq else np.asarray(value, dtype=np.int64)
    if kind == "bool":
        return np.broadcast_to(np.asarray(value, dtype=bool), (n,)).copy() if not seq else np.asarray(value, dtype=bool)
    if kind == "date":     # -> Excel serials
        if isinstance(value, np.ndarray) and value.dtype.kind in "iu":       # serials already: no per-element pass
            return value.astype(np.int64, copy=False)
        if seq:
            return np.array([v if isinstance(v, (int, np.integer)) else int(v.excel_dt()) for v in value], dtype=np.int64)
        return np.full(n, value if isinstance(value, (int, np.integer)) else int(value.excel_dt()), dtype=np.int64)
    # enums and strings: codes into a table of distinct objects (or given that way: ``(codes array, table)``)
    if isinstance(value, tuple) and len(value) == 2 and isinstance(value[0], np.ndarray) and isinstance(value[1], (list, tuple)):
        codes = np.asarray(value[0], dtype=np.int64)
        if codes.shape != (n,) or (codes.size and (codes.min() < 0 or codes.max() >= len(value[1]))):
            raise LibError("OISTerms: a coded column needs one valid code per trade")
        return codes, list(value[1])
    items = list(value) if seq else [value]
    table, codes = [], np.empty(len(items), dtype=np.int64)
    index = {}
    for i, v in enumerate(items):
        k = index.get(v)
        if k is None:
            k = index[v] = len(table)
            table.append(v)
        codes[i] = k
    return (codes if seq else np.full(n, codes[0], dtype=np.int64)), table


def unique_rows(columns):
    """``(rows, inverse)`` of the distinct rows of the integer matrix with these columns, like ``np.unique(axis=0)`` but
    through one mixed-radix int64 key when the columns' ranges allow it (they do for dates, lags and table codes): a
    1-D sort instead of a lexicographic one over structured rows - 0.1 s instead of 7 s per million rows."""
    cols = [np.asarray(c, dtype=np.int64) for c in columns]
    if not cols or cols[0].size == 0:
        return np.zeros((0, len(cols)), dtype=np.int64), np.zeros(0, dtype=np.int64)
    lo = [int(c.min()) for c in cols]
    span = [int(c.max()) - l + 1 for c, l in zip(cols, lo)]
    total = 1
    for sp in span:
        total *= sp
    if total >= 2 ** 62:
        rows, inverse = np.unique(np.stack(cols, axis=1), axis=0, return_inverse=True)
        return rows, inverse.reshape(-1)
    key = np.zeros(cols[0].shape[0], dtype=np.int64)
    for c, l, sp in zip(cols, lo, span):
        key = key * sp + (c - l)
    uniq, inverse = np.unique(key, return_inverse=True)
    rows = np.empty((uniq.shape[0], len(cols)), dtype=np.int64)
    rest = uniq.copy()
    for j in range(len(cols) - 1, -1, -1):
        rest, digit = np.divmod(rest, span[j])
        rows[:, j] = digit + lo[j]
    return rows, inverse.reshape(-1)


_FIXED_DENOMINATOR = None          # filled on first use: day counts whose year fraction is days / constant


def _fixed_denominators():
    global _FIXED_DENOMINATOR
    if _FIXED_DENOMINATOR is None:
        from ..utils.day_count import DayCountTypes
        from ..utils.global_vars import gDaysInYear
        _FIXED_DENOMINATOR = {DayCountTypes.ACT_365F: 365, DayCountTypes.ACT_360: 360, DayCountTypes.SIMPLE: gDaysInYear}
    return _FIXED_DENOMINATOR


def _legs_by_templates(cols, coded, pick, value_dt):
    """Unit-notional, unit-coupon leg arrays of trades ``pick`` through one template `OIS` per distinct schedule (the
    object path, once per distinct combination of schedule-defining terms; NumPy gathers)."""
    from ..utils.date import Date
    from .rates.ois import OIS
    names = list(coded)
    keys, inverse = unique_rows([cols["eff"][pick], cols["lag"][pick]] + [coded[k][0][pick] for k in names])
    templates = []
    for row in keys:
        kw = {k: coded[k][1][int(row[2 + j])] for j, k in enumerate(names)}
        optional = {}
        if kw["bd"] is not None:
            optional["bd_type"] = kw["bd"]
        if kw["cal"] is not None:
            optional["cal_type"] = kw["cal"]
        if kw["dg"] is not None:
            optional["dg_type"] = kw["dg"]
        swap = OIS(effective_dt=Date._from_serial(int(row[0])), term_dt_or_tenor=kw["tenor"],
                   fixed_leg_type=SwapTypes.PAY, fixed_coupon=1.0, fixed_freq_type=kw["fixed_freq"],
                   fixed_dc_type=kw["fixed_dc"], floating_index=kw["index"], currency=kw["ccy"], notional=1.0,
                   float_freq_type=kw["float_freq"], float_dc_type=kw["float_dc"], payment_lag=int(row[1]),
                   **optional)
        templates.append(compile_ois([swap], value_dt))
    n_fix = np.array([t.fix_tp.shape[0] for t in templates], dtype=np.int64)
    n_flt = np.array([t.flt_tp.shape[0] for t in templates], dtype=np.int64)

    def gather(counts, fld):
        lens = counts[inverse]
        off = np.concatenate(([0], np.cumsum(lens)))
        cat = np.concatenate([getattr(t, fld) for t in templates]) if templates else np.zeros(0)
        starts = np.concatenate(([0], np.cumsum(counts)))[:-1]
        idx = np.repeat(starts[inverse] - off[:-1], lens) + np.arange(off[-1])
        return off.astype(np.int64), cat[idx]

    fix_off, fix_tp = gather(n_fix, "fix_tp")
    flt_off, flt_tp = gather(n_flt, "flt_tp")
    return {"pick": pick, "fix_off": fix_off, "fix_tp": fix_tp, "fix_alpha": gather(n_fix, "fix_pay")[1],   # unit notional * unit coupon
            "flt_off": flt_off, "flt_tp": flt_tp, "flt_ts": gather(n_flt, "flt_ts")[1],
            "flt_te": gather(n_flt, "flt_te")[1], "flt_alpha": gather(n_flt, "flt_alpha")[1]}


def _legs_by_arrays(cols, coded, pick, value_dt, bd, weekend):
    """The same arrays with the schedules, payment lags and year fractions of all trades ``pick`` computed on arrays
    (`utils.schedule_np`; one business-day rule and calendar per call).  Returns the piece and the mask of trades (of
    ``pick``) whose two schedules are plain - the others have to go through `_legs_by_templates`."""
    from ..utils import schedule_np as S
    from ..utils.calendar import Calendar, CalendarTypes
    from ..utils.frequency import annual_frequency
    den_of = _fixed_denominators()
    eff = cols["eff"][pick]
    count, unit = S.parse_tenors(coded["tenor"][1])
    code = coded["tenor"][0][pick]
    term = S.add_tenor(eff, count[code], unit[code])
    if (eff > S.adjust(term, bd, weekend)).any():
        raise LibError("Start date after maturity date")
    value_serial = int(value_dt.excel_dt())
    den = lambda key: np.array([den_of.get(d, 0) for d in coded[key][1]])[coded[key][0][pick]]
    mpp = lambda key: np.array([int(12 / annual_frequency(f)) for f in coded[key][1]], dtype=np.int64)[coded[key][0][pick]]
    lag = cols["lag"][pick]
    fix_off, fix_tp, _, _, fix_alpha, fix_plain = S.leg_times(eff, term, mpp("fixed_freq"), lag, bd, weekend, den("fixed_dc"), value_serial)
    flt_off, flt_tp, flt_ts, flt_te, flt_alpha, flt_plain = S.leg_times(eff, term, mpp("float_freq"), lag, bd, weekend,
                                                                       den("float_dc"), value_serial)
    piece = {"pick": pick, "fix_off": fix_off, "fix_tp": fix_tp, "fix_alpha": fix_alpha, "flt_off": flt_off, "flt_tp": flt_tp,
             "flt_ts": flt_ts, "flt_te": flt_te, "flt_alpha": flt_alpha}
    return piece, fix_plain & flt_plain


def _take_piece(piece, keep):
    """Trades ``keep`` (positions inside the piece) of a piece."""
    out = {"pick": piece["pick"][keep]}
    for side, fields in (("fix", ("fix_tp", "fix_alpha")), ("flt", ("flt_tp", "flt_ts", "flt_te", "flt_alpha"))):
        src = piece[side + "_off"]
        lens = (src[1:] - src[:-1])[keep]
        off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        idx = np.repeat(src[:-1][keep] - off[:-1], lens) + np.arange(off[-1])
        out[side + "_off"] = off
        for f in fields:
            out[f] = piece[f][idx]
    return out


def _merge_pieces(pieces, n):
    """Pieces (disjoint sets of trades, any order) -> CSR arrays in trade order."""
    out = {}
    for side, fields in (("fix", ("fix_tp", "fix_alpha")), ("flt", ("flt_tp", "flt_ts", "flt_te", "flt_alpha"))):
        lens = np.zeros(n, dtype=np.int64)
        for p in pieces:
            lens[p["pick"]] = p[side + "_off"][1:] - p[side + "_off"][:-1]
        off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        cols = {f: np.empty(off[-1], dtype=np.float64) for f in fields}
        for p in pieces:
            pl = p[side + "_off"][1:] - p[side + "_off"][:-1]
            dst = np.repeat(off[:-1][p["pick"]] - p[side + "_off"][:-1], pl) + np.arange(p[side + "_off"][-1])
            for f in fields:
                cols[f][dst] = p[f]
        out[side + "_off"], out[side + "_len"] = off, lens
        out.update(cols)
    return out


def compile_ois_terms(terms: OISTerms, value_dt) -> TradeBatch:
    """`TradeBatch` for the trades described by ``terms`` as of ``value_dt``.

    Schedules, payment lags and year fractions are computed on arrays for the whole book (`utils.schedule_np`:
    BACKWARD date generation, WEEKEND / NONE calendar, any business-day rule, day counts with a fixed denominator) -
    bit for bit what the `OIS` objects produce; trades on other conventions, or whose schedule hits the reference's
    front-dropping de-duplication (schedule.py:256-266), go through one template object per distinct schedule."""
    from ..utils.calendar import BusDayAdjustTypes, CalendarTypes, DateGenRuleTypes

    notional = np.asarray(terms.notional, dtype=np.float64).reshape(-1)
    n = notional.shape[0]
    coupon = _column(terms.coupon, n, "float")
    spread = _column(terms.float_spread, n, "float")
    pay_fixed = _column(terms.pay_fixed, n, "bool")
    cols = {"eff": _column(terms.effective_dt, n, "date"), "lag": _column(terms.payment_lag, n, "int")}
    float_freq = terms.fixed_freq_type if terms.float_freq_type is None else terms.float_freq_type
    float_dc = terms.fixed_dc_type if terms.float_dc_type is None else terms.float_dc_type
    coded = {}
    for name, value in (("tenor", terms.tenor), ("fixed_freq", terms.fixed_freq_type), ("float_freq", float_freq),
                        ("fixed_dc", terms.fixed_dc_type), ("float_dc", float_dc), ("bd", terms.bd_type),
                        ("cal", terms.cal_type), ("dg", terms.dg_type), ("index", terms.floating_index),
                        ("ccy", terms.currency)):
        coded[name] = _column(value, n, "code")

    # distinct schedules first: books repeat them (the benchmark portfolio has 360 among a million trades), and both
    # routes below work per schedule; the per-trade arrays are gathered at the end
    names = list(coded)
    keys, inverse = unique_rows([cols["eff"], cols["lag"]] + [coded[k][0] for k in names])
    n_trades, n = n, keys.shape[0]
    cols = {"eff": keys[:, 0].copy(), "lag": keys[:, 1].copy()}
    coded = {k: (keys[:, 2 + j].copy(), coded[k][1]) for j, k in enumerate(names)}

    # which schedules the arrays can do: by table entry, then per schedule
    den_of = _fixed_denominators()
    table_ok = lambda key, ok: np.array([ok(v) for v in coded[key][1]], dtype=bool)[coded[key][0]]
    by_arrays = (table_ok("fixed_dc", lambda d: d in den_of) & table_ok("float_dc", lambda d: d in den_of) &
                 table_ok("dg", lambda g: g is None or g == DateGenRuleTypes.BACKWARD) &
                 table_ok("cal", lambda c: c is None or c in (CalendarTypes.WEEKEND, CalendarTypes.NONE)))
    pieces = []
    if by_arrays.any():
        # one call per (business-day rule, calendar) present - normally one
        combo = coded["bd"][0] * len(coded["cal"][1]) + coded["cal"][0]
        for c in np.unique(combo[by_arrays]):
            pick = np.nonzero(by_arrays & (combo == c))[0]
            bd = coded["bd"][1][int(c) // len(coded["cal"][1])]
            cal = coded["cal"][1][int(c) % len(coded["cal"][1])]
            piece, plain = _legs_by_arrays(cols, coded, pick, value_dt, BusDayAdjustTypes.FOLLOWING if bd is None else bd,
                                           cal is None or cal == CalendarTypes.WEEKEND)
            if not plain.all():
                by_arrays[pick[~plain]] = False
                piece = _take_piece(piece, np.nonzero(plain)[0])
            pieces.append(piece)
    rest = np.nonzero(~by_arrays)[0]
    if rest.size:
        pieces.append(_legs_by_templates(cols, coded, rest, value_dt))
    m = _merge_pieces(pieces, n)
    if n != n_trades or not np.array_equal(inverse, np.arange(n_trades)):
        m = _take_piece(dict(m, pick=np.arange(n)), inverse)          # schedule -> trades
        m["fix_len"] = m["fix_off"][1:] - m["fix_off"][:-1]
    # payment = year_frac * notional * coupon, in the leg's evaluation order (swap_fixed_leg.py:190)
    fix_pay = m["fix_alpha"] * np.repeat(notional, m["fix_len"]) * np.repeat(coupon, m["fix_len"])
    sign_fix = np.where(pay_fixed, -1.0, 1.0)
    return TradeBatch(m["fix_off"], m["flt_off"], m["fix_tp"], fix_pay, m["flt_tp"], m["flt_ts"], m["flt_te"], m["flt_alpha"],
                      notional.copy(), spread, sign_fix, -sign_fix)
