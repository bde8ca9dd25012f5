"""Cross-currency basis swaps through the OIS kernels.

Mirrors `Engine._compute_xccy` (cavour/market/position/engine.py:1411-1988) for `XccyBasisSwap`: PV in domestic
currency, three delta ladders and three gamma matrices - domestic OIS pillars, foreign OIS pillars, basis
pillars.  The reference differentiates two `_float_leg_jax` calls (domestic leg on its own curve; foreign leg
discounted on the XCCY curve with forwards off the foreign OIS curve) w.r.t. three knot-DF vectors and chains
them with the curves' Jacobians / Hessians.  Every piece is a sum of terms ``c * exp(sum beta * ln d_k)`` over
ONE curve's knots once the other curve is held fixed - which is exactly how the reference defines its ladders
(the XCCY curve is fixed when the foreign OIS rates move, :1702-1712) - so the existing kernels do all of it:

1. domestic leg: a float leg plus two "fixed" flows (-N at the effective time, +N at maturity) on the
   domestic curve's tables;
2. foreign-rate ladders: the foreign coupons on the foreign OIS tables with notional ``N * D_x(tp_j)``, paid at
   time 0 (where D = 1), accrual from ``ts_j`` to ``te_j``: the Greeks of ``sum_j N D_x(tp_j) D_f(ts_j) / D_f(te_j)``;
3. foreign PV and basis ladders: fixed flows ``N (fwd_j + s) alpha_j`` at ``tp_j`` plus the two exchanges, on
   tables uploaded from the XCCY curve's ``(_times, _dfs, _jac_basis, _hess_basis)``.

Piece 2 is one trade per swap whose coupons carry a per-coupon notional multiplier
(`adr_trades_upload_weighted`), so a book of swaps costs three launches.

Flows dated exactly at the value time follow the reference's masks: coupons and exchanges count (``>=``), and
since their discount factor is 1 they are added to the PV on the host (the kernels' fixed-flow mask is ``>``).

Cross-gamma foreign OIS x basis (`cross_gamma_for_basis`): the mixed second derivative of the PV w.r.t. the foreign
par rates and the basis spreads with the two curves' knot discount factors as the channels,
``(sign / spot) sum_j grad_r[N D_f(ts_j) / D_f(te_j)] (x) grad_s[D_x(tp_j)]`` - per coupon an outer product of two
ladders the kernels already produce (each coupon priced as a trade of its own on the foreign tables and on the XCCY
tables).  This is NOT the reference's block (:1895-1960), which keeps only the term that comes from the XCCY
bootstrap's own dependence on the foreign curve and contracts a tensor indexed by the foreign curve's OWN nodes with
the Jacobian of the engine's knot grid (two different sizes whenever the curve has more than annual points); see
DESIGN.md section 9.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass

import numpy as np

from ... import _native
from ...requests.results import AnalyticsResult, CrossGamma, Delta, Gamma, Risk, Valuation
from ...trades.compiler import TradeBatch
from ...utils.day_count import DayCountTypes
from ...utils.error import LibError
from ...utils.global_types import CurveTypes, InterpTypes, RequestTypes, SwapTypes
from ...utils.global_vars import gDaysInYear
from ...utils.helpers import times_from_dates, to_tenor


def _times(dts, value_dt, dc):
    return np.array([times_from_dates(d, value_dt, dc) for d in dts], dtype=np.float64)


def _sign(leg):
    return 1.0 if leg._leg_type == SwapTypes.RECEIVE else -1.0


@dataclass
class RawXccy:
    """Times and per-swap terms of a book of basis swaps, before any curve is consulted (CSR over the coupons).
    Domestic leg in its own day count; foreign payment / exchange times in the XCCY curve's day count, foreign
    accrual times in the foreign leg's (engine.py:1500-1506, 1519-1520)."""
    dom_off: np.ndarray; dom_tp: np.ndarray; dom_ts: np.ndarray; dom_te: np.ndarray; dom_al: np.ndarray
    for_off: np.ndarray; for_tpx: np.ndarray; for_ts: np.ndarray; for_te: np.ndarray; for_al: np.ndarray
    dom_exch_t: np.ndarray      # [n, 2] effective / maturity time, domestic day count
    for_exch_t: np.ndarray      # [n, 2] the same in the XCCY curve's day count
    dom_exch: np.ndarray; for_exch: np.ndarray          # [n] bool: the leg exchanges notional
    dom_n: np.ndarray; for_n: np.ndarray; dom_spread: np.ndarray; for_spread: np.ndarray
    dom_sign: np.ndarray; for_sign: np.ndarray

    @property
    def n(self):
        return int(self.dom_n.shape[0])


def raw_from_swaps(swaps, value_dt, xdc) -> RawXccy:
    """`RawXccy` of a list of `XccyBasisSwap` objects (one Python pass over the objects)."""
    cols = {k: [] for k in ("dtp", "dts", "dte", "dal", "ftp", "fts", "fte", "fal")}
    d_off, f_off = [0], [0]
    d_ex, f_ex = [], []
    for swap in swaps:
        dl, fl = swap._domestic_leg, swap._foreign_leg
        ddc, fdc = dl._dc_type, fl._dc_type
        cols["dtp"].append(_times(dl._payment_dts, value_dt, ddc)); cols["dts"].append(_times(dl._start_accrued_dts, value_dt, ddc))
        cols["dte"].append(_times(dl._end_accrued_dts, value_dt, ddc)); cols["dal"].append(np.asarray(dl._year_fracs, dtype=np.float64))
        cols["ftp"].append(_times(fl._payment_dts, value_dt, xdc)); cols["fts"].append(_times(fl._start_accrued_dts, value_dt, fdc))
        cols["fte"].append(_times(fl._end_accrued_dts, value_dt, fdc)); cols["fal"].append(np.asarray(fl._year_fracs, dtype=np.float64))
        d_off.append(d_off[-1] + cols["dtp"][-1].size)
        f_off.append(f_off[-1] + cols["ftp"][-1].size)
        d_ex.append([times_from_dates(swap._effective_dt, value_dt, ddc), times_from_dates(swap._maturity_dt, value_dt, ddc)])
        f_ex.append([times_from_dates(swap._effective_dt, value_dt, xdc), times_from_dates(swap._maturity_dt, value_dt, xdc)])
    cat = lambda k: np.concatenate(cols[k]) if cols[k] else np.zeros(0)
    f64 = lambda v: np.array(v, dtype=np.float64)
    return RawXccy(np.asarray(d_off, dtype=np.int64), cat("dtp"), cat("dts"), cat("dte"), cat("dal"),
                   np.asarray(f_off, dtype=np.int64), cat("ftp"), cat("fts"), cat("fte"), cat("fal"),
                   f64(d_ex).reshape(-1, 2), f64(f_ex).reshape(-1, 2),
                   np.array([bool(s._domestic_leg._notional_exchange) for s in swaps], dtype=bool),
                   np.array([bool(s._foreign_leg._notional_exchange) for s in swaps], dtype=bool),
                   f64([s._domestic_leg._notional for s in swaps]), f64([s._foreign_leg._notional for s in swaps]),
                   f64([s._domestic_leg._spread for s in swaps]), f64([s._foreign_leg._spread for s in swaps]),
                   f64([_sign(s._domestic_leg) for s in swaps]), f64([_sign(s._foreign_leg) for s in swaps]))


@dataclass
class XccyTerms:
    """Economic terms of n basis swaps on one currency pair (the `XccyBasisSwap` constructor arguments of the same
    names; scalars broadcast, sequences have length n).  ``effective_dt``: `Date`s or integer Excel serials;
    ``tenor``: tenor strings."""
    effective_dt: object
    tenor: object
    domestic_notional: object
    foreign_notional: object
    domestic_spread: object
    foreign_spread: object
    domestic_freq_type: object
    foreign_freq_type: object
    domestic_dc_type: object
    foreign_dc_type: object
    domestic_floating_index: object
    foreign_floating_index: object
    domestic_currency: object
    foreign_currency: object
    domestic_payment_lag: object = 0
    foreign_payment_lag: object = 0


_FIXED_DENOMINATOR = {DayCountTypes.ACT_365F: 365, DayCountTypes.ACT_360: 360, DayCountTypes.SIMPLE: gDaysInYear}


def _concat_raw(parts) -> RawXccy:
    out = {}
    for f in dataclasses.fields(RawXccy):
        cols = [getattr(p, f.name) for p in parts]
        if f.name in ("dom_off", "for_off"):
            ends = np.cumsum([c[-1] for c in cols])
            cols = [cols[0]] + [c[1:] + e for c, e in zip(cols[1:], ends[:-1])]
        out[f.name] = np.concatenate(cols)
    return RawXccy(**out)


def _take_raw(raw: RawXccy, pick) -> RawXccy:
    """Swaps ``pick`` of ``raw``, in that order."""
    out = {f.name: getattr(raw, f.name)[pick] for f in dataclasses.fields(RawXccy)
           if f.name[4:] not in ("off", "tp", "tpx", "ts", "te", "al")}
    for side, cols in (("dom", ("dom_tp", "dom_ts", "dom_te", "dom_al")), ("for", ("for_tpx", "for_ts", "for_te", "for_al"))):
        src = getattr(raw, side + "_off")
        lens = (src[1:] - src[:-1])[pick]
        off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
        idx = np.repeat(src[:-1][pick] - off[:-1], lens) + np.arange(off[-1])
        out[side + "_off"] = off
        for c in cols:
            out[c] = getattr(raw, c)[idx]
    return RawXccy(**out)


def _slice_columns(cols, pick):
    return {k: ((v[0][pick], v[1]) if isinstance(v, tuple) else v[pick]) for k, v in cols.items()}


def _raw_by_templates(cols, singles, value_dt, xdc) -> RawXccy:
    """One template swap object per distinct combination of schedule-defining terms, per-swap arrays gathered with
    NumPy (the route of `trades.compiler.compile_ois_terms`)."""
    from ...trades.rates.xccy_basis_swap import XccyBasisSwap
    from ...utils.date import Date
    names = ("tenor", "dfreq", "ffreq", "ddc", "fdc")
    from ...trades.compiler import unique_rows
    keys, inverse = unique_rows([cols["eff"], cols["dlag"], cols["flag"]] + [cols[k][0] for k in names])
    templates = []
    for row in keys:
        kw = {k: cols[k][1][int(row[3 + j])] for j, k in enumerate(names)}
        swap = XccyBasisSwap(effective_dt=Date._from_serial(int(row[0])), term_dt_or_tenor=kw["tenor"],
                             domestic_notional=1.0, foreign_notional=1.0, domestic_spread=0.0, foreign_spread=0.0,
                             domestic_freq_type=kw["dfreq"], foreign_freq_type=kw["ffreq"], domestic_dc_type=kw["ddc"],
                             foreign_dc_type=kw["fdc"], domestic_floating_index=singles["dindex"],
                             foreign_floating_index=singles["findex"], domestic_currency=singles["dccy"],
                             foreign_currency=singles["fccy"], domestic_payment_lag=int(row[1]),
                             foreign_payment_lag=int(row[2]))
        templates.append(raw_from_swaps([swap], value_dt, xdc))
    raw = _take_raw(_concat_raw(templates), inverse)
    raw.dom_n, raw.for_n = cols["dom_n"].copy(), cols["for_n"].copy()
    raw.dom_spread, raw.for_spread = cols["dspread"].copy(), cols["fspread"].copy()
    return raw


def _raw_by_arrays(cols, value_serial, xdc) -> RawXccy:
    """The same without any per-swap or per-schedule Python: `utils.schedule_np` on the whole book.  Returns the
    `RawXccy` and the mask of swaps whose two schedules are plain (see `schedule_np.backward_schedules`)."""
    from ...utils import schedule_np as S
    from ...utils.calendar import BusDayAdjustTypes
    from ...utils.frequency import annual_frequency
    bd = BusDayAdjustTypes.FOLLOWING                        # XccyBasisSwap's defaults: WEEKEND calendar, FOLLOWING, BACKWARD
    eff = cols["eff"]
    count, unit = S.parse_tenors(cols["tenor"][1])
    code = cols["tenor"][0]
    term = S.add_tenor(eff, count[code], unit[code])
    maturity = S.adjust(term, bd)
    if (eff > maturity).any():
        raise LibError("Start date after maturity date")
    den = lambda key: np.array([_FIXED_DENOMINATOR.get(d, 0) for d in cols[key][1]])[cols[key][0]]   # (0: not in this slice)
    mpp = lambda key: np.array([int(12 / annual_frequency(f)) for f in cols[key][1]], dtype=np.int64)[cols[key][0]]
    xden = _FIXED_DENOMINATOR[xdc]

    def leg(freq_key, dc_key, lag, pay_den):
        return S.leg_times(eff, term, mpp(freq_key), lag, bd, True, den(dc_key), value_serial, pay_den)

    d_off, dtp, dts_, dte, dal, d_plain = leg("dfreq", "ddc", cols["dlag"], None)
    f_off, ftp, fts, fte, fal, f_plain = leg("ffreq", "fdc", cols["flag"], xden)
    two = np.stack([eff, maturity], axis=1) - value_serial
    n = eff.shape[0]
    yes = np.ones(n, dtype=bool)
    raw = RawXccy(d_off, dtp, dts_, dte, dal, f_off, ftp, fts, fte, fal, two / den("ddc")[:, None], two / xden, yes, yes.copy(),
                  cols["dom_n"].copy(), cols["for_n"].copy(), cols["dspread"].copy(), cols["fspread"].copy(),
                  np.ones(n), -np.ones(n))                 # domestic received, foreign paid (xccy_basis_swap.py:150-168)
    return raw, d_plain & f_plain


def raw_from_terms(terms: XccyTerms, value_dt, xdc) -> RawXccy:
    """`RawXccy` of swaps given by their terms, without per-swap objects - the cross-currency counterpart of
    `trades.compiler.compile_ois_terms`.  Schedules, payment lags and year fractions
    are computed on arrays for the whole book (`utils.schedule_np`); swaps on a day count without a fixed
    denominator, or whose schedule needs the reference's de-duplication, go through one template object per
    distinct schedule instead."""
    from ...trades.compiler import _column
    dom_n = np.asarray(terms.domestic_notional, dtype=np.float64).reshape(-1)
    n = dom_n.shape[0]
    col = lambda v, kind: _column(v, n, kind)
    one = lambda v: v[0] if isinstance(v, (list, tuple, np.ndarray)) else v
    cols = {"eff": col(terms.effective_dt, "date"), "dlag": col(terms.domestic_payment_lag, "int"),
            "flag": col(terms.foreign_payment_lag, "int"), "tenor": col(terms.tenor, "code"),
            "dfreq": col(terms.domestic_freq_type, "code"), "ffreq": col(terms.foreign_freq_type, "code"),
            "ddc": col(terms.domestic_dc_type, "code"), "fdc": col(terms.foreign_dc_type, "code"),
            "dom_n": dom_n, "for_n": col(terms.foreign_notional, "float"),
            "dspread": col(terms.domestic_spread, "float"), "fspread": col(terms.foreign_spread, "float")}
    singles = {"dindex": one(terms.domestic_floating_index), "findex": one(terms.foreign_floating_index),
               "dccy": one(terms.domestic_currency), "fccy": one(terms.foreign_currency)}
    fixed = lambda key: np.array([d in _FIXED_DENOMINATOR for d in cols[key][1]], dtype=bool)[cols[key][0]]
    by_arrays = fixed("ddc") & fixed("fdc") if xdc in _FIXED_DENOMINATOR else np.zeros(n, dtype=bool)
    if not by_arrays.any():
        return _raw_by_templates(cols, singles, value_dt, xdc)
    pick = np.nonzero(by_arrays)[0]
    raw, plain = _raw_by_arrays(_slice_columns(cols, pick) if pick.size < n else cols, int(value_dt.excel_dt()), xdc)
    if pick.size == n and plain.all():
        return raw
    by_arrays[pick[~plain]] = False
    rest = np.nonzero(~by_arrays)[0]
    slow = _raw_by_templates(_slice_columns(cols, rest), singles, value_dt, xdc)
    fast = _take_raw(raw, np.nonzero(plain)[0])
    order = np.empty(n, dtype=np.int64)
    order[np.concatenate((np.nonzero(by_arrays)[0], rest))] = np.arange(n)
    return _take_raw(_concat_raw([fast, slow]), order)


def _exchange_flows(t2, notional, on, sign, scale):
    """Fixed flows (-N at the effective time, +N at maturity) of the legs that exchange notional: those strictly
    after the value time as CSR arrays, those AT the value time (discount factor 1) as a PV constant per swap."""
    amounts = np.stack([-notional, notional], axis=1)
    keep = on[:, None] & (t2 > 0.0)
    const = np.where(on[:, None] & (t2 == 0.0), sign[:, None] * amounts / scale, 0.0).sum(axis=1)
    off = np.concatenate(([0], np.cumsum(keep.sum(axis=1)))).astype(np.int64)
    return off, t2[keep], amounts[keep], const


def compile_xccy(raw: RawXccy, spot, df_x, df_f):
    """The three trade batches of the assembly above, plus the per-swap PV of the flows dated at the value time.

    ``df_x(t)`` / ``df_f(t)``: discount factors off the XCCY curve and off the foreign OIS curve's engine grid at
    arrays of times - on the product path the device lookups of `adr_curve_df` (the reference evaluates them inside
    its leg function, engine.py:1640-1712).  Returns ``(domestic, foreign_rates, foreign_flows, pv_const)``;
    ``foreign_rates`` carries `flt_weight`."""
    n = raw.n
    zeros_n, none = np.zeros(n), np.zeros(0)
    dfix_off, dfix_tp, dfix_pay, pv_const = _native.exchange_flows_host(raw.dom_exch_t, raw.dom_n, raw.dom_exch, raw.dom_sign, 1.0)
    domestic = TradeBatch(dfix_off, raw.dom_off, dfix_tp, dfix_pay, raw.dom_tp, raw.dom_ts, raw.dom_te, raw.dom_al,
                          raw.dom_n, raw.dom_spread, raw.dom_sign, raw.dom_sign)

    # foreign coupons, all swaps at once: forwards off the foreign OIS grid, discount factors off the XCCY knots (device
    # lookups), then the two foreign batches in one native pass over the coupons (`adr_xccy_assemble_host`;
    # `compile_xccy_np` is the array form it replaces, kept as the checker of tests/test_book_native.py)
    tp_x, ts, te, al = raw.for_tpx, raw.for_ts, raw.for_te, raw.for_al
    m = tp_x.shape[0]
    dx = df_x(np.concatenate((tp_x, [0.0])))                           # one device round trip per curve; the native pass takes
    df2 = df_f(np.concatenate((ts, te)))                               # the ratios D_x(tp) / D_x(0) and D_f(ts) / D_f(te) itself
    # (the two round trips on two host threads: no faster - measured 41 against 38 ms for the whole step)
    (kept_off, k_ts, k_te, k_al, k_c, fix_off, flow_tp, flow_pay, pv_const) = _native.xccy_assemble_host(
        raw.for_off, tp_x, ts, te, al, dx, df2, raw.for_n, raw.for_spread, raw.for_sign, spot, raw.for_exch_t,
        raw.for_exch, pv_const)
    foreign_rates = TradeBatch(np.zeros(n + 1, dtype=np.int64), kept_off, none, none, np.zeros(k_ts.shape[0]),
                               k_ts, k_te, k_al, raw.for_n, zeros_n, raw.for_sign, raw.for_sign, flt_weight=k_c)
    foreign_flows = TradeBatch(fix_off, np.zeros(n + 1, dtype=np.int64), flow_tp, flow_pay, none, none, none, none,
                               raw.for_n, zeros_n, raw.for_sign, raw.for_sign)
    return domestic, foreign_rates, foreign_flows, pv_const


def compile_xccy_np(raw: RawXccy, spot, df_x, df_f):
    """`compile_xccy` on NumPy arrays alone (the form the native pass replaces; the checker of the tests).
    The three trade batches of the assembly above, plus the per-swap PV of the flows dated at the value time.

    ``df_x(t)`` / ``df_f(t)``: discount factors off the XCCY curve and off the foreign OIS curve's engine grid at
    arrays of times - on the product path the device lookups of `adr_curve_df` (the reference evaluates them inside
    its leg function, engine.py:1640-1712).  Returns ``(domestic, foreign_rates, foreign_flows, pv_const)``;
    ``foreign_rates`` carries `flt_weight`."""
    n = raw.n
    zeros_n, none = np.zeros(n), np.zeros(0)
    dfix_off, dfix_tp, dfix_pay, pv_const = _exchange_flows(raw.dom_exch_t, raw.dom_n, raw.dom_exch, raw.dom_sign, 1.0)
    domestic = TradeBatch(dfix_off, raw.dom_off, dfix_tp, dfix_pay, raw.dom_tp, raw.dom_ts, raw.dom_te, raw.dom_al,
                          raw.dom_n, raw.dom_spread, raw.dom_sign, raw.dom_sign)

    # foreign coupons, all swaps at once: forwards off the foreign OIS grid, discount factors off the XCCY knots
    off = raw.for_off
    owner = np.repeat(np.arange(n), np.diff(off))
    tp_x, ts, te, al = raw.for_tpx, raw.for_ts, raw.for_te, raw.for_al
    c = df_x(tp_x) / df_x(0.0)                                          # relative to the value time
    accrues = al > 0
    fwd = np.where(accrues, (df_f(ts) / df_f(te) - 1.0) / np.where(accrues, al, 1.0), 0.0)
    live = tp_x >= 0.0
    keep = live & accrues
    kept_off = np.concatenate(([0], np.cumsum(np.bincount(owner[keep], minlength=n)))).astype(np.int64)
    foreign_rates = TradeBatch(np.zeros(n + 1, dtype=np.int64), kept_off, none, none, np.zeros(int(keep.sum())),
                               ts[keep], te[keep], al[keep], raw.for_n, zeros_n, raw.for_sign, raw.for_sign,
                               flt_weight=c[keep])

    amounts = (fwd + raw.for_spread[owner]) * al * raw.for_n[owner]
    at_value_time = live & (tp_x == 0.0)
    np.add.at(pv_const, owner[at_value_time], (raw.for_sign[owner] * amounts)[at_value_time] / spot)
    e_off, e_tp, e_pay, e_const = _exchange_flows(raw.for_exch_t, raw.for_n, raw.for_exch, raw.for_sign, spot)
    pv_const = pv_const + e_const
    later = live & (tp_x > 0.0)
    # per swap: its later coupons, then its exchanges
    cnt = np.bincount(owner[later], minlength=n) + np.diff(e_off)
    fix_off = np.concatenate(([0], np.cumsum(cnt))).astype(np.int64)
    flow_tp, flow_pay = np.empty(int(fix_off[-1])), np.empty(int(fix_off[-1]))
    c_owner = owner[later]                                  # sorted: the coupons are laid out swap by swap
    c_first = np.concatenate(([0], np.cumsum(np.bincount(c_owner, minlength=n))))[:-1]
    coupon_pos = fix_off[:-1][c_owner] + (np.arange(c_owner.size) - c_first[c_owner])     # from the swap's first slot
    flow_tp[coupon_pos], flow_pay[coupon_pos] = tp_x[later], amounts[later]
    e_owner = np.repeat(np.arange(n), np.diff(e_off))
    exch_pos = fix_off[1:][e_owner] - (e_off[1:][e_owner] - np.arange(e_off[-1]))         # back from its last slot
    flow_tp[exch_pos], flow_pay[exch_pos] = e_tp, e_pay
    foreign_flows = TradeBatch(fix_off, np.zeros(n + 1, dtype=np.int64), flow_tp, flow_pay, none, none, none, none,
                               raw.for_n, zeros_n, raw.for_sign, raw.for_sign)
    return domestic, foreign_rates, foreign_flows, pv_const


def cross_gamma_batches(raw: RawXccy):
    """Every foreign coupon that accrues and is paid after the value time as a trade of its own, twice: on the foreign
    OIS tables (notional N, accrual ts -> te, paid at time 0: value N D_f(ts) / D_f(te)) and on the XCCY tables (a unit
    flow at tp).  Returns ``(rates_batch, flows_batch, owner)``; ``owner[j]`` is the swap of coupon j."""
    n = raw.n
    owner = np.repeat(np.arange(n), np.diff(raw.for_off))
    keep = (raw.for_tpx > 0.0) & (raw.for_al > 0)
    m = int(keep.sum())
    own = owner[keep]
    off, zero_off, none = np.arange(m + 1, dtype=np.int64), np.zeros(m + 1, dtype=np.int64), np.zeros(0)
    ones, zeros = np.ones(m), np.zeros(m)
    rates = TradeBatch(zero_off, off, none, none, zeros.copy(), raw.for_ts[keep], raw.for_te[keep], raw.for_al[keep],
                       raw.for_n[own], zeros.copy(), ones, ones.copy())
    flows = TradeBatch(off, zero_off, raw.for_tpx[keep], ones.copy(), none, none, none, none, ones.copy(), zeros.copy(),
                       ones.copy(), ones.copy())
    return rates, flows, own


def cross_gamma_for_basis(ctx, for_dev, x_dev, raw: RawXccy, spot, n_basis, per_trade=True, aggregate=False, chunk=256):
    """d2 PV / d(foreign par rate) d(basis spread), per bp^2, domestic currency: ``[n, P_for, P_basis]`` per swap and /
    or the book's sum.  Two delta-only launches over the coupons (lite / general kernels) and a contraction."""
    rates, flows, own = cross_gamma_batches(raw)
    out = {}
    p_for = for_dev.n_pillars
    if rates.n_trades == 0:
        if per_trade:
            out["cross_for_basis"] = np.zeros((raw.n, p_for, n_basis))
        if aggregate:
            out["agg_cross_for_basis"] = np.zeros((p_for, n_basis))
        return out
    kw = dict(want_value=False, want_delta=True, want_gamma=False)
    d_r = np.asarray(_price(ctx, for_dev, rates, kw)["delta"])                    # 1e-4 grad_r [N D_f(ts)/D_f(te)]
    d_s = _trim(_price(ctx, x_dev, flows, kw)["delta"], "delta", n_basis)          # 1e-4 grad_s D_x(tp)
    scale = (raw.for_sign / spot)[own]
    d_r = d_r * scale[:, None]
    if aggregate:
        out["agg_cross_for_basis"] = d_r.T @ d_s
    if per_trade:
        cross = np.zeros((raw.n, p_for, n_basis))
        first = np.searchsorted(own, np.arange(raw.n + 1))                         # coupons are laid out swap by swap
        for lo in range(0, raw.n, chunk):
            hi = min(lo + chunk, raw.n)
            a, b = first[lo], first[hi]
            if a == b:
                continue
            outer = d_r[a:b, :, None] * d_s[a:b, None, :]
            starts = first[lo:hi] - a
            live = np.flatnonzero(np.diff(np.append(starts, b - a)) > 0)           # reduceat needs non-empty segments
            cross[lo + live] = np.add.reduceat(outer, starts[live], axis=0)
        out["cross_for_basis"] = cross
    return out


def _xccy_device_curve(ctx, xccy):
    """The XCCY curve's tables as an `adr_curve` (cached on the curve object).  The fast kernel's packed layout
    wants an even pillar count (16-byte gamma stores), so an odd basis ladder gets one all-zero pillar appended
    here and dropped again from the results (`_trim`)."""
    times, dfs = np.asarray(xccy._times, dtype=np.float64), np.asarray(xccy._dfs, dtype=np.float64)
    # valid for exactly these knots, this scheme and this context
    key = (times.tobytes(), dfs.tobytes(), xccy._interp_type.value)
    hit = getattr(xccy, "_adr_device_curve", None)
    if hit is not None and hit[0] == key and hit[1] is ctx:
        return hit[2]
    jac = getattr(xccy, "_jac_basis", None)
    hess = getattr(xccy, "_hess_basis", None) if jac is not None else None
    jac = np.zeros((times.size, 2)) if jac is None else np.asarray(jac, dtype=np.float64)
    hess = None if hess is None else np.asarray(hess, dtype=np.float64)
    if jac.shape[1] % 2:
        jac = np.pad(jac, ((0, 0), (0, 1)))
        hess = None if hess is None else np.pad(hess, ((0, 0), (0, 1), (0, 1)))
    dev = _native.DeviceCurve(ctx, xccy._interp_type.value, times, dfs, jac, hess)
    xccy._adr_device_curve = (key, ctx, dev)
    return dev


def _trim(a, kind, P):
    """Drop the padding pillar of `_xccy_device_curve` from a delta ([..., P']) or gamma ([..., P', P']) array."""
    a = np.asarray(a)
    return a[..., :P] if kind == "delta" else a[..., :P, :P]


def _curves(engine, swaps):
    first = swaps[0]
    for s in swaps:
        if (s._domestic_floating_index, s._foreign_floating_index, s._domestic_currency, s._foreign_currency) != \
           (first._domestic_floating_index, first._foreign_floating_index, first._domestic_currency, first._foreign_currency):
            raise LibError("a batch of cross-currency swaps must share its currencies and floating indices")
    return _curves_for(engine, first._domestic_floating_index, first._foreign_floating_index,
                       first._domestic_currency, first._foreign_currency)


def _curves_for(engine, dom_index, for_index, dom_ccy, for_ccy):
    model = engine.model
    dom_model = getattr(model.curves, dom_index.name)
    for_model = getattr(model.curves, for_index.name)
    name = f"{for_ccy.name}_{dom_ccy.name}_BASIS"
    try:
        xccy = getattr(model.curves, name)
    except AttributeError:
        raise LibError(f"XCCY curve {name} not found in model.")
    if getattr(xccy, "_jac_basis", None) is None:
        raise LibError("the XCCY curve carries no basis Jacobian (build it with use_ad=True)")
    dom_cur, for_cur = engine._device_curve(dom_model), engine._device_curve(for_model)
    x_dev = _xccy_device_curve(dom_cur["ctx"], xccy)
    return dom_model, for_model, xccy, dom_cur, for_cur, x_dev


def book_batches(engine, swaps):
    """Curves and the three trade batches of a book given as `XccyBasisSwap` objects or as `XccyTerms`:
    ``(dom_model, for_model, xccy, dom_cur, for_cur, x_dev, (domestic, foreign_rates, foreign_flows), pv_const, spot,
    raw)``.
    The per-coupon discount factors come from the device (`adr_curve_df` on the uploaded XCCY and foreign tables)."""
    if isinstance(swaps, XccyTerms):
        one = lambda v: v[0] if isinstance(v, (list, tuple, np.ndarray)) else v
        cur = _curves_for(engine, one(swaps.domestic_floating_index), one(swaps.foreign_floating_index),
                          one(swaps.domestic_currency), one(swaps.foreign_currency))
    else:
        swaps = list(swaps)
        if not swaps:
            raise LibError("price_xccy_batch needs at least one swap (the book's currency pair names its curves)")
        cur = _curves(engine, swaps)
    dom_model, for_model, xccy, dom_cur, for_cur, x_dev = cur
    ctx = dom_cur["ctx"]
    raw = (raw_from_terms(swaps, engine.model.value_dt, xccy._dc_type) if isinstance(swaps, XccyTerms)
           else raw_from_swaps(swaps, engine.model.value_dt, xccy._dc_type))
    spot = xccy._spot_fx
    domestic, foreign_rates, foreign_flows, pv_const = compile_xccy(
        raw, spot, lambda t: _native.curve_df(ctx, x_dev, t), lambda t: _native.curve_df(ctx, for_cur["dev"], t))
    return cur + ((domestic, foreign_rates, foreign_flows), pv_const, spot, raw)


def compile_xccy_legs(raw: RawXccy, spot):
    """The two trade batches of the ONE-LAUNCH foreign leg (`adr_price_xccy_foreign`): the domestic legs as in `compile_xccy`,
    and the foreign legs AS THEY ARE - payment times on the XCCY curve's day count, accrual times on the leg's own, the
    notional exchanges as fixed flows - with no discount factor from any curve: the kernel looks D_x(tp), D_f(ts), D_f(te) up
    itself (engine.py:1640-1733).  Returns ``(domestic, foreign_legs, pv_const)``; a batch built this way can be priced again
    under other curves without touching the host."""
    dfix_off, dfix_tp, dfix_pay, pv_const = _native.exchange_flows_host(raw.dom_exch_t, raw.dom_n, raw.dom_exch, raw.dom_sign, 1.0)
    domestic = TradeBatch(dfix_off, raw.dom_off, dfix_tp, dfix_pay, raw.dom_tp, raw.dom_ts, raw.dom_te, raw.dom_al,
                          raw.dom_n, raw.dom_spread, raw.dom_sign, raw.dom_sign)
    e_off, e_tp, e_pay, e_const = _native.exchange_flows_host(raw.for_exch_t, raw.for_n, raw.for_exch, raw.for_sign, spot)
    foreign_legs = TradeBatch(e_off, raw.for_off, e_tp, e_pay, raw.for_tpx, raw.for_ts, raw.for_te, raw.for_al,
                              raw.for_n, raw.for_spread, raw.for_sign, raw.for_sign)
    return domestic, foreign_legs, pv_const + e_const


# VALUE / DELTA requests price the foreign leg in ONE launch on two curves (False: always the three-batch assembly)
FUSED_FOREIGN_LEG = True


def _price_fused(engine, swaps, want_value, want_delta, per_trade, aggregate):
    """PV and the three delta ladders with the foreign leg in one launch; None when the launch does not take the book
    (a leg of more than 390 coupons, a leg whose accrual ends all coincide with its payment times, curves it is not built
    for: more than 32 pillars, one of the two on LINEAR_FWD_RATES and the other not): the caller then falls back to the three-batch assembly."""
    if isinstance(swaps, XccyTerms):
        one = lambda v: v[0] if isinstance(v, (list, tuple, np.ndarray)) else v
        cur = _curves_for(engine, one(swaps.domestic_floating_index), one(swaps.foreign_floating_index),
                          one(swaps.domestic_currency), one(swaps.foreign_currency))
    else:
        cur = _curves(engine, swaps)
    dom_model, for_model, xccy, dom_cur, for_cur, x_dev = cur
    ctx = dom_cur["ctx"]
    raw = (raw_from_terms(swaps, engine.model.value_dt, xccy._dc_type) if isinstance(swaps, XccyTerms)
           else raw_from_swaps(swaps, engine.model.value_dt, xccy._dc_type))
    spot = xccy._spot_fx
    domestic, foreign_legs, pv_const = compile_xccy_legs(raw, spot)
    dom_tr, for_tr = _native.upload_many(ctx, [domestic, foreign_legs])
    try:
        try:
            frn = _native.price_xccy_foreign(ctx, for_cur["dev"], x_dev, for_tr, want_value=want_value, want_delta=want_delta,
                                             per_trade=per_trade, aggregate=aggregate)
        except LibError as exc:
            if "(-2)" in str(exc):          # ADR_ERR_UNSUPPORTED: not a book for this launch
                return None
            raise
        dom = _native.price(ctx, dom_cur["dev"], dom_tr, want_value=want_value, want_delta=want_delta, want_gamma=False,
                            per_trade=per_trade, aggregate=aggregate)
    finally:
        dom_tr.close(); for_tr.close()
    n_basis = len(xccy.swap_times)
    out = {}
    for pre in (("",) if per_trade else ()) + (("agg_",) if aggregate else ()):
        if want_value:
            const = pv_const if pre == "" else float(pv_const.sum())
            out[pre + "pv"] = dom[pre + "pv"] + frn[pre + "pv"] / spot + const
        if want_delta:
            out[pre + "delta_dom"] = np.asarray(dom[pre + "delta"])
            out[pre + "delta_for"] = np.asarray(frn[pre + "delta_foreign"]) / spot
            out[pre + "delta_basis"] = _trim(frn[pre + "delta_basis"], "delta", n_basis) / spot
    out["tenors"] = (to_tenor(list(dom_model.swap_times)), to_tenor(list(for_model.swap_times)), to_tenor(list(xccy.swap_times)))
    return out


def price_xccy_batch(engine, swaps, reqs, per_trade=True, aggregate=False, cross_gamma=False):
    """VALUE / DELTA / GAMMA of a book of cross-currency basis swaps on one currency pair: three launches.
    ``swaps``: `XccyBasisSwap` objects, or `XccyTerms` (no per-swap objects: the vectorised compiler).

    Returns a dict: ``pv [n]``, ``delta_dom [n, P_d]``, ``delta_for [n, P_f]``, ``delta_basis [n, P_b]`` and the
    three ``gamma_*`` (per request, when ``per_trade``), ``agg_*`` sums over the book (when ``aggregate``);
    with ``cross_gamma`` and GAMMA also ``cross_for_basis [n, P_f, P_b]`` (`cross_gamma_for_basis`);
    everything in domestic currency, per bp / bp^2."""
    reqs = set(reqs)
    if FUSED_FOREIGN_LEG and RequestTypes.GAMMA not in reqs and (isinstance(swaps, XccyTerms) or len(list(swaps)) > 0):
        if not isinstance(swaps, XccyTerms):
            swaps = list(swaps)
        fused = _price_fused(engine, swaps, RequestTypes.VALUE in reqs, RequestTypes.DELTA in reqs, per_trade, aggregate)
        if fused is not None:
            return fused
    (dom_model, for_model, xccy, dom_cur, for_cur, x_dev,
     (domestic, foreign_rates, foreign_flows), pv_const, spot, raw) = book_batches(engine, swaps)
    ctx = dom_cur["ctx"]
    want_value = RequestTypes.VALUE in reqs
    want_gamma = RequestTypes.GAMMA in reqs
    want_delta = want_gamma or RequestTypes.DELTA in reqs
    kw = dict(want_delta=want_delta, want_gamma=want_gamma, per_trade=per_trade, aggregate=aggregate)
    # the three batches are uploaded together (one host thread each), then priced one after the other
    todo = [(dom_cur["dev"], domestic, dict(kw, want_value=want_value)), (x_dev, foreign_flows, dict(kw, want_value=want_value))]
    if want_delta:
        todo.append((for_cur["dev"], foreign_rates, dict(kw, want_value=False)))
    uploaded = _native.upload_many(ctx, [b for _, b, _ in todo])
    try:
        priced = [_native.price(ctx, cur, tr, **k) for (cur, _, k), tr in zip(todo, uploaded)]
    finally:
        for tr in uploaded:
            tr.close()
    dom, frn = priced[0], priced[1]
    rates = priced[2] if want_delta else {}
    out = {}
    for pre in (("",) if per_trade else ()) + (("agg_",) if aggregate else ()):
        if want_value:
            const = pv_const if pre == "" else float(pv_const.sum())
            out[pre + "pv"] = dom[pre + "pv"] + frn[pre + "pv"] / spot + const
        for kind in (("delta",) if want_delta else ()) + (("gamma",) if want_gamma else ()):
            out[f"{pre}{kind}_dom"] = np.asarray(dom[pre + kind])
            out[f"{pre}{kind}_for"] = np.asarray(rates[pre + kind]) / spot
            out[f"{pre}{kind}_basis"] = _trim(frn[pre + kind], kind, len(xccy.swap_times)) / spot
    if cross_gamma and want_gamma:
        out.update(cross_gamma_for_basis(ctx, for_cur["dev"], x_dev, raw, spot, len(xccy.swap_times), per_trade, aggregate))
    out["tenors"] = (to_tenor(list(dom_model.swap_times)), to_tenor(list(for_model.swap_times)),
                     to_tenor(list(xccy.swap_times)))
    return out


# What `Risk.cross_gamma(foreign OIS, basis)` of a GAMMA request holds:
#   "direct" (default) - d2 PV / d r_for d s_basis with the XCCY curve's knot DFs held fixed w.r.t. the foreign curve:
#                        sum_j grad_r[N D_f(ts_j)/D_f(te_j)] (x) grad_s[D_x(tp_j)] (`cross_gamma_for_basis`); the matrix is
#                        labelled `definition="direct"`.  NOT the reference's block (engine.py:1892-1958), which keeps
#                        only the bootstrap's mixed-Hessian term and contracts tensors of different sizes;
#   "off"              - no cross-gamma is attached (`has_cross_gamma` is False), for consumers that compare slot by slot
#                        with the reference and would otherwise read a differently defined number.
# Like the reference (engine.py:1894) nothing is attached when the XCCY curve carries no `_mixed_hess_foreign_basis`.
CROSS_GAMMA_MODE = "direct"


def compute_xccy(engine, derivative, reqs):
    if RequestTypes.CASHFLOWS in reqs:
        # the reference's block for this request (engine.py:1970-1986) ends in a NameError (`risk_ccy` is never
        # assigned in _compute_xccy), so there is no behaviour to mirror
        raise NotImplementedError("CASHFLOWS is not available for cross-currency swaps")
    model = engine.model
    x_name = f"{derivative._foreign_currency.name}_{derivative._domestic_currency.name}_BASIS"
    x_curve = getattr(model.curves, x_name, None)             # (a missing curve is reported by `_curves_for` below)
    attach = (CROSS_GAMMA_MODE == "direct" and RequestTypes.GAMMA in reqs
              and getattr(x_curve, "_mixed_hess_foreign_basis", None) is not None)
    res = price_xccy_batch(engine, [derivative], reqs, cross_gamma=attach)
    ccy = derivative._domestic_currency
    curves = (derivative._domestic_floating_index, derivative._foreign_floating_index, CurveTypes.USD_GBP_BASIS)
    value = delta = gamma = None
    if RequestTypes.VALUE in reqs:
        value = Valuation(amount=float(res["pv"][0]), currency=ccy)
    if RequestTypes.DELTA in reqs:
        delta = Risk([Delta(np.array(res[k][0]), t, ccy, c)
                      for k, t, c in zip(("delta_dom", "delta_for", "delta_basis"), res["tenors"], curves)])
    if RequestTypes.GAMMA in reqs:
        cross = None
        if attach:
            cross = [CrossGamma(risk_matrix=np.array(res["cross_for_basis"][0]), tenors_curve1=res["tenors"][1],
                                tenors_curve2=res["tenors"][2], curve_type_1=derivative._foreign_floating_index,
                                curve_type_2=CurveTypes.USD_GBP_BASIS, currency=ccy, definition="direct")]
        gamma = Risk([Gamma(np.array(res[k][0]), t, ccy, c)
                      for k, t, c in zip(("gamma_dom", "gamma_for", "gamma_basis"), res["tenors"], curves)],
                     cross_gammas=cross)
    return AnalyticsResult(value=value, risk=delta, gamma=gamma)


def compute_ois_xccy_collateral(engine, derivative, reqs, collateral_ccy):
    """An OIS collateralised in another currency (`Engine._compute_ois_xccy_collateral`, engine.py:217-503):
    both legs discounted on the ``{swap ccy}_{collateral ccy}_XCCY`` curve, forwards off the swap's own OIS curve,
    PV and ladders converted to the collateral currency.  Pieces 2 and 3 of the assembly above with the OIS's
    float leg in the foreign leg's role and its fixed coupons among the flows; every time in the fixed leg's day
    count (:264-283).  Like the reference: no GAMMA (:489-494), CASHFLOWS is an empty table (:496-501)."""
    model = engine.model
    if RequestTypes.GAMMA in reqs:
        raise NotImplementedError("GAMMA not yet supported for OIS with cross-currency collateral. "
                                  "Only VALUE and DELTA are currently implemented.")
    ois_model = getattr(model.curves, derivative._floating_index.name)
    name = f"{derivative._currency.name}_{collateral_ccy.name}_XCCY"
    try:
        xccy = getattr(model.curves, name)
    except AttributeError:
        raise LibError(f"XCCY curve {name} not found in model. Required for cross-currency collateral valuation.")
    spot = xccy._spot_fx
    value_dt = model.value_dt
    fx, fl = derivative._fixed_leg, derivative._float_leg
    dc = fx._dc_type
    ois_cur = engine._device_curve(ois_model)
    ctx = ois_cur["ctx"]
    x_dev = _xccy_device_curve(ctx, xccy)
    df_o = lambda t: _native.curve_df(ctx, ois_cur["dev"], t)       # forwards off the swap's own OIS tables
    df_x = lambda t: _native.curve_df(ctx, x_dev, t)                # discounting off the XCCY tables

    tp, ts, te = (_times(d, value_dt, dc) for d in (fl._payment_dts, fl._start_accrued_dts, fl._end_accrued_dts))
    al = np.asarray(fl._year_fracs, dtype=np.float64)
    accrues, live = al > 0, tp >= 0.0
    fwd = np.where(accrues, (df_o(ts) / df_o(te) - 1.0) / np.where(accrues, al, 1.0), 0.0)
    amounts = _sign(fl) * (fwd + fl._spread) * al * fl._notional
    fixed_tp = _times(fx._payment_dts, value_dt, dc)
    fixed_pay = _sign(fx) * fx._cpn * np.asarray(fx._year_fracs, dtype=np.float64) * fx._notional
    if fx._principal != 0.0 or fl._principal != 0.0:
        raise LibError("principal payments are not supported on this path")     # every OIS has principal 0 (ois.py:149)
    later, fixed_later = live & (tp > 0.0), fixed_tp > 0.0
    pv_const = float(amounts[live & (tp == 0.0)].sum())          # paid at the value time: discount factor 1
    none, one, zero = np.zeros(0), np.ones(1), np.zeros(1)
    flow_tp, flow_pay = np.concatenate([tp[later], fixed_tp[fixed_later]]), np.concatenate([amounts[later], fixed_pay[fixed_later]])
    flows = TradeBatch(np.array([0, flow_tp.size]), np.array([0, 0]), flow_tp, flow_pay, none, none, none, none,
                       one, zero, one, one)
    want_delta = RequestTypes.DELTA in reqs
    has_jac = getattr(xccy, "_jac_basis", None) is not None
    on_x = _price(ctx, x_dev, flows, dict(want_value=True, want_delta=want_delta and has_jac, want_gamma=False))

    value = delta = cashflows = None
    if RequestTypes.VALUE in reqs:
        value = Valuation(amount=float(on_x["pv"][0] + pv_const) / spot, currency=collateral_ccy)
    if want_delta:
        keep = live & accrues
        weight = df_x(tp[keep]) / df_x(0.0)
        m = int(keep.sum())
        rates = TradeBatch(np.array([0, 0]), np.array([0, m]), none, none, np.zeros(m), ts[keep], te[keep], al[keep],
                           np.array([fl._notional]), zero, one, np.array([_sign(fl)]), flt_weight=weight)
        on_o = _price(ctx, ois_cur["dev"], rates, dict(want_value=False, want_delta=True, want_gamma=False))
        ladders = [Delta(np.array(on_o["delta"][0]) / spot, to_tenor(list(ois_model.swap_times)), collateral_ccy,
                         derivative._floating_index)]
        if has_jac:
            ladders.append(Delta(_trim(on_x["delta"][0], "delta", len(xccy.swap_times)) / spot,
                                 to_tenor(list(xccy.swap_times)), collateral_ccy,
                                 CurveTypes.USD_GBP_BASIS))
        delta = Risk(ladders)
    if RequestTypes.CASHFLOWS in reqs:
        from ...requests.results import Cashflows
        cashflows = Cashflows([], derivative._currency)
    return AnalyticsResult(value=value, risk=delta, gamma=None, cashflows=cashflows)


def _price(ctx, dev_curve, batch, kw):
    trades = _native.DeviceTrades(ctx, batch)
    try:
        return _native.price(ctx, dev_curve, trades, **kw)
    finally:
        trades.close()
