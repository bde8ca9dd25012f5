"""Deterministic synthetic GBP/USD cross-currency books for benchmarks (BASELINE.json configs[3] and [4]).

Market: the two 32-pillar quote sets of the reference's tests (GBP SONIA ACT/365F, USD SOFR ACT/360, both
FLAT_FWD_RATES here) and a 17-pillar GBP/USD basis curve, 25 bp at 1Y rising linearly to 45 bp at 40Y.
Book (`synthesize_book`): every swap is drawn on its own - receive SONIA + U(0, 5 bp) / pay SOFR + U(10, 60 bp),
remaining maturity U{12..360} whole months, effective today or 4 / 9 months ago (a third each), annual domestic leg,
annual or semi-annual foreign leg, foreign notional round(U(1e6, 5e7), -5), domestic notional = spot x foreign -
`numpy.random.default_rng(seed)`; about 2 000 distinct schedules, no two swaps alike.  The terms go through the
vectorised compiler (`xccy_engine.raw_from_terms`: one template per distinct schedule, NumPy gathers) and the
per-coupon discount factors through the device lookups (`adr_curve_df`).  `template_swaps` / `take` remain for the
small object-path tests.
"""
from __future__ import annotations

import numpy as np

from ..utils.calendar import BusDayAdjustTypes
from ..utils.currency import CurrencyTypes
from ..utils.day_count import DayCountTypes
from ..utils.frequency import FrequencyTypes
from ..utils.global_types import CurveTypes, InterpTypes, SwapTypes
from .compiler import TradeBatch

SPOT = 0.79
BASIS_TENORS = ["1Y", "18M", "2Y", "3Y", "4Y", "5Y", "6Y", "7Y", "8Y", "9Y", "10Y", "12Y", "15Y", "20Y", "25Y", "30Y", "40Y"]


def build_market(value_dt, gbp_px, usd_px, tenors, interp=InterpTypes.FLAT_FWD_RATES):
    """A `Model` with GBP_OIS_SONIA, USD_OIS_SOFR and USD_GBP_BASIS."""
    from ..models.models import Model
    m = Model(value_dt)
    for name, px, dc in (("GBP_OIS_SONIA", gbp_px, DayCountTypes.ACT_365F), ("USD_OIS_SOFR", usd_px, DayCountTypes.ACT_360)):
        m.build_curve(name=name, px_list=list(px), tenor_list=list(tenors), spot_days=0, swap_type=SwapTypes.PAY,
                      fixed_dcc_type=dc, fixed_freq_type=FrequencyTypes.ANNUAL, float_freq_type=FrequencyTypes.ANNUAL,
                      float_dc_type=dc, bus_day_type=BusDayAdjustTypes.MODIFIED_FOLLOWING, interp_type=interp)
    m.build_xccy_curve(name="USD_GBP_BASIS", domestic_curve_name="GBP_OIS_SONIA", foreign_curve_name="USD_OIS_SOFR",
                       basis_spreads=list(np.linspace(25.0, 45.0, len(BASIS_TENORS))), tenor_list=BASIS_TENORS,
                       spot_fx=SPOT, domestic_dc_type=DayCountTypes.ACT_365F, foreign_dc_type=DayCountTypes.ACT_360,
                       interp_type=interp)
    return m


def template_swaps(value_dt):
    from .rates.xccy_basis_swap import XccyBasisSwap
    out = []
    for years in range(1, 31):
        for back in (0, 4, 9):
            for freq in (FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL):
                eff = value_dt.add_months(-back)
                out.append(XccyBasisSwap(effective_dt=eff, term_dt_or_tenor=eff.add_months(12 * years + back),
                                         domestic_notional=SPOT * 1e6, foreign_notional=1e6, domestic_spread=0.0,
                                         foreign_spread=0.0030, domestic_freq_type=FrequencyTypes.ANNUAL,
                                         foreign_freq_type=freq, domestic_dc_type=DayCountTypes.ACT_365F,
                                         foreign_dc_type=DayCountTypes.ACT_360,
                                         domestic_floating_index=CurveTypes.GBP_OIS_SONIA,
                                         foreign_floating_index=CurveTypes.USD_OIS_SOFR,
                                         domestic_currency=CurrencyTypes.GBP, foreign_currency=CurrencyTypes.USD))
    return out


def take(batch: TradeBatch, pick, scale) -> TradeBatch:
    """Trades ``pick`` of a batch (with repetition), their notionals and fixed amounts multiplied by ``scale``."""
    pick = np.asarray(pick, dtype=np.int64)
    scale = np.asarray(scale, dtype=np.float64)

    def gather(off, cols):
        cnt = np.diff(off)[pick]
        new_off = np.concatenate(([0], np.cumsum(cnt))).astype(np.int64)
        idx = np.repeat(off[:-1][pick] - new_off[:-1], cnt) + np.arange(new_off[-1])
        return new_off, [c[idx] for c in cols], np.repeat(np.arange(pick.size), cnt)

    fix_off, (fix_tp, fix_pay), owner = gather(batch.fix_off, (batch.fix_tp, batch.fix_pay))
    weighted = batch.flt_weight is not None
    flt_off, cols, _ = gather(batch.flt_off, (batch.flt_tp, batch.flt_ts, batch.flt_te, batch.flt_alpha)
                              + ((batch.flt_weight,) if weighted else ()))
    return TradeBatch(fix_off, flt_off, fix_tp, fix_pay * scale[owner], cols[0], cols[1], cols[2], cols[3],
                      batch.notional[pick] * scale, batch.spread[pick], batch.fix_sign[pick], batch.flt_sign[pick],
                      cols[4] if weighted else None)


def draw_terms(value_dt, n, seed=20240430):
    """`xccy_engine.XccyTerms` of ``n`` distinct swaps, and an estimate of each swap's coupon count (for sharding)."""
    from ..market.position.xccy_engine import XccyTerms
    rng = np.random.default_rng(seed)
    months = rng.integers(12, 361, n)
    back = np.array([0, 4, 9])[rng.integers(0, 3, n)]
    semi = rng.random(n) < 0.5
    for_n = np.round(rng.uniform(1e6, 5e7, n), -5)
    eff_of = {int(b): int(value_dt.add_months(-int(b)).excel_dt()) for b in (0, 4, 9)}
    total = months + back
    tenor_months = np.unique(total)
    terms = XccyTerms(effective_dt=np.array([eff_of[int(b)] for b in back], dtype=np.int64),
                      tenor=(np.searchsorted(tenor_months, total), [f"{int(m)}M" for m in tenor_months]),
                      domestic_notional=SPOT * for_n, foreign_notional=for_n,
                      domestic_spread=np.round(rng.uniform(0.0, 0.0005, n), 6),
                      foreign_spread=np.round(rng.uniform(0.0010, 0.0060, n), 6),
                      domestic_freq_type=FrequencyTypes.ANNUAL,
                      foreign_freq_type=(semi.astype(np.int64), [FrequencyTypes.ANNUAL, FrequencyTypes.SEMI_ANNUAL]),
                      domestic_dc_type=DayCountTypes.ACT_365F, foreign_dc_type=DayCountTypes.ACT_360,
                      domestic_floating_index=CurveTypes.GBP_OIS_SONIA, foreign_floating_index=CurveTypes.USD_OIS_SOFR,
                      domestic_currency=CurrencyTypes.GBP, foreign_currency=CurrencyTypes.USD)
    years = (months + back + 11) // 12
    return terms, years * (2 + semi.astype(np.int64))


def slice_terms(terms, lo, hi):
    """Swaps lo..hi-1 of an `XccyTerms` (per-swap sequences are cut, scalars kept)."""
    import dataclasses
    n = len(np.asarray(terms.domestic_notional).reshape(-1))
    coded = lambda v: isinstance(v, tuple) and len(v) == 2 and isinstance(v[0], np.ndarray)     # (codes, table) columns
    cut = lambda v: ((v[0][lo:hi], v[1]) if coded(v) else
                     v[lo:hi] if isinstance(v, (list, tuple, np.ndarray)) and len(v) == n else v)
    return dataclasses.replace(terms, **{f.name: cut(getattr(terms, f.name)) for f in dataclasses.fields(terms)})


def synthesize_book(engine, value_dt, n, seed=20240430, rank=0, world_size=1):
    """The three trade batches of a book of ``n`` distinct swaps (see `xccy_engine.compile_xccy`) and their device
    curves: ``[(batch, device curve)] * 3`` in the order domestic, foreign rates, foreign flows, plus ``spot``.
    With ``world_size > 1``: rank ``rank``'s contiguous share of that ONE book, cut by coupon count
    (`distributed.shard_by_work`); only the share is compiled."""
    from ..market.position import xccy_engine as XE
    terms, work = draw_terms(value_dt, n, seed)
    if world_size > 1:
        from ..distributed import shard_by_work
        lo, hi = shard_by_work(work, world_size)[rank]
        terms = slice_terms(terms, lo, hi)
    dom_model, for_model, xccy, dom_cur, for_cur, x_dev, (dom, rates, flows), pv_const, spot, _ = XE.book_batches(engine, terms)
    return [(dom, dom_cur["dev"]), (rates, for_cur["dev"]), (flows, x_dev)], spot
