"""Engine-side curve construction: knot grid, bootstrap and its par-rate derivatives.

Product (host) code - runs once per curve, the result is uploaded to the GPU.
It replaces `Engine.build_curve_ad` (cavour/market/position/engine.py:2246-2360)
and the `jacrev` / `hessian` calls of `Engine._cached_curve` (:2362-2412):

* the knot grid is the reference's: one t=0 point plus one point per calibration
  swap and coupon period at the *exact* running sum of that swap's accrual
  fractions, stably sorted, duplicates kept; a point's predecessor is the first
  sorted point whose ``round(t, 2)`` key equals the key of its previous coupon;
* knot DFs follow the scan ``d = (1 - r*PV01_prev) / (1 + r*acc)``;
* where the reference differentiates the scan with JAX, the same derivatives are
  propagated here in closed form alongside the values (forward recurrences for
  d DF/d r [K,P] and d2 DF/d r2 [K,P,P]).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np


@dataclass
class EngineCurve:
    """What the reference's per-Engine cache dict holds (engine.py:2405-2411)."""
    times: np.ndarray            # [K] knot times, sorted, duplicates kept
    dfs: np.ndarray              # [K]
    jac: np.ndarray              # [K, P]    d dfs / d par rates
    hess: np.ndarray | None      # [K, P, P] d2 dfs / d par rates2
    key_collisions: List[Tuple[float, float, float]] = field(default_factory=list)
    # the rate-independent description of the scan (what a device-side rebuild under other rates needs)
    acc: np.ndarray | None = None        # [K] accrual fraction of the period ending at the knot
    pillar: np.ndarray | None = None     # [K] calibration swap of the knot
    prev_idx: np.ndarray | None = None   # [K] knot whose PV01 the knot builds on, -1 for none
    rates: np.ndarray | None = None      # [P] par rates the values were built with

    @property
    def n_knots(self) -> int:
        return int(self.times.shape[0])

    @property
    def n_pillars(self) -> int:
        return int(self.jac.shape[1])


def expand_knot_grid(swap_rates, year_fracs):
    """Sorted bootstrap points (engine.py:2283-2334).

    Returns ``times, acc, pillar, prev_idx`` arrays of length K = 1 + sum(len(fracs))
    and the list of rounded-key collisions between distinct times (these make the
    reference pick a wrong predecessor; they are reported, not repaired)."""
    mats = [0.0]
    keys = [0.0]
    accs = [0.0]
    prev_keys = [None]
    pillars = [0]          # the t=0 point borrows pillar 0's rate; its DF is 1 regardless
    for i, fracs in enumerate(year_fracs):
        run = 0.0
        for j, frac in enumerate(fracs):
            frac = float(frac)
            before = run
            run += frac
            mats.append(run)
            keys.append(round(run, 2))          # Python's correctly rounded round(), as the reference
            accs.append(frac)
            prev_keys.append(round(before, 2) if j > 0 else None)
            pillars.append(i)
    order = sorted(range(len(mats)), key=mats.__getitem__)   # stable: ties keep swap order

    first_with_key = {}
    collisions = []
    for pos, src in enumerate(order):
        kk = keys[src]
        if kk not in first_with_key:
            first_with_key[kk] = pos
        else:
            other = mats[order[first_with_key[kk]]]
            if abs(other - mats[src]) > 1e-6:
                collisions.append((kk, other, mats[src]))
    prev_idx = [(-1 if prev_keys[src] is None else first_with_key.get(prev_keys[src], -1)) for src in order]
    return (np.array([mats[s] for s in order], dtype=np.float64),
            np.array([accs[s] for s in order], dtype=np.float64),
            np.array([pillars[s] for s in order], dtype=np.int64),
            np.array(prev_idx, dtype=np.int64),
            collisions)


def build_engine_curve(swap_rates, swap_times, year_fracs, with_hessian: bool = True) -> EngineCurve:
    """Bootstrap the engine grid and propagate first/second par-rate derivatives."""
    rates = np.array([float(r) for r in swap_rates], dtype=np.float64)
    P = rates.shape[0]
    times, acc, pillar, prev_idx, collisions = expand_knot_grid(swap_rates, year_fracs)
    K = times.shape[0]

    dfs = np.zeros(K)
    jac = np.zeros((K, P))
    pv01 = np.zeros(K)
    dpv01 = np.zeros((K, P))
    hess = d2pv01 = None
    if with_hessian:
        hess = np.zeros((K, P, P))
        d2pv01 = np.zeros((K, P, P))

    for i in range(K):
        s = int(pillar[i])
        r = rates[s]
        a = acc[i]
        pi = int(prev_idx[i])
        # A predecessor that sorts after the point (possible only under a key
        # collision) has not been written yet in the reference's scan and reads as 0.
        if 0 <= pi < i:
            Pp, dPp = pv01[pi], dpv01[pi]
            d2Pp = d2pv01[pi] if with_hessian else None
        else:
            Pp, dPp, d2Pp = 0.0, None, None
        v = 1.0 + r * a
        d = (1.0 - r * Pp) / v if pi >= 0 else 1.0 / v

        dd = np.zeros(P) if dPp is None else (-r / v) * dPp
        dd[s] -= (Pp + d * a) / v
        dfs[i] = d
        jac[i] = dd
        pv01[i] = Pp + a * d
        dpv01[i] = a * dd if dPp is None else dPp + a * dd

        if with_hessian:
            d2d = np.zeros((P, P)) if d2Pp is None else (-r / v) * d2Pp
            cross = dd * (a / v)
            if dPp is not None:
                cross = cross + dPp / v
            d2d[s, :] -= cross
            d2d[:, s] -= cross
            hess[i] = d2d
            d2pv01[i] = a * d2d if d2Pp is None else d2Pp + a * d2d

    return EngineCurve(times=times, dfs=dfs, jac=jac, hess=hess, key_collisions=collisions,
                       acc=acc, pillar=pillar, prev_idx=prev_idx, rates=rates)
