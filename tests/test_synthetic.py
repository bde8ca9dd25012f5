"""The vectorised portfolio synthesiser equals the `OIS(...)` API path bit for bit."""
import numpy as np
import pytest

from adrates_amd.trades import synthetic
from adrates_amd.trades.compiler import compile_ois

from . import _fixtures as F

FIELDS = ("fix_off", "flt_off", "fix_tp", "fix_pay", "flt_tp", "flt_ts", "flt_te", "flt_alpha", "notional",
          "spread", "fix_sign", "flt_sign")


@pytest.mark.parametrize("kind", ["offgrid", "ongrid"])
def test_synthesiser_equals_api(kind):
    vd = F.README_VALUE_DT
    fast = synthetic.synthesize(vd, 1000, kind=kind)
    slow = compile_ois(synthetic.swaps_from_terms(vd, *synthetic.draw_terms(1000, kind)), vd)
    for f in FIELDS:
        assert np.array_equal(getattr(fast, f), getattr(slow, f)), f
    m = np.diff(fast.flt_off)
    assert m.min() >= 1 and m.max() <= 30
    if kind == "offgrid":
        assert 14.5 < m.mean() < 16.5                       # SURVEY: mean 15.5 coupons


def test_slice_rebases_offsets():
    b = synthetic.synthesize(F.README_VALUE_DT, 50, seed=1)
    s = b.slice(10, 25)
    assert s.n_trades == 15 and s.fix_off[0] == 0 and s.flt_off[0] == 0
    assert np.array_equal(s.flt_tp, b.flt_tp[b.flt_off[10]:b.flt_off[25]])
    assert b.slice(7, 7).n_trades == 0


@pytest.mark.parametrize("world", [2, 3, 8])
def test_rank_shards_of_one_portfolio_equal_shard_batch(world):
    """bench.py's ranks compile only their own share of the global portfolio; the shares are exactly what
    `distributed.shard_batch` cuts from the whole (same bounds, same arrays), and together they are the whole."""
    from adrates_amd.distributed import shard_batch
    vd = F.README_VALUE_DT
    whole = synthetic.synthesize(vd, 4000, seed=5)
    covered = 0
    for rank in range(world):
        mine, (lo, hi) = synthetic.shard_of_portfolio(vd, 4000, rank, world, seed=5)
        want, bounds = shard_batch(whole, rank, world)
        assert (lo, hi) == bounds and lo == covered
        covered = hi
        for f in FIELDS:
            assert np.array_equal(getattr(mine, f), getattr(want, f)), f
    assert covered == 4000
