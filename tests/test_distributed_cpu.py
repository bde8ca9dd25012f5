"""N > 1 path on CPU: two gloo ranks shard the trade axis, price their block (with the C oracle standing
in for the GPU kernels - this test exercises the sharding and the collective, not the kernels) and
all-reduce the aggregate ladder; the result must equal the single-process aggregate."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from adrates_amd import distributed as D
from adrates_amd.trades import synthetic

from . import _fixtures as F


def test_shard_bounds_balance_cashflows():
    b = synthetic.synthesize(F.README_VALUE_DT, 5000, seed=9)
    for world in (1, 2, 3, 8):
        bounds = D.shard_bounds(b.flt_off, b.fix_off, world)
        assert bounds[0][0] == 0 and bounds[-1][1] == b.n_trades
        assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
        flows = [int(b.flt_off[hi] - b.flt_off[lo]) for lo, hi in bounds]
        assert max(flows) - min(flows) <= 2 * 30 + 2
    tiny = synthetic.synthesize(F.README_VALUE_DT, 3, seed=9)
    bounds = D.shard_bounds(tiny.flt_off, tiny.fix_off, 8)          # more ranks than trades
    assert sum(hi - lo for lo, hi in bounds) == 3 and all(hi >= lo for lo, hi in bounds)


def _worker(rank, world, port_no, n, out_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_no))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cavour_oracle as O
    from oracle import port
    curve = F.readme_model().curves.GBP_OIS_SONIA
    cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    batch = synthetic.synthesize(F.README_VALUE_DT, n, seed=21)
    mine, (lo, hi) = D.shard_batch(batch, rank, world)
    r = port.price(4, cache["times"], cache["dfs"], cache["jac"], cache["hess"], mine, n_threads=2)
    P = cache["jac"].shape[1]
    agg = torch.zeros(1 + P + P * P, dtype=torch.float64)
    agg[0] = float(r["pv"].sum())
    agg[1:1 + P] = torch.from_numpy(r["delta"].sum(0))
    agg[1 + P:] = torch.from_numpy(r["gamma"].sum(0).reshape(-1))
    agg2 = agg.clone()
    D.allreduce_aggregate(agg)
    D.allgather_sum_fixed_order(agg2)
    if rank == 0:
        np.save(out_path, np.stack([agg.numpy(), agg2.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_aggregate_equals_single_process(tmp_path):
    from oracle import cavour_oracle as O
    from oracle import port
    n = 4000
    out = str(tmp_path / "agg.npy")
    port_no = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port_no, n, out), nprocs=2, join=True)
    got = np.load(out)
    curve = F.readme_model().curves.GBP_OIS_SONIA
    cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    batch = synthetic.synthesize(F.README_VALUE_DT, n, seed=21)
    r = port.price(4, cache["times"], cache["dfs"], cache["jac"], cache["hess"], batch)
    want = np.concatenate([[r["pv"].sum()], r["delta"].sum(0), r["gamma"].sum(0).reshape(-1)])
    for row in got:
        assert np.allclose(row, want, rtol=1e-11, atol=1e-9 * np.abs(want).max())


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: a mixed OIS / cross-currency book sharded over the ranks, one all-reduce for all of its
# aggregate ladders.  The C port stands in for the kernels (as above, and as in tests/test_xccy_engine_host.py).
# ---------------------------------------------------------------------------------------------------------------
def _mixed_book():
    from tests.test_gpu_xccy import VALUE_DT, _swap
    from adrates_amd.utils import FrequencyTypes
    swaps = []
    for years in (1, 2, 3, 5, 7, 10, 12, 15):
        for back, freq in ((0, FrequencyTypes.ANNUAL), (5, FrequencyTypes.SEMI_ANNUAL)):
            eff = VALUE_DT.add_months(-back)
            swaps.append(_swap(eff.add_months(12 * years + back), 0.0030 + 0.0001 * years, effective=eff, freq=freq,
                               notional=1e6 * (1 + years)))
    return swaps


def _mixed_aggregates(rank, world):
    """Aggregates of this rank's share of the book as one dict (keys agg_*)."""
    from unittest import mock
    from adrates_amd import _native
    from adrates_amd.market.position import xccy_engine
    from adrates_amd.market.position.engine import Engine
    from adrates_amd.utils import RequestTypes
    from oracle import cavour_oracle as O
    from oracle import port
    from tests.test_gpu_xccy import VALUE_DT, _model
    from tests.test_xccy_engine_host import _HostCurve, _HostTrades, _host_curve_df, _host_price
    m = _model()
    gbp = m.curves.GBP_OIS_SONIA
    cache = O.cached_curve(gbp.swap_rates, gbp.swap_times, gbp.year_fracs)
    ois = synthetic.synthesize(VALUE_DT, 600, seed=4)
    mine, _ = D.shard_batch(ois, rank, world)
    r = port.price(gbp._interp_type.value, cache["times"], cache["dfs"], cache["jac"], cache["hess"], mine, n_threads=2)
    out = dict(agg_ois_pv=float(r["pv"].sum()), agg_ois_delta=r["delta"].sum(0), agg_ois_gamma=r["gamma"].sum(0))
    book = _mixed_book()
    lo, hi = D.shard_by_work([len(s._foreign_leg._payment_dts) + len(s._domestic_leg._payment_dts) for s in book], world)[rank]
    with mock.patch.object(_native, "DeviceCurve", _HostCurve), mock.patch.object(_native, "DeviceTrades", _HostTrades), \
         mock.patch.object(_native, "price", _host_price), mock.patch.object(_native, "default_context", lambda: None), \
         mock.patch.object(_native, "curve_df", _host_curve_df):
        x = xccy_engine.price_xccy_batch(Engine(m), book[lo:hi], {RequestTypes.VALUE, RequestTypes.DELTA, RequestTypes.GAMMA},
                                         per_trade=False, aggregate=True)
    out.update({k: v for k, v in x.items() if k.startswith("agg_")})
    out["tenors"] = x["tenors"]
    return out


def _mixed_worker(rank, world, port_no, out_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_no))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    agg = D.allreduce_book(_mixed_aggregates(rank, world))
    if rank == 0:
        np.savez(out_path, **{k: v for k, v in agg.items() if k.startswith("agg_")})
    dist.barrier()
    dist.destroy_process_group()


def test_mixed_ois_xccy_book_two_ranks_one_allreduce(tmp_path):
    out = str(tmp_path / "mixed.npz")
    port_no = 31500 + (os.getpid() % 2000)
    mp.spawn(_mixed_worker, args=(2, port_no, out), nprocs=2, join=True)
    got = np.load(out)
    want = _mixed_aggregates(0, 1)
    keys = sorted(k for k in want if k.startswith("agg_"))
    assert sorted(got.files) == keys and len(keys) == 10      # 3 OIS + pv and 3 x (delta, gamma) of the XCCY book
    for k in keys:
        w = np.asarray(want[k])
        assert np.allclose(got[k], w, rtol=1e-11, atol=1e-9 * max(1.0, np.abs(w).max())), k
    assert abs(float(want["agg_pv"])) > 1.0 and np.abs(want["agg_delta_basis"]).max() > 1.0


def test_shard_by_work_and_book_allreduce_single_process():
    bounds = D.shard_by_work([5, 1, 1, 1, 5, 5, 2], 3)
    assert bounds[0][0] == 0 and bounds[-1][1] == 7 and all(bounds[i][1] == bounds[i + 1][0] for i in range(2))
    agg = dict(agg_pv=2.5, agg_delta=np.arange(4.0), tenors=("1Y",))
    same = D.allreduce_book(dict(agg))            # no process group: identity
    assert same["agg_pv"] == 2.5 and np.array_equal(same["agg_delta"], np.arange(4.0)) and same["tenors"] == ("1Y",)


# ---------------------------------------------------------------------------------------------------------------
# Aggregates that do not depend on the world size: canonical chunks, all-gather, fixed-order sum (distributed.py).
# ---------------------------------------------------------------------------------------------------------------
def _chunk_ladders(rank, world, n):
    from oracle import cavour_oracle as O
    from oracle import port
    curve = F.readme_model().curves.GBP_OIS_SONIA
    cache = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    batch = synthetic.synthesize(F.README_VALUE_DT, n, seed=33)
    P = cache["jac"].shape[1]
    rows = []
    for lo, hi in D.canonical_chunks(batch.flt_off, batch.fix_off, rank, world):
        r = port.price(4, cache["times"], cache["dfs"], cache["jac"], cache["hess"], batch.slice(lo, hi), n_threads=1)
        rows.append(np.concatenate([[r["pv"].sum()], r["delta"].sum(0), r["gamma"].sum(0).reshape(-1)]))
    return torch.from_numpy(np.stack(rows)), P


def _chunk_worker(rank, world, port_no, n, out_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_no))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, _ = _chunk_ladders(rank, world, n)
    total = D.allgather_chunks_fixed_order(rows)
    if rank == world - 1:            # (any rank: they all hold the same bits)
        np.save(out_path, total.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_canonical_chunks_give_the_same_bits_on_one_two_and_three_ranks(tmp_path):
    n = 2400
    rows, _ = _chunk_ladders(0, 1, n)
    assert rows.shape[0] == D.CANONICAL_CHUNKS
    single = D.allgather_chunks_fixed_order(rows).numpy()
    for world in (2, 3):
        out = str(tmp_path / f"chunks{world}.npy")
        mp.spawn(_chunk_worker, args=(world, 29500 + (os.getpid() + 7 * world) % 2000, n, out), nprocs=world, join=True)
        assert np.array_equal(np.load(out), single), world
    with pytest.raises(ValueError):
        D.canonical_chunks(np.arange(10), np.arange(10), 0, 5)          # 5 does not divide 24
