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
