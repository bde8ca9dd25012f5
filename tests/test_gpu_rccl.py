"""The native exchange step: adr_allreduce_agg over an RCCL communicator.  One GPU is all a test box has, so the
communicator has a single rank (sum over one rank = identity); what is checked is the linkage, the argument
handling and that the reduction is enqueued on the caller's stream after the pricing kernels."""
import ctypes as C

import numpy as np
import pytest

from adrates_amd import _native
from adrates_amd.trades import synthetic

from . import _fixtures as F
from .test_gpu_parity_batch import _device_curve

pytestmark = pytest.mark.gpu


def test_allreduce_of_the_aggregate_ladder_single_rank(gpu_ctx):
    import torch
    rccl = C.CDLL("librccl.so")
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    try:
        vd = F.README_VALUE_DT
        curve = F.readme_model().curves.GBP_OIS_SONIA
        host, dc = _device_curve(gpu_ctx, curve)
        batch = synthetic.synthesize(vd, 3000, seed=9)
        dt = _native.DeviceTrades(gpu_ctx, batch)
        P = 32
        dev = torch.device("cuda", 0)
        agg = torch.zeros(1 + P + P * P, dtype=torch.float64, device=dev)
        stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(stream):
            _native.price_dev(gpu_ctx, dc, dt, 7, 0, 0, 0, agg.data_ptr(), stream.cuda_stream)
            rc = _native.load().adr_allreduce_agg(gpu_ctx._h, comm, C.c_void_p(agg.data_ptr()), agg.numel(),
                                                  C.c_void_p(stream.cuda_stream))
            assert rc == 0
        stream.synchronize()
        want = _native.price(gpu_ctx, dc, dt, per_trade=False, aggregate=True)
        got = agg.cpu().numpy()
        assert got[0] == want["agg_pv"]
        assert np.array_equal(got[1:1 + P], want["agg_delta"])
        assert np.array_equal(got[1 + P:].reshape(P, P), want["agg_gamma"])
        # argument checks
        assert _native.load().adr_allreduce_agg(gpu_ctx._h, None, C.c_void_p(agg.data_ptr()), agg.numel(), None) < 0
        assert b"adr_allreduce_agg" in _native.load().adr_last_error()
    finally:
        rccl.ncclCommDestroy(comm)


def test_bench_process_group_path_with_one_rank(tmp_path):
    """bench.py's N > 1 code path (init_process_group("nccl"), all_reduce of the aggregate on the launch stream,
    barrier, MAX over ranks) rehearsed with a single rank, and the JSON contract of its one output line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0", ADR_BENCH_FORCE_DIST="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--trades", "20000", "--steps", "3",
                          "--warmup", "1", "--cpu-baseline-seconds", "0"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert d["value"] > 1e6 and d["roofline"]["bound"] == "hbm" and 0.0 < d["roofline"]["frac"] < 1.0
    assert abs(d["value"] - 20000 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    # the round-3 keys: upload time apart from `value`, the in-run parity spot check, the RCCL all-reduce check
    assert d["upload_ms"]["trades"] > 0.0
    assert d["parity_spot_check"]["ok"] and d["parity_spot_check"]["trades"] >= 1000
    assert d["parity_spot_check"]["max_error"] <= 1e-10
    assert d["allreduce_check"]["status"] == "ok" and d["allreduce_check"]["ranks"] == 1


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher, the way the driver calls it, walked end to end on a one-GPU box:
    bench.py spawns its own two ranks (before touching the GPU), each compiles its share of the ONE portfolio, prices
    it, the aggregate ladders are all-reduced and rank 0 prints one line.  ADR_BENCH_REHEARSE_ONE_GPU=1 puts both
    ranks on device 0 and swaps RCCL (which refuses two ranks on one device) for gloo; everything else is the code
    of the real run.  The line is marked as a rehearsal - its value says nothing about scaling."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ADR_BENCH_REHEARSE_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--trades", "30000", "--steps", "3",
                          "--warmup", "1", "--xccy-swaps", "2000", "--cpu-baseline-seconds", "0"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and "REHEARSAL" in d["data"]
    cfg = d["config"]
    assert cfg["trades_total"] == 60000 and 0 < cfg["rank0_trades"] < 60000 and cfg["xccy_swaps_total"] == 4000
    assert d["value"] > 1e5
    # the async, double-buffered all-reduce protocol ran with two ranks (host-staged over gloo here, RCCL on a real
    # node) and the reduced ladder of the last step equals the sum of the two ranks' own ladders
    chk = d["allreduce_check"]
    assert chk["status"] == "ok" and chk["ranks"] == 2 and chk["max_rel_error"] <= 1e-12


def _run_bench(extra_args, env_extra, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    if "RANK" not in env_extra:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--min-warmup-ms", "0",
                          "--cpu-baseline-seconds", "0"] + extra_args, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_aggregate_only_mode():
    """`bench.py --aggregate-only`: Portfolio.compute's request as its own reported mode - the ladder alone, checked in the
    run against the sums of oracle/port.c's per-trade ladders."""
    d = _run_bench(["--trades", "50000", "--aggregate-only"], {})
    assert d["mode"] == "aggregate_only" and "AGGREGATE ONLY" in d["config"]["workload"]
    assert d["parity_spot_check"]["ok"] and d["parity_spot_check"]["max_error"] <= 1e-10
    assert d["value"] > 1e8 and d["roofline"]["traffic"] is None and len(d["build"]["source_sha256"]) == 64


def test_bench_native_collective_single_rank():
    """`--collective native`: the communicator from adr_rccl_unique_id / adr_rccl_comm_init (id broadcast over the process
    group) and adr_allreduce_agg on the launch stream - with one rank, which is all a one-GPU box can give RCCL."""
    import os
    d = _run_bench(["--trades", "20000", "--collective", "native"],
                   dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200), RANK="0", WORLD_SIZE="1",
                        LOCAL_RANK="0", ADR_BENCH_FORCE_DIST="1"))
    assert "adr_allreduce_agg" in d["config"]["collective"]
    assert d["allreduce_check"]["status"] == "ok" and d["parity_spot_check"]["ok"]


def test_bench_allgather_collective_is_bitwise_independent_of_the_world_size():
    """`--collective allgather`: 24 canonical chunks priced one by one, their ladders all-gathered and summed in chunk order.
    The SAME 60 000-trade portfolio on 1, 2 and 3 ranks (rehearsed on one GPU over gloo) gives the same bits."""
    sums = {}
    for world, per_rank in ((1, 60000), (2, 30000), (3, 20000)):
        env = {"ADR_BENCH_REHEARSE_ONE_GPU": "1"} if world > 1 else {}
        d = _run_bench(["--gpus", str(world), "--trades", str(per_rank), "--collective", "allgather", "--aggregate-only"], env)
        assert d["config"]["trades_total"] == 60000 and d["parity_spot_check"]["ok"]
        if world > 1:
            assert d["allreduce_check"]["status"] == "ok" and d["allreduce_check"]["ranks"] == world
        sums[world] = d["aggregate_sha256"]
    assert sums[1] == sums[2] == sums[3], sums
