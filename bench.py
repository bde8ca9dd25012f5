#!/usr/bin/env python3
"""Throughput benchmark of the OIS PV + delta + gamma path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` prints ONE JSON line on rank 0.
For N > 1 it runs one rank per GPU over RCCL: either the driver launches it under ``torch.distributed.run``
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or - called plainly with ``--gpus N`` - it starts
those N ranks itself as a child ``torch.distributed.run`` BEFORE anything touches the GPU and relays the child's
output and exit code (a process that has initialised HIP is never re-exec'ed).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): per GPU a synthetic
portfolio of 1,000,000 spot-starting OIS on the README 32-pillar GBP SONIA curve (SURVEY.md section 8(d):
maturity U{1..360} months, annual ACT/365F legs with a front stub, coupon U(1%,7%), notional
round(U(1e6,5e7),-5), pay/receive 50/50, seed 20240430); one step = PV, 32-pillar delta ladder and
full 32x32 gamma of every trade written to HBM, plus the portfolio aggregate, all-reduced over the ranks.
Inputs are resident in HBM before the timed region; weak scaling: ONE portfolio of N x 1,000,000 trades is cut
into N contiguous shards of near-equal cash-flow count (adrates_amd/distributed.py), one per rank.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--trades", type=int, default=1_000_000, help="trades per GPU")
    ap.add_argument("--kind", default="offgrid", choices=["offgrid", "ongrid"])
    ap.add_argument("--interp", default="LINEAR_ZERO_RATES", choices=["LINEAR_ZERO_RATES", "FLAT_FWD_RATES", "LINEAR_FWD_RATES"])
    ap.add_argument("--requests", default="value,delta,gamma")
    ap.add_argument("--xccy-swaps", type=int, default=0,
                    help="BASELINE configs[4]: add a book of this many GBP/USD basis swaps per GPU to every step "
                         "(three more launches) and all-reduce all aggregate ladders in one buffer; 0 = the headline "
                         "OIS workload only")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0,
                    help="approximate CPU time budget of the baseline leg (0 disables it)")
    ap.add_argument("--min-warmup-ms", type=float, default=200.0,
                    help="keep running untimed warm-up steps beyond --warmup until this much wall time has passed: "
                         "the GPU clock needs ~100 ms of work to ramp, and W short steps (0.3 ms each for a "
                         "delta-only request) would leave the timed region on a cold clock")
    ap.add_argument("--aggregate-only", action="store_true",
                    help="Portfolio.compute's request: the book's ladder [pv, delta, gamma] and NO per-trade output "
                         "(adr_price_dev with agg only: knot-space sums + one projection per launch); reported with "
                         '"mode": "aggregate_only"')
    ap.add_argument("--collective", default="torch", choices=["torch", "native", "allgather"],
                    help="the one exchange step for N > 1: torch = torch.distributed all_reduce (RCCL), asynchronous and "
                         "double-buffered; native = adr_allreduce_agg (ncclAllReduce on the launch stream, communicator from "
                         "adr_rccl_comm_init); allgather = canonical chunks priced one by one, chunk ladders all-gathered and "
                         "summed in chunk order: the same bits on any number of ranks (adrates_amd/distributed.py)")
    ap.add_argument("--print-spawn-command", action="store_true",
                    help="with --gpus N > 1 and no WORLD_SIZE: print the launcher command instead of running it")
    return ap.parse_args(argv)


def spawn_ranks(args, argv):
    """``python bench.py --gpus N`` without a launcher: start the N ranks as a child process group.  Nothing in
    this process has touched the GPU yet (`torch.cuda.device_count()` does not initialise HIP on this image), and
    the child is a fresh interpreter, so no GPU-holding process is ever replaced."""
    import subprocess
    import torch
    have = torch.cuda.device_count()
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + \
          [a for a in argv if a != "--print-spawn-command"]
    if args.print_spawn_command:
        print(" ".join(cmd), flush=True)
        return 0
    if have < args.gpus and os.environ.get("ADR_BENCH_REHEARSE_ONE_GPU") != "1":
        print(f"bench.py: --gpus {args.gpus} requested but only {have} HIP device(s) are visible", file=sys.stderr)
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(curve, value_dt, interp_value, want_gamma, budget_s):
    """Time the CPU oracle (oracle/ - the build's restatement of the reference algorithm) on a bounded
    sample of the same synthetic workload.  Reported beside the GPU number; never the target."""
    try:
        from oracle import port as cpu_port
    except Exception as exc:  # the C restatement is optional test infrastructure
        return {"value": None, "unit": "trades/s", "cores": 0, "kind": "port", "sample": f"unavailable: {exc}"}
    return cpu_port.timed_baseline(curve, value_dt, interp_value, want_gamma, budget_s)


def cpu_baseline_b0(curve, value_dt, interp_value, want_gamma, budget_s):
    """the single-thread, one-trade-at-a-time loop over the autodiff restatement."""
    try:
        from oracle import port as cpu_port
        return cpu_port.timed_baseline_b0(curve, value_dt, interp_value, want_gamma, budget_s)
    except Exception as exc:
        return {"value": None, "unit": "trades/s", "cores": 0, "kind": "port", "sample": f"unavailable: {exc}"}


def parity_spot_check(host_curve, interp_value, batch, pv, delta, gamma, n_sample=1024):
    """After the timed region: ~1 000 trades of the batch that was timed (four contiguous runs spread over the batch,
    i.e. over the kernel's blocks) against oracle/port.c on the same inputs.  The oracle only checks here."""
    import numpy as np
    from oracle import port as cpu_port
    n = batch.n_trades
    run = max(1, min(n, n_sample) // 4)
    starts = sorted({min(max(0, n - run), (n * k) // 4) for k in range(4)})
    worst, checked = 0.0, 0
    for lo in starts:
        hi = min(n, lo + run)
        sub = batch.slice(lo, hi)
        ref = cpu_port.price(interp_value, host_curve.times, host_curve.dfs, host_curve.jac, host_curve.hess, sub,
                             want_delta=delta is not None, want_gamma=gamma is not None, n_threads=4)
        got = {"pv": pv[lo:hi].cpu().numpy(),
               "delta": None if delta is None else delta[lo:hi].cpu().numpy(),
               "gamma": None if gamma is None else gamma[lo:hi].cpu().numpy()}
        worst = max(worst, cpu_port.parity_error(got, ref, sub.notional))
        checked += hi - lo
    return {"max_error": worst, "trades": checked, "tolerance": 1e-10, "ok": bool(worst <= 1e-10),
            "against": "oracle/port.c (C restatement of the reference algorithm), metric of tests/_parity.py, "
                       "computed after the timed region on the timed batch's own outputs"}


def aggregate_spot_check(ctx, dev_curve, host_curve, interp_value, batch, mask, want_delta, want_gamma, n_sample=4096):
    """Aggregate-only mode: a contiguous book of ~4 000 trades of the timed batch priced the same way (ladder only) against
    the sums of oracle/port.c's per-trade ladders; error relative to the sum of absolute per-trade entries."""
    import numpy as np
    from adrates_amd import _native
    from oracle import port as cpu_port
    n = batch.n_trades
    lo = max(0, n // 2 - n_sample // 2)
    sub = batch.slice(lo, min(n, lo + n_sample))
    ref = cpu_port.price(interp_value, host_curve.times, host_curve.dfs, host_curve.jac, host_curve.hess, sub,
                         want_delta=want_delta, want_gamma=want_gamma, n_threads=4)
    dts = _native.DeviceTrades(ctx, sub)
    got = _native.price(ctx, dev_curve, dts, want_delta=want_delta, want_gamma=want_gamma, per_trade=False, aggregate=True)
    dts.close()
    worst = 0.0
    for key, r in (("agg_pv", ref["pv"]), ("agg_delta", ref.get("delta")), ("agg_gamma", ref.get("gamma"))):
        if r is None:
            continue
        scale = float(np.max(np.abs(r).sum(0))) if r.ndim > 1 else float(np.abs(r).sum())
        worst = max(worst, float(np.max(np.abs(np.asarray(got[key]) - r.sum(0)))) / max(scale, 1e-300))
    return {"max_error": worst, "trades": sub.n_trades, "tolerance": 1e-10, "ok": bool(worst <= 1e-10),
            "against": "sums of oracle/port.c's per-trade ladders over a contiguous book cut from the timed batch, priced "
                       "ladder-only like the timed launches; error relative to the sum of absolute per-trade entries"}


def measured_fp64(n, want_gamma, kind, interp, kern_ms):
    """fp64 vector-ALU figures of the dominant kernel (the gamma configuration sits within ~2x
    of the FMA bound): wave-instruction counts per trade from the committed PMC pass of this same command
    (tools/pmc.sh -> profiles/r*_final_fp64.json), turned into TFLOP/s with THIS run's kernel time.  Peak: 78.6 TFLOP/s
    fp64 vector (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz, MI355X_MICROARCH.md)."""
    import glob
    if not (want_gamma and kind == "offgrid" and interp == "LINEAR_ZERO_RATES"):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_fp64.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            c = json.load(f)
        stale = profile_is_stale(c)
        if stale:
            return {"source": f"stale: profiles/{os.path.basename(files[-1])} {stale}"}
        per_trade = {k: float(c["per_trade"][k]) for k in ("fma_f64", "mul_f64", "add_f64", "trans_f64", "valu_total")}
        flops = 64.0 * (2.0 * per_trade["fma_f64"] + per_trade["mul_f64"] + per_trade["add_f64"] + per_trade["trans_f64"])
        tf = flops * n / (kern_ms * 1e-3) / 1e12
        return {"wave_instructions_per_trade": per_trade, "flop_per_trade": flops, "achieved_tflops": tf,
                "peak_tflops": 78.6, "frac": tf / 78.6,
                "valu_issue_frac": per_trade["valu_total"] * n * 4.0 / (kern_ms * 1e-3 * 2.4e9 * 1024),
                "source": f"profiles/{os.path.basename(files[-1])}: rocprofv3 SQ_INSTS_VALU_*_F64 pass of this command on an "
                          "earlier box (counts are per trade and box independent); time from this run"}
    except Exception:
        return None


class HostStagedAllReduce:
    """The rehearsal's stand-in for RCCL's asynchronous all-reduce (gloo reduces host tensors): `start` queues a D2H copy
    of the buffer into pinned memory behind the step's kernels; the collective itself is issued (async_op=True) once
    that copy has completed - checked when the NEXT step has been launched, so the host never idles the GPU - and `wait`
    completes it and copies the reduced ladder back before the buffer is priced into again.  Same `pending[j]` /
    two-buffer protocol as the RCCL branch: with two ranks both buffers' collectives are in flight across steps."""

    def __init__(self, dist, torch, stream, agg):
        self.dist, self.torch, self.stream, self.agg = dist, torch, stream, agg
        self.host = torch.empty(agg.shape, dtype=agg.dtype, pin_memory=True)
        self.copied = torch.cuda.Event()
        self.work = None
        self.staged = False

    def start(self):
        self.host.copy_(self.agg, non_blocking=True)          # on the launch stream, behind this step's kernels
        self.copied.record(self.stream)
        self.staged = True

    def issue(self):
        if self.staged and self.work is None:
            self.copied.synchronize()
            self.work = self.dist.all_reduce(self.host, async_op=True)

    def wait(self):
        self.issue()
        if self.work is not None:
            self.work.wait()
            self.work = None
            self.agg.copy_(self.host, non_blocking=True)      # back on the launch stream, ahead of the next pricing call
        self.staged = False


def profile_is_stale(summary):
    """'' when the committed counter summary was taken on the sources the running library is built from (their sha256,
    adrates_amd/_native.py::build_identity), else the reason: per-trade instruction and byte counts of another revision of
    the kernel must not be multiplied into this run's time."""
    from adrates_amd import _native
    have = summary.get("source_sha256")
    now = _native.build_identity()["source_sha256"]
    if not have:
        return "carries no source hash"
    if have != now:
        return f"was taken on sources {have[:12]}, the running library is built from {now[:12]}"
    return ""


def measured_traffic(n, want_gamma, kind, interp):
    """(HBM bytes per launch, where the figure comes from): the committed rocprofv3 PMC summary of this same
    command (tools/profile.sh + tools/profile_summary.py -> profiles/*_traffic.json), or (None, reason) when the
    workload differs from the profiled one.  bench.py cannot run the profiler on itself, so this is a figure
    read from a file, not measured in this run - `traffic_source` in the JSON line says so."""
    import glob
    if not (n == 1_000_000 and want_gamma and kind == "offgrid" and interp == "LINEAR_ZERO_RATES"):
        return None, "none: no committed PMC pass for this workload"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_traffic.json")))
    if not files:
        return None, "none: profiles/r*_final_traffic.json missing"
    try:
        with open(files[-1]) as f:
            c = json.load(f)
        stale = profile_is_stale(c)
        if stale:
            return None, f"stale: profiles/{os.path.basename(files[-1])} {stale}"
        return float(c["hbm_bytes_per_launch"]), (
                f"profiles/{os.path.basename(files[-1])}: rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this command on "
                f"an earlier box (sources {c['source_sha256'][:12]} = the running library's), read from the file - not "
                "measured in this run")
    except Exception as exc:
        return None, f"none: {exc}"


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args, argv))      # this process never touches the GPU
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) "
                         "or call bench.py without a launcher")
    # ADR_BENCH_REHEARSE_ONE_GPU=1: every rank uses device 0 and the collectives run over gloo (RCCL refuses two
    # ranks on one device) - a one-GPU box can then walk the whole N > 1 path (rank spawn, sharding of the one
    # portfolio, all-reduce of the ladders, max-over-ranks timing).  The line it prints is NOT a scaling result.
    rehearse = os.environ.get("ADR_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    # ADR_BENCH_FORCE_DIST=1 runs the process-group code path with a single rank (a one-GPU box can rehearse
    # the collective calls the N > 1 runs make; needs MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE in the env)
    use_dist = world > 1 or os.environ.get("ADR_BENCH_FORCE_DIST") == "1"
    if use_dist:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from adrates_amd import _native
    from adrates_amd.market.curves.curve_tables import build_engine_curve
    from adrates_amd.trades import synthetic
    from adrates_amd.utils import InterpTypes
    from adrates_amd.trades.market_data import README_VALUE_DT, gbp_model

    agg_only = args.aggregate_only
    chunked = args.collective == "allgather"
    if chunked and args.xccy_swaps > 0:
        raise SystemExit("--collective allgather prices the OIS portfolio in canonical chunks; not with --xccy-swaps")
    if args.collective == "native" and rehearse and world > 1:
        raise SystemExit("--collective native needs one GPU per rank (RCCL refuses two ranks on one device)")
    reqs = {r.strip().lower() for r in args.requests.split(",")}
    want_gamma = "gamma" in reqs
    want_delta = want_gamma or "delta" in reqs
    mask = 1 | (2 if want_delta else 0) | (4 if want_gamma else 0)

    interp = InterpTypes[args.interp]
    model = gbp_model(README_VALUE_DT, interp)
    curve = model.curves.GBP_OIS_SONIA
    host_curve = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    P = host_curve.n_pillars

    ctx = _native.Context(local_rank)
    dev_curve = _native.DeviceCurve(ctx, interp.value, host_curve.times, host_curve.dfs, host_curve.jac,
                                    host_curve.hess)
    # ONE portfolio of world x --trades trades; this rank compiles and uploads its contiguous shard only (the cut
    # points are those of distributed.shard_batch: near-equal cash-flow counts)
    n_total = world * args.trades
    chunks = []          # --collective allgather: this rank's canonical chunks (trade ranges of `batch`)
    batch, (lo, hi) = synthetic.shard_of_portfolio(README_VALUE_DT, n_total, rank, world, kind=args.kind,
                                                   **(dict(canonical_chunks=True, chunks_out=chunks) if chunked else {}))
    n = batch.n_trades
    torch.cuda.synchronize()
    t_up = time.perf_counter()
    if chunked:          # one device batch (and one aggregate ladder) per canonical chunk
        pieces = [(_native.DeviceTrades(ctx, batch.slice(a, b)), a) for a, b in chunks]
        dev_trades = None
        in_bytes = sum(p[0].input_bytes for p in pieces)
    else:
        dev_trades = _native.DeviceTrades(ctx, batch)   # host-side row tables + H2D copies (blocking): reported apart
        pieces = [(dev_trades, 0)]
        in_bytes = dev_trades.input_bytes
    upload_ms = (time.perf_counter() - t_up) * 1e3
    out_bytes = 8 * (1 + P + P * P) * len(pieces) if agg_only else \
        8 * n * (1 + (P if want_delta else 0) + (P * P if want_gamma else 0))
    curve_bytes = 16 * host_curve.n_knots + 8 * host_curve.n_knots * P * (1 + (P if want_gamma else 0))
    algo_bytes = in_bytes + out_bytes + curve_bytes

    dev = torch.device("cuda", local_rank)
    pv = torch.empty(n, dtype=torch.float64, device=dev) if not agg_only else None
    delta = torch.empty((n, P), dtype=torch.float64, device=dev) if (want_delta and not agg_only) else None
    gamma = torch.empty((n, P, P), dtype=torch.float64, device=dev) if (want_gamma and not agg_only) else None

    # optional mixed book (BASELINE configs[4]): a cross-currency book per GPU next to the OIS portfolio
    xccy = []          # [(device trades, device curve, pv, delta, gamma, offset of its aggregate in `agg`)]
    n_x = args.xccy_swaps
    agg_len = 1 + P + P * P
    if n_x > 0:
        from adrates_amd.market.position.engine import Engine
        from adrates_amd.trades import synthetic_xccy
        from adrates_amd.trades.market_data import GBP_PX, TENORS, USD_PX
        market = synthetic_xccy.build_market(README_VALUE_DT, GBP_PX, USD_PX, TENORS)
        _native.set_default_context(ctx)                # the engine uploads the book's curves through this rank's context
        parts, _ = synthetic_xccy.synthesize_book(Engine(market), README_VALUE_DT, world * n_x,
                                                  seed=synthetic.DEFAULT_SEED + 1000, rank=rank, world_size=world)
        n_x = parts[0][0].n_trades                      # this rank's share of the world x --xccy-swaps book
        for b, cur in parts:
            Px = cur.n_pillars
            xccy.append((_native.DeviceTrades(ctx, b), cur, torch.empty(n_x, dtype=torch.float64, device=dev) if not agg_only else None,
                         torch.empty((n_x, Px), dtype=torch.float64, device=dev) if (want_delta and not agg_only) else None,
                         torch.empty((n_x, Px, Px), dtype=torch.float64, device=dev) if (want_gamma and not agg_only) else None, agg_len))
            agg_len += 1 + Px + Px * Px
    # one buffer for every aggregate ladder of the step: a single all-reduce whatever the book holds.  Two of them,
    # used alternately: step k's all-reduce runs on RCCL's stream while step k + 1 prices into the other buffer
    # (the collective is 8-28 KB and latency-bound: serialised it would idle the GPU for its whole round trip)
    aggs = [torch.zeros(agg_len, dtype=torch.float64, device=dev) for _ in range(2)]
    pending = [None, None]              # the all-reduce still reading / writing aggs[j]
    step_no = [0]
    from adrates_amd import distributed as D
    chunk_aggs = torch.zeros((len(pieces), agg_len), dtype=torch.float64, device=dev) if chunked else None
    gathered = torch.zeros((D.CANONICAL_CHUNKS, agg_len), dtype=torch.float64, device=dev) if chunked else None
    native_comm = None
    if args.collective == "native" and use_dist:
        import ctypes as C
        lib = _native.load()
        uid = (C.c_ubyte * 128)()
        if rank == 0:
            _native._check(lib.adr_rccl_unique_id(uid), "adr_rccl_unique_id")
        box = [bytes(uid)]
        dist.broadcast_object_list(box, src=0)          # the id travels over the process group that already exists
        native_comm = C.c_void_p()
        _native._check(lib.adr_rccl_comm_init(ctx._h, box[0], world, rank, C.byref(native_comm)), "adr_rccl_comm_init")

    def ptr(t, offset_elems=0):
        return 0 if t is None else t.data_ptr() + 8 * offset_elems

    def price_pieces(agg):
        for k, (trades_k, first) in enumerate(pieces):
            _native.price_dev(ctx, dev_curve, trades_k, mask, ptr(pv, first), ptr(delta, first * P), ptr(gamma, first * P * P),
                              chunk_aggs[k].data_ptr() if chunked else agg.data_ptr(), stream.cuda_stream)

    def price_xccy(agg):
        for trades_x, cur, pv_x, de_x, ga_x, off in xccy:
            _native.price_dev(ctx, cur, trades_x, mask, pv_x.data_ptr() if pv_x is not None else 0,
                              de_x.data_ptr() if de_x is not None else 0,
                              ga_x.data_ptr() if ga_x is not None else 0, agg.data_ptr() + 8 * off, stream.cuda_stream)
    # a non-default torch stream: the kernels, the HIP events that time them and the RCCL all-reduce all
    # go to this one stream (torch.cuda.Event only sees the stream it is recorded on)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)

    staged = [HostStagedAllReduce(dist, torch, stream, a) for a in aggs] if (use_dist and rehearse) else None

    def reduce_small(t, op=None):
        """Blocking all-reduce of a small device tensor (timings, counts): over RCCL, or through the host for gloo."""
        op = op or dist.ReduceOp.SUM
        if rehearse:
            h = t.detach().cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def step(events=None):
        j = step_no[0] & 1
        step_no[0] += 1
        agg = aggs[j]
        if pending[j] is not None:          # the launch stream waits for the collective that last used this buffer
            pending[j].wait()
            pending[j] = None
        if events is not None:
            events[0].record(stream)
        price_pieces(agg)
        if events is not None:
            events[1].record(stream)
        price_xccy(agg)
        if chunked:
            # chunk ladders -> the book ladder, the same additions whatever the world size (distributed.py)
            if use_dist and rehearse:
                host_parts = [torch.empty((len(pieces), agg_len), dtype=torch.float64) for _ in range(world)]
                dist.all_gather(host_parts, chunk_aggs.cpu())
                gathered.copy_(torch.cat(host_parts, dim=0))
            elif use_dist:
                dist.all_gather_into_tensor(gathered, chunk_aggs)
            else:
                gathered.copy_(chunk_aggs)
            torch.sum(gathered, dim=0, out=agg)
        elif native_comm is not None:
            # the library's own exchange step: ncclAllReduce of the ladder on the launch stream, behind this step's kernels
            _native._check(_native.load().adr_allreduce_agg(ctx._h, native_comm, agg.data_ptr(), agg_len, stream.cuda_stream),
                           "adr_allreduce_agg")
        elif use_dist:
            # the one exchange step: 1 + P + P*P doubles (per curve) over RCCL/xGMI, started behind this step's kernels
            if rehearse:
                staged[j].start()
                pending[j] = staged[j]
                if pending[j ^ 1] is not None:
                    pending[j ^ 1].issue()      # the previous step's ladder has reached the host by now: reduce it
            else:
                pending[j] = dist.all_reduce(agg, async_op=True)

    def fence():
        for j in range(2):
            if pending[j] is not None:
                pending[j].wait()
                pending[j] = None
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    warm_steps = args.warmup
    t_warm = time.perf_counter()
    while (time.perf_counter() - t_warm) * 1e3 < args.min_warmup_ms:     # untimed, like the W steps above
        for _ in range(10):
            step()
        fence()
        warm_steps += 10

    # Device time of the pricing launches from HIP events on the launch stream.  One rank: ONE pair around the whole
    # timed region (a pair per step adds an event barrier in front of every launch, ~50 us of idle GPU per step -
    # nothing against the 2.4 ms headline step, 15 % of a 0.35 ms delta-only step).  Several ranks: a pair per step
    # around the pricing call, so that the all-reduce stays outside the kernel figure.
    per_step = use_dist or n_x > 0          # (the mixed book: the figure stays the OIS launch alone)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps if per_step else 1)]
    t0 = time.perf_counter()
    if not per_step:
        ev[0][0].record(stream)
    for i in range(args.steps):
        step(ev[i] if per_step else None)
    if not per_step:
        ev[0][1].record(stream)
    fence()
    elapsed = time.perf_counter() - t0

    if per_step:
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    else:
        kern_ms = ev[0][0].elapsed_time(ev[0][1]) / args.steps
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        reduce_small(t, dist.ReduceOp.MAX)
    elapsed = float(t.item())
    counts = torch.tensor([n, n_x], dtype=torch.int64, device=dev)
    if use_dist:
        reduce_small(counts)                             # units all ranks processed per step
    # the reduced ladder of the LAST timed step against the per-rank ladders, gathered once and summed in rank order
    allreduce_check = None
    if use_dist and chunked:
        # every rank formed the book ladder from the same gathered chunk ladders: they must agree bit for bit
        mine = aggs[(step_no[0] - 1) & 1].detach().cpu()
        everyone = [torch.zeros(agg_len, dtype=torch.float64) for _ in range(world)]
        if rehearse:
            dist.all_gather(everyone, mine)
        else:
            dev_all = [torch.zeros(agg_len, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(dev_all, mine.to(dev))
            everyone = [t.cpu() for t in dev_all]
        same = all(torch.equal(everyone[0], t) for t in everyone[1:])
        allreduce_check = {"status": "ok" if same else "MISMATCH", "max_rel_error": 0.0 if same else float("nan"), "ranks": world,
                           "what": "the book ladder every rank formed from the all-gathered canonical chunk ladders, compared "
                                   "bit for bit across the ranks"}
    elif use_dist:
        reduced = aggs[(step_no[0] - 1) & 1].detach().cpu()
        local = torch.zeros(agg_len, dtype=torch.float64, device=dev)
        price_pieces(local)
        price_xccy(local)
        stream.synchronize()
        parts = [torch.zeros(agg_len, dtype=torch.float64) for _ in range(world)]
        if rehearse:
            dist.all_gather(parts, local.cpu())
        else:
            dev_parts = [torch.zeros(agg_len, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(dev_parts, local)
            parts = [p.cpu() for p in dev_parts]
        total = parts[0].clone()
        for p_ in parts[1:]:
            total += p_
        err = float((reduced - total).abs().max() / max(float(total.abs().max()), 1e-300))
        allreduce_check = {"status": "ok" if err <= 1e-12 else "MISMATCH", "max_rel_error": err, "ranks": world,
                           "what": "all-reduced aggregate of the last timed step vs the sum, in rank order, of the "
                                   "per-rank aggregates gathered once after the timed region"}
    n_all, nx_all = int(counts[0].item()), int(counts[1].item())

    spot_ok = True
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = (n_all + nx_all) * args.steps / elapsed
        plain = not agg_only and not chunked           # the committed counter passes are of the plain bench command
        traffic, traffic_source = measured_traffic(n, want_gamma, args.kind, args.interp) if plain else \
            (None, "none: no committed PMC pass for this mode")
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "OIS trades/sec PV+delta+gamma, 32-pillar curve; achieved HBM GB/s",
            "value": value, "unit": "trades/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_steps_run": warm_steps,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if not rehearse else "synthetic; REHEARSAL: all ranks on device 0, gloo collectives - not a scaling result",
            "config": {"workload": f"{args.trades} random-tenor OIS per GPU ({args.kind}), PV + {P}-pillar delta"
                                   + (f" + full {P}x{P} gamma" if want_gamma else "")
                                   + f", {args.interp}, README GBP SONIA curve (BASELINE configs[2])",
                       "trades_per_gpu": args.trades, "trades_total": n_all, "rank0_trades": n,
                       "pillars": P, "knots": host_curve.n_knots,
                       "requests": sorted(reqs), "parallelism": f"one portfolio cut into {world} contiguous shards of equal "
                                                               f"cash-flow count, RCCL all-reduce of {agg_len} doubles",
                       "collective": {"torch": "torch.distributed all_reduce (RCCL), asynchronous, double-buffered",
                                      "native": "adr_allreduce_agg (ncclAllReduce on the launch stream)",
                                      "allgather": f"{D.CANONICAL_CHUNKS} canonical chunks, {len(pieces)} priced by rank 0 one by one; "
                                                   "chunk ladders all-gathered, summed in chunk order"}[args.collective]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms": kern_ms, "kernel": ("rank 0's pricing launches, HIP events on the launch stream around "
                                    + ("each step's pricing call" if per_step else "the whole timed region, divided by steps")), "algorithmic_bytes_per_launch": algo_bytes,
                         "algorithmic_bytes_per_trade": algo_bytes / n},
        }
        line["build"] = _native.build_identity()
        if agg_only:
            # Portfolio.compute's request (cavour/market/portfolio/portfolio.py:39-66): the book ladder alone.  The pass reads
            # the trades once and writes 1 + P + P*P doubles: its HBM roofline is the input stream; what bounds it is the
            # vector-ALU work of the lookups and exponentials, like the PV-only pass.
            line["mode"] = "aggregate_only"
            line["config"]["workload"] += "; AGGREGATE ONLY: the book's ladder, no per-trade output (knot-space sums, one projection per launch)"
            line["roofline"]["note"] = ("algorithmic bytes = the trade inputs + curve tables + one ladder per launch; the kernel "
                                        "is bound by vector-ALU issue (three lookups and exponentials per coupon), not by HBM")
        if chunked:
            import hashlib
            line["aggregate_sha256"] = hashlib.sha256(aggs[(step_no[0] - 1) & 1].detach().cpu().numpy().tobytes()).hexdigest()
        if n_x > 0:
            line["metric"] = "OIS trades + XCCY swaps per second, PV+delta+gamma, aggregate ladders all-reduced"
            line["config"]["workload"] += (f" + {args.xccy_swaps} GBP/USD basis swaps per GPU with SONIA / SOFR / basis ladders "
                                           f"(BASELINE configs[4]; the roofline object is the OIS kernel alone)")
            line["config"]["xccy_swaps_per_gpu"] = args.xccy_swaps
            line["config"]["xccy_swaps_total"] = nx_all
            line["config"]["allreduce_doubles"] = agg_len
        line["upload_ms"] = {"trades": upload_ms, "what": "adr_trades_upload of rank 0's batch: host-side construction of "
                             "the row tables + H2D copies, blocking, once per portfolio; outside `value`",
                             "input_bytes": in_bytes}
        if allreduce_check is not None:
            line["allreduce_check"] = allreduce_check
            spot_ok = spot_ok and allreduce_check["status"] == "ok"
        fp64 = measured_fp64(n, want_gamma, args.kind, args.interp, kern_ms) if plain else None
        if fp64 is not None:
            line["roofline"]["fp64_valu"] = fp64
        if n_x == 0 and not agg_only:
            check = parity_spot_check(host_curve, interp.value, batch, pv, delta, gamma)
            line["parity_spot_check"] = check
            spot_ok = spot_ok and check["ok"]
        if n_x == 0 and agg_only:
            check = aggregate_spot_check(ctx, dev_curve, host_curve, interp.value, batch, mask, want_delta, want_gamma)
            line["parity_spot_check"] = check
            spot_ok = spot_ok and check["ok"]
        if args.cpu_baseline_seconds > 0 and world == 1 and n_x == 0:
            line["cpu_baseline"] = cpu_baseline(curve, README_VALUE_DT, interp.value, want_gamma,
                                                args.cpu_baseline_seconds)
            line["cpu_baseline_b0"] = cpu_baseline_b0(curve, README_VALUE_DT, interp.value, want_gamma,
                                                      min(6.0, args.cpu_baseline_seconds))
        print(json.dumps(line), flush=True)

    if native_comm is not None:
        _native.load().adr_rccl_comm_destroy(native_comm)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0 and not spot_ok:
        raise SystemExit("bench.py: parity_spot_check (timed batch vs oracle, 1e-10) or allreduce_check failed - see the JSON line")


if __name__ == "__main__":
    main()
