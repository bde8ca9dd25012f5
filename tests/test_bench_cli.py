"""bench.py's launcher handling (no GPU needed): `python bench.py --gpus N` must never print a silent one-GPU
number for N > 1 - it starts N ranks itself before any GPU call, or says why it cannot."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_plain_call_with_several_gpus_reaches_the_rank_spawn():
    out = _run(["--gpus", "8", "--steps", "5", "--warmup", "2", "--print-spawn-command"])
    assert out.returncode == 0, out.stderr
    cmd = out.stdout.strip().split()
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    at = cmd.index(BENCH)
    assert cmd[at + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]      # the ranks get the same arguments


def test_plain_call_fails_loudly_without_enough_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        return          # a multi-GPU box would really start the ranks
    out = _run(["--gpus", "2", "--steps", "1"])
    assert out.returncode != 0 and "only" in out.stderr and "device" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]        # no bench line


def test_mismatched_world_size_is_rejected():
    out = _run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr
    out = _run(["--gpus", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr
