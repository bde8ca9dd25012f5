#!/usr/bin/env python3
"""Parity sweep at scale (run on the GPU box): random curves x the three interpolation schemes x mixed portfolios
of 100 000 trades (tests/_sweeps.py::ois_case), HIP path against the C oracle (oracle/port.c).  Prints one JSON line
per case and a summary; `python tests/sweep_gpu_parity.py [cases] [trades] [first case]`.  The collected GPU suite
runs the same 23 curves at 20 000 trades each (tests/test_gpu_sweeps.py)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adrates_amd import _native
from tests._sweeps import ois_case

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # first case number (cases are seeded by their number)
ctx = _native.default_context()
worst_all, skipped, t_start = 0.0, 0, time.time()
for case in range(first, first + cases):
    r = ois_case(ctx, case, n)
    skipped += "skipped" in r
    worst_all = max(worst_all, r.get("worst", 0.0))
    print(json.dumps(r), flush=True)
print(json.dumps({"summary": True, "cases": cases, "first_case": first, "skipped": skipped, "trades_per_case": n, "worst_error": worst_all, "tolerance": 1e-10,
                  "pass": bool(worst_all <= 1e-10), "seconds": round(time.time() - t_start, 1)}))
