#!/usr/bin/env python3
"""Parity of the cross-currency book at scale (run on the GPU box): the three trade batches of a synthetic book of
distinct GBP/USD basis swaps priced by the HIP kernels and by the C oracle (tests/_sweeps.py::xccy_book_case).
`python tests/sweep_gpu_xccy.py [swaps]`; the collected suite runs the 100 000-swap book of BASELINE.json configs[3]
(tests/test_gpu_mixed_book.py)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adrates_amd import _native
from tests._sweeps import xccy_book_case

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rows = xccy_book_case(_native.default_context(), n)
for r in rows:
    print(json.dumps(r), flush=True)
worst = max(r["judged"] for r in rows)
print(json.dumps({"summary": True, "swaps": n, "worst_error": worst, "tolerance": 1e-10, "pass": bool(worst <= 1e-10)}))
