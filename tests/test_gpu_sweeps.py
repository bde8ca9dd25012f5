"""The parity sweep at scale as collected GPU tests: 23 random curves x the three interpolation schemes x mixed
portfolios (tests/_sweeps.py), HIP path against the C oracle at 1e-10.  6 000 trades per curve here so that the
whole GPU suite stays within the driver's time limit; `python tests/sweep_gpu_parity.py` runs the same cases at
100 000 trades each (profiles/r*_parity_sweep.jsonl)."""
import pytest

from ._sweeps import ois_case

pytestmark = pytest.mark.gpu


@pytest.mark.slow
@pytest.mark.parametrize("first", [0, 8, 16])
def test_random_curve_sweep(gpu_ctx, first):
    priced = 0
    for case in range(first, min(first + 8, 23)):
        r = ois_case(gpu_ctx, case, 6_000)
        if "skipped" in r:            # quotes that bootstrap to a non-positive discount factor: refused by the library
            continue
        priced += 1
        assert r["worst"] <= 1e-10, r
        assert r["agg_gamma_rel"] <= 1e-10, r
    assert priced >= 6
