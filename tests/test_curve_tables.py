"""Product curve construction (closed-form recurrences, native log-space tables) vs the AD oracle."""
import numpy as np
import pytest

from adrates_amd import _native
from adrates_amd.market.curves.curve_tables import build_engine_curve, expand_knot_grid
from adrates_amd.utils import FrequencyTypes, LibError
from oracle import cavour_oracle as O

from . import _fixtures as F

CURVES = {
    "gbp_apr": lambda: F.readme_model().curves.GBP_OIS_SONIA,
    "gbp_dec": lambda: F.gbp_model(F.TEST_VALUE_DT).curves.GBP_OIS_SONIA,
    "usd_dec": lambda: F.usd_model().curves.USD_OIS_SOFR,
    "gbp_5pillar": lambda: F.gbp_model(px=[5.19, 5.13, 5.04, 4.75, 4.24],
                                       tenors=["1M", "3M", "6M", "1Y", "5Y"]).curves.GBP_OIS_SONIA,
    "gbp_semi": lambda: F.gbp_model(F.TEST_VALUE_DT, px=F.GBP_PX[:25], tenors=F.TENORS[:25],
                                    freq=FrequencyTypes.SEMI_ANNUAL).curves.GBP_OIS_SONIA,
}


@pytest.mark.parametrize("name", list(CURVES))
def test_closed_form_derivatives_match_ad(name):
    import time
    curve = CURVES[name]()
    t0 = time.perf_counter()
    host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    t1 = time.perf_counter()
    ref = O.cached_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    t2 = time.perf_counter()
    # (the reference's method - AD through the scan - against the closed-form recurrences, for DESIGN.md section 5)
    print(f"{name}: closed forms {1e3 * (t1 - t0):.1f} ms, torch.func jacrev + hessian of the scan {t2 - t1:.2f} s")
    assert np.array_equal(host.times, ref["times"])
    assert np.array_equal(host.dfs, ref["dfs"])                       # same arithmetic, same bits
    assert np.allclose(host.jac, ref["jac"], rtol=1e-12, atol=1e-14)
    assert np.allclose(host.hess, ref["hess"], rtol=1e-11, atol=1e-12)
    assert np.allclose(host.hess, np.swapaxes(host.hess, 1, 2), rtol=0, atol=0)
    assert host.dfs[0] == 1.0 and not host.jac[0].any()
    assert host.n_knots == 1 + sum(len(f) for f in curve.year_fracs)


def test_knot_grid_structure_readme():
    curve = F.readme_model().curves.GBP_OIS_SONIA
    times, acc, pillar, prev, collisions = expand_knot_grid(curve.swap_rates, curve.year_fracs)
    assert times.shape == (264,) and len(np.unique(times)) == 66
    assert np.all(np.diff(times) >= 0)
    assert max(np.sum(times == t) for t in np.unique(times)) == 17     # cluster at t = 1.0
    # first duplicate of a cluster belongs to the shortest swap, last one to the longest
    at1 = np.where(times == 1.0)[0]
    assert pillar[at1[0]] == 14 and pillar[at1[-1]] == 31
    assert prev[0] == -1 and np.all(prev[1:][acc[1:] > 0] < np.arange(1, 264)[acc[1:] > 0])
    assert all(k == 0.0 for k, _, _ in collisions)                     # only the benign t=0 / 1D key clash


def test_native_log_space_tables(native_lib):
    curve = F.readme_model().curves.GBP_OIS_SONIA
    host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    t = _native.curve_tables_host(host.times, host.dfs, host.jac, host.hess)
    idx = t["knot_index"]
    assert len(idx) == 107                                             # SURVEY: 107 of 264 knots reachable
    # kept knots = first and last of every run of equal times
    x = host.times
    first = np.r_[True, x[1:] != x[:-1]]
    last = np.r_[x[1:] != x[:-1], True]
    assert np.array_equal(idx, np.where(first | last)[0])
    d = host.dfs[idx]
    lj = host.jac[idx] / d[:, None]
    lc = host.hess[idx] / d[:, None, None] - lj[:, :, None] * lj[:, None, :]
    assert np.allclose(t["log_df"], np.log(d), rtol=4e-16, atol=0)
    assert np.allclose(t["lj"], lj, rtol=1e-15, atol=0)
    assert np.allclose(t["lc"], lc, rtol=1e-14, atol=1e-18)
    # second derivative of ln d by differencing the AD oracle is overkill; symmetry is cheap
    assert np.array_equal(t["lc"], np.swapaxes(t["lc"], 1, 2))


def test_native_table_argument_checks(native_lib):
    good = dict(times=np.array([0.0, 1.0, 2.0]), dfs=np.array([1.0, 0.95, 0.9]), jac=np.zeros((3, 2)))
    _native.curve_tables_host(**good)
    with pytest.raises(LibError):
        _native.curve_tables_host(np.array([0.0, 2.0, 1.0]), good["dfs"], good["jac"])       # unsorted
    with pytest.raises(LibError):
        _native.curve_tables_host(good["times"], np.array([1.0, -0.1, 0.9]), good["jac"])    # DF <= 0
    with pytest.raises(LibError):
        _native.curve_tables_host(good["times"], good["dfs"], np.zeros((3, 260)))            # too many pillars (256 at most)
    with pytest.raises(LibError, match="value time"):
        _native.curve_tables_host(np.array([0.5, 1.0, 2.0]), good["dfs"], good["jac"])       # grid not anchored at t = 0
    with pytest.raises(LibError, match="value time"):
        _native.curve_tables_host(good["times"], np.array([0.99, 0.95, 0.9]), good["jac"])   # D(0) != 1
    with pytest.raises(LibError, match="no sensitivity"):
        _native.curve_tables_host(good["times"], good["dfs"], np.full((3, 2), 0.1))


def test_curves_without_a_core_are_left_to_the_general_kernel(native_lib):
    """A two-pillar toy curve (examples/c_abi_example.c) has no knot that depends on three pillars, hence no core
    tables: the packed LDS layout must refuse it (the fast kernel's core arrays would be empty - this faulted on
    the GPU once)."""
    r1, r2 = 0.04, 0.045
    d1 = 1 / (1 + r1); d2 = (1 - r2 * d1) / (1 + r2); dd1 = -d1 * d1
    jac = np.array([[0, 0], [dd1, 0], [-r2 * dd1 / (1 + r2), (-d1 * (1 + r2) - (1 - r2 * d1)) / (1 + r2) ** 2]])
    hess = np.zeros((3, 2, 2)); hess[1, 0, 0] = 2 * d1 ** 3
    hess[2] = [[-r2 * 2 * d1 ** 3 / (1 + r2), -dd1 / (1 + r2) ** 2], [-dd1 / (1 + r2) ** 2, 2 * (1 + d1) / (1 + r2) ** 3]]
    info = _native.curve_layout_host(np.array([0.0, 1.0, 2.0]), np.array([1.0, d1, d2]), jac, hess)
    assert info["packed_ok"] == 0
    # annual chains: four pillars are still below the layout's minimum core, ten are not
    from adrates_amd.market.curves.curve_tables import build_engine_curve
    for n, want in ((4, 0), (10, 1)):
        h = build_engine_curve([0.04 + 0.001 * i for i in range(n)], [float(i + 1) for i in range(n)],
                               [[1.0] * (i + 1) for i in range(n)])
        assert _native.curve_layout_host(h.times, h.dfs, h.jac, h.hess)["packed_ok"] == want


def test_more_than_32_pillars_are_tiled(native_lib):
    """Curves of 33-64 pillars: the host tables come in pillar tiles of 32 (curve_tables.hpp); through the plain-layout
    accessor they are the same log-space tables, LJ = J / d and LC = C / d - LJ LJ^T, for every pillar."""
    rng = np.random.default_rng(4)
    K, P = 9, 40
    times = np.concatenate(([0.0], np.sort(rng.uniform(0.1, 30.0, K - 1))))
    dfs = np.concatenate(([1.0], np.exp(-0.03 * times[1:])))
    jac = np.vstack((np.zeros((1, P)), rng.normal(0, 1, (K - 1, P))))
    hess = rng.normal(0, 1, (K, P, P))
    hess = hess + np.swapaxes(hess, 1, 2)
    hess[0] = 0.0
    t = _native.curve_tables_host(times, dfs, jac, hess)
    idx = t["knot_index"]
    lj = jac[idx] / dfs[idx, None]
    assert np.array_equal(t["lj"], lj)
    assert np.allclose(t["lc"], hess[idx] / dfs[idx, None, None] - lj[:, :, None] * lj[:, None, :], rtol=0, atol=1e-15)
    info = _native.curve_layout_host(times, dfs, jac, hess)
    assert info["packed_ok"] == 0            # the packed layout of the fast kernels is for one tile


def test_readme_sized_curves_keep_their_lds_resident_variants(native_lib):
    """Both LDS-resident table sets of the 32-pillar benchmark curve fit the 160 KB of a CU: the fast kernels' packed
    layout and the general kernel's resident convexity rows (which once fell 214 bytes short because of a table
    reserve, silently sending >32-coupon payment-lag legs to the L2-streaming variant)."""
    curve = F.readme_model().curves.GBP_OIS_SONIA
    host = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
    info = _native.curve_layout_host(host.times, host.dfs, host.jac, host.hess)
    assert info["packed_ok"] == 1 and info["entries_per_lane"] == 7 and info["core_pairs"] == 153
    assert 0 < info["lds_bytes"] <= 160 * 1024
    assert 0 < info["general_lds_bytes"] <= 160 * 1024 and info["general_lds_rows"] == 1
    # every fringe pair sits in the lane of one of its own pillars: two slots of 32 entries behind the core slots
    assert info["packed_entries"] == 32 * 7


def test_hub_layout_with_identity_row_positions_is_found_for_the_benchmark_curves():
    """`curve_tables.cpp::hub_layout` (round 3): the star decomposition of the core pairs is constrained so that entry
    lane + 32 * slot of a convexity row sits at that position of the compact row - 25 lanes of 5 pairs, 7 lanes of 4 for
    the 17 core pillars of the README curve - and the exact kernel variants (hub layout) are what these curves get; the
    LDS image still fits one CU."""
    from adrates_amd import _native
    from adrates_amd.trades.market_data import gbp_model, usd_model
    for model, name in ((gbp_model(), "GBP_OIS_SONIA"), (usd_model(), "USD_OIS_SOFR")):
        c = getattr(model.curves, name)
        h = build_engine_curve(c.swap_rates, c.swap_times, c.year_fracs)
        info = _native.curve_layout_host(h.times, h.dfs, h.jac, h.hess)
        assert info["packed_ok"] == 1 and info["hub_layout"] == 1, info
        assert info["core_pillars"] == 17 and info["core_pairs"] == 153 and info["core_slots_per_lane"] == 5
        assert info["entries_per_lane"] == 7 and info["lds_bytes"] <= 160 * 1024


def test_wide_layout_of_curves_with_more_than_32_pillars():
    """33-64 pillars: the wide layout (curve_tables.hpp) - the packed upper triangle of the gamma matrix in 7 / 10 / 17
    chunks of 128 entries, the kernel's LDS image inside a CU's 160 KB, and a knot's convexity row confined to a few chunks
    by the pillar order of the packing (CPU; the GPU parity tests are tests/test_gpu_many_pillars.py)."""
    from tests import _fixtures as F

    def years(t):
        return int(t[:-1]) / {"D": 365.0, "W": 52.0, "M": 12.0, "Y": 1.0}[t[-1]]

    base_t = np.array([years(t) for t in F.TENORS])
    extra = [f"{y}Y" for y in range(1, 50) if f"{y}Y" not in F.TENORS]
    for P, chunks in ((33, 7), (40, 7), (41, 7), (42, 10), (49, 10), (50, 17), (64, 17)):
        tenors = sorted(list(F.TENORS) + extra, key=years)[:P]
        px = [float(np.interp(years(t), base_t, F.GBP_PX)) if t not in F.TENORS else F.GBP_PX[F.TENORS.index(t)] for t in tenors]
        curve = F.gbp_model(F.README_VALUE_DT, px=px, tenors=tenors).curves.GBP_OIS_SONIA
        h = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
        info = _native.curve_layout_host(h.times, h.dfs, h.jac, h.hess)
        assert info["packed_ok"] == 0                                   # the fast kernels' layout is for one pillar tile
        assert info["wide_chunks"] == chunks, (P, info)
        assert 0 < info["wide_lds_bytes"] <= 160 * 1024, (P, info)
        # the padded triangle fits the chunks: sum over columns b of 2 * ceil((b + 1) / 2) entries
        assert sum(2 * ((b + 2) // 2) for b in range(P)) <= 128 * chunks
        # annual pillars sorted by use: a knot's pairs sit in a prefix of the packed array
        assert 1 <= info["wide_max_knot_chunks"] <= max(3, chunks * 2 // 3), (P, info)
    info32 = _native.curve_layout_host(*(lambda h: (h.times, h.dfs, h.jac, h.hess))(
        build_engine_curve(*(lambda c: (c.swap_rates, c.swap_times, c.year_fracs))(F.gbp_model().curves.GBP_OIS_SONIA))))
    assert info32["wide_chunks"] == 0 and info32["packed_ok"] == 1


def test_odd_pillar_counts_keep_the_packed_layout():
    """31 and 17 pillars: the packed layout (fast kernels) no longer needs an even pillar count - the layout analysis is the
    same as for the 32-pillar curve minus the dropped pillar (CPU; GPU parity: tests/test_gpu_many_pillars.py)."""
    from tests import _fixtures as F
    for drop in (0, 13, 31):
        tenors = [t for i, t in enumerate(F.TENORS) if i != drop]
        px = [p for i, p in enumerate(F.GBP_PX) if i != drop]
        curve = F.gbp_model(F.README_VALUE_DT, px=px, tenors=tenors).curves.GBP_OIS_SONIA
        h = build_engine_curve(curve.swap_rates, curve.swap_times, curve.year_fracs)
        info = _native.curve_layout_host(h.times, h.dfs, h.jac, h.hess)
        assert info["packed_ok"] == 1 and info["wide_chunks"] == 0, (drop, info)
        assert info["core_pillars"] in (16, 17) and 0 < info["lds_bytes"] <= 160 * 1024
