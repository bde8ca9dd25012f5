"""ctypes binding of libadrates_hip.so (C-ABI: include/adrates.h).

There is deliberately no CPU fallback: if the shared library has not been built
or no HIP device is usable, every pricing entry point raises.  Build the library
with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C adrates_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .utils.error import LibError

# ADRATES_HIP_LIB lets a tuning run point at an alternative build of the same library
_LIB_PATH = os.environ.get("ADRATES_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                               "libadrates_hip.so")
_lib = None

REQ_VALUE, REQ_DELTA, REQ_GAMMA = 1, 2, 4
MAX_PILLARS = 256            # ADR_MAX_PILLARS (uploaded curves); the device curve builder: 64

_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_vp = C.c_void_p

_SIGNATURES = {
    "adr_version": (C.c_int, []),
    "adr_last_error": (C.c_char_p, []),
    "adr_device_count": (C.c_int, []),
    "adr_init": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "adr_free_ctx": (None, [_vp]),
    "adr_sync": (C.c_int, [_vp]),
    "adr_curve_upload": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.POINTER(_vp)]),
    "adr_curve_upload_ex": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_uint32, C.POINTER(_vp)]),
    "adr_free_curve": (None, [_vp]),
    "adr_curve_pillars": (C.c_int, [_vp]),
    "adr_curve_tables_host": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _i32p, _dp, _dp, _dp]),
    "adr_curve_layout_host": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _i64p]),
    "adr_curve_plan_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp, _dp, _i32p, _i32p, _dp, _dp, _dp,
                                        C.POINTER(_vp)]),
    "adr_free_curve_plan": (None, [_vp]),
    "adr_curve_set_build": (C.c_int, [_vp, _vp, C.c_int, _dp, C.POINTER(_vp)]),
    "adr_curve_set_size": (C.c_int, [_vp]),
    "adr_curve_set_get": (_vp, [_vp, C.c_int]),
    "adr_curve_set_download": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "adr_free_curve_set": (None, [_vp]),
    "adr_trades_upload": (C.c_int, [_vp, C.c_int64, _i64p, _i64p, _dp, _dp, _dp, _dp, _dp, _dp,
                                    _dp, _dp, _dp, _dp, C.POINTER(_vp)]),
    "adr_trades_upload_weighted": (C.c_int, [_vp, C.c_int64, _i64p, _i64p, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                             _dp, _dp, _dp, _dp, C.POINTER(_vp)]),
    "adr_free_trades": (None, [_vp]),
    "adr_trades_count": (C.c_int64, [_vp]),
    "adr_trades_input_bytes": (C.c_int64, [_vp]),
    "adr_price": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _dp, _dp, _dp, _dp]),
    "adr_price_dev": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp]),
    "adr_allreduce_agg": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "adr_price_xccy_foreign": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, _dp, _dp, _dp, _dp, _dp]),
    "adr_price_xccy_foreign_dev": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "adr_route_host": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_uint32, C.c_int64, _i64p, _i64p, _dp, _dp, _dp, _dp,
                                 C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]),
    "adr_rccl_unique_id": (C.c_int, [_vp]),
    "adr_rccl_comm_init": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "adr_rccl_comm_destroy": (None, [_vp]),
    "adr_curve_df": (C.c_int, [_vp, _vp, C.c_int64, _dp, _dp]),
    "adr_curve_df_dev": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, _vp]),
    "adr_leg_counts_host": (C.c_int, [C.c_int64, _i64p, _i64p, _i64p, _i64p]),
    "adr_leg_times_host": (C.c_int, [C.c_int64, _i64p, _i64p, _i64p, _i64p, C.c_int, C.c_int, _dp, C.c_int64, C.c_double,
                                     _i64p, _dp, _dp, _dp, _dp, C.POINTER(C.c_uint8)]),
    "adr_exchange_flows_host": (C.c_int, [C.c_int64, _dp, _dp, C.POINTER(C.c_uint8), _dp, C.c_double, _i64p, _dp, _dp, _dp]),
    "adr_xccy_assemble_host": (C.c_int, [C.c_int64, _i64p, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, _dp,
                                         C.POINTER(C.c_uint8), _i64p, _dp, _dp, _dp, _dp, _i64p, _dp, _dp, _dp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def library_path() -> str:
    return _LIB_PATH


def load():
    """Load the shared library (no GPU needed for this step) and declare the
    argument types of every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise LibError(f"HIP extension not built: {_LIB_PATH} is missing "
                       "(run __graft_entry__.build() or make -C adrates_amd/csrc); "
                       "there is no CPU fallback for the pricing path")
    lib = C.CDLL(_LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc < 0:
        msg = load().adr_last_error().decode("utf-8", "replace")
        raise LibError(f"{what} failed ({rc}): {msg}")
    return rc


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, typ=_dp):
    return None if a is None else a.ctypes.data_as(typ)


class Context:
    """One per GPU/process: owns the device selection, a stream and scratch."""

    def __init__(self, device: int = 0):
        lib = load()
        h = _vp()
        _check(lib.adr_init(int(device), C.byref(h)), "adr_init")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            load().adr_free_ctx(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _check(load().adr_sync(self._h), "adr_sync")


class DeviceCurve:
    """Curve tables resident on the GPU (adr_curve_upload)."""

    PILLAR_TILES = 1      # ADR_CURVE_PILLAR_TILES (include/adrates.h)

    def __init__(self, ctx: Context, interp_method: int, times, dfs, jac, hess=None, flags: int = 0):
        times, dfs, jac = _f64(times), _f64(dfs), _f64(jac)
        K, P = jac.shape
        if times.shape != (K,) or dfs.shape != (K,):
            raise LibError("curve arrays have inconsistent shapes")
        hess_c = None
        if hess is not None:
            hess_c = _f64(hess)
            if hess_c.shape != (K, P, P):
                raise LibError("hess must have shape [K, P, P]")
        h = _vp()
        _check(load().adr_curve_upload_ex(ctx._h, int(interp_method), K, P, _ptr(times), _ptr(dfs), _ptr(jac),
                                          _ptr(hess_c), int(flags), C.byref(h)), "adr_curve_upload")
        self._h, self._ctx = h, ctx
        self.n_pillars, self.n_knots = P, K
        self.has_hess = hess is not None

    def close(self):
        if getattr(self, "_h", None):
            load().adr_free_curve(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CurvePlan:
    """Rate-independent half of a curve build (adr_curve_plan_create): the bootstrap scan of one knot grid
    and the table layout of its base curve.  ``host`` is an `EngineCurve` (curve_tables.build_engine_curve)."""

    def __init__(self, ctx: Context, interp_method: int, host):
        times, dfs, jac = _f64(host.times), _f64(host.dfs), _f64(host.jac)
        K, P = jac.shape
        acc = _f64(host.acc)
        pillar = np.ascontiguousarray(host.pillar, dtype=np.int32)
        prev_idx = np.ascontiguousarray(host.prev_idx, dtype=np.int32)
        if acc.shape != (K,) or pillar.shape != (K,) or prev_idx.shape != (K,):
            raise LibError("scan arrays must have one entry per knot")
        hess_c = _f64(host.hess) if host.hess is not None else None
        h = _vp()
        _check(load().adr_curve_plan_create(ctx._h, int(interp_method), K, P, _ptr(times), _ptr(acc),
                                            _ptr(pillar, _i32p), _ptr(prev_idx, _i32p), _ptr(dfs), _ptr(jac),
                                            _ptr(hess_c), C.byref(h)), "adr_curve_plan_create")
        self._h, self._ctx = h, ctx
        self.n_pillars, self.n_knots = P, K
        self.has_hess = hess_c is not None

    def build(self, rates) -> "CurveSet":
        """Bootstrap one curve per row of ``rates`` [S, P] (decimal par rates) on the GPU."""
        return CurveSet(self, rates)

    def close(self):
        if getattr(self, "_h", None):
            load().adr_free_curve_plan(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _SetCurve:
    """A curve owned by a `CurveSet`; accepted wherever a `DeviceCurve` is."""

    def __init__(self, handle, owner):
        self._h, self._owner = handle, owner        # keeps the set (and its plan) alive
        self.n_pillars, self.n_knots, self.has_hess = owner.n_pillars, owner.n_knots, owner.has_hess


class CurveSet:
    """Curves bootstrapped together on the GPU (adr_curve_set_build)."""

    def __init__(self, plan: CurvePlan, rates):
        rates = _f64(np.atleast_2d(rates))
        if rates.ndim != 2 or rates.shape[1] != plan.n_pillars:
            raise LibError("rates must have shape [n_scenarios, n_pillars]")
        h = _vp()
        _check(load().adr_curve_set_build(plan._ctx._h, plan._h, rates.shape[0], _ptr(rates), C.byref(h)),
               "adr_curve_set_build")
        self._h, self._plan, self._ctx = h, plan, plan._ctx
        self.n_pillars, self.n_knots, self.has_hess = plan.n_pillars, plan.n_knots, plan.has_hess
        self.n_curves = rates.shape[0]

    def __len__(self):
        return self.n_curves

    def __getitem__(self, i: int) -> _SetCurve:
        if not 0 <= i < self.n_curves:
            raise IndexError(i)
        return _SetCurve(_vp(load().adr_curve_set_get(self._h, int(i))), self)

    def download(self, i: int):
        """Dense arrays of scenario ``i``: ``dfs [K]``, ``jac [K, P]``, ``hess [K, P, P]`` (None without hess)."""
        K, P = self.n_knots, self.n_pillars
        dfs, jac = np.empty(K), np.empty((K, P))
        hess = np.empty((K, P, P)) if self.has_hess else None
        _check(load().adr_curve_set_download(self._h, int(i), _ptr(dfs), _ptr(jac), _ptr(hess)),
               "adr_curve_set_download")
        return dfs, jac, hess

    def close(self):
        if getattr(self, "_h", None):
            load().adr_free_curve_set(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceTrades:
    """A batch of OIS trades resident on the GPU (adr_trades_upload)."""

    def __init__(self, ctx: Context, batch):
        b = batch
        n = int(b.n_trades)
        arrs = dict(fix_off=np.ascontiguousarray(b.fix_off, dtype=np.int64),
                    flt_off=np.ascontiguousarray(b.flt_off, dtype=np.int64))
        for name in ("fix_tp", "fix_pay", "flt_tp", "flt_ts", "flt_te", "flt_alpha",
                     "notional", "spread", "fix_sign", "flt_sign"):
            arrs[name] = _f64(getattr(b, name))
        if arrs["fix_off"].shape != (n + 1,) or arrs["flt_off"].shape != (n + 1,):
            raise LibError("offset arrays must have n_trades + 1 entries")
        weight = getattr(b, "flt_weight", None)      # per-coupon notional multipliers (XCCY assembly) or None
        if weight is not None:
            weight = _f64(weight)
            if weight.shape != arrs["flt_tp"].shape:
                raise LibError("flt_weight must have one entry per float coupon")
        h = _vp()
        _check(load().adr_trades_upload_weighted(
            ctx._h, n, _ptr(arrs["fix_off"], _i64p), _ptr(arrs["flt_off"], _i64p),
            _ptr(arrs["fix_tp"]), _ptr(arrs["fix_pay"]), _ptr(arrs["flt_tp"]), _ptr(arrs["flt_ts"]),
            _ptr(arrs["flt_te"]), _ptr(arrs["flt_alpha"]), _ptr(weight), _ptr(arrs["notional"]), _ptr(arrs["spread"]),
            _ptr(arrs["fix_sign"]), _ptr(arrs["flt_sign"]), C.byref(h)), "adr_trades_upload")
        self._h, self._ctx = h, ctx
        self.n_trades = n

    @property
    def input_bytes(self) -> int:
        return int(load().adr_trades_input_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None):
            load().adr_free_trades(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def upload_many(ctx: Context, batches):
    """`DeviceTrades` of several batches at once, one host thread per batch (adr_trades_upload may be called from several
    threads on one ctx; ctypes releases the GIL): validation and classification of one batch overlap the copies and the
    device-side table builds of the others - the three batches of a cross-currency book in about half the serial time."""
    batches = list(batches)
    if len(batches) <= 1:
        return [DeviceTrades(ctx, b) for b in batches]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(batches)) as pool:
        return list(pool.map(lambda b: DeviceTrades(ctx, b), batches))


def price(ctx: Context, curve: DeviceCurve, trades: DeviceTrades, want_value=True, want_delta=True,
          want_gamma=True, per_trade=True, aggregate=False):
    """Blocking pricing call returning numpy arrays (adr_price).

    Returns a dict with ``pv [n]``, ``delta [n, P]``, ``gamma [n, P, P]`` (when
    ``per_trade``) and ``agg_pv``, ``agg_delta [P]``, ``agg_gamma [P, P]`` (when
    ``aggregate``)."""
    n, P = trades.n_trades, curve.n_pillars
    mask = (REQ_VALUE if want_value else 0) | (REQ_DELTA if want_delta else 0) | (REQ_GAMMA if want_gamma else 0)
    pv = np.empty(n) if (per_trade and want_value) else None
    delta = np.empty((n, P)) if (per_trade and want_delta) else None
    gamma = np.empty((n, P, P)) if (per_trade and want_gamma) else None
    agg = np.empty(1 + P + P * P) if aggregate else None
    _check(load().adr_price(ctx._h, curve._h, trades._h, mask, _ptr(pv), _ptr(delta), _ptr(gamma), _ptr(agg)),
           "adr_price")
    out = {}
    if pv is not None:
        out["pv"] = pv
    if delta is not None:
        out["delta"] = delta
    if gamma is not None:
        out["gamma"] = gamma
    if agg is not None:
        out["agg_pv"] = float(agg[0])
        out["agg_delta"] = agg[1:1 + P].copy()
        out["agg_gamma"] = agg[1 + P:].reshape(P, P).copy()
    return out


def price_xccy_foreign(ctx: Context, foreign_curve: DeviceCurve, xccy_curve: DeviceCurve, legs: DeviceTrades, want_value=True,
                       want_delta=True, per_trade=True, aggregate=False):
    """The foreign legs of a cross-currency book on two curves in one launch (adr_price_xccy_foreign): ``pv [n]``,
    ``delta_foreign [n, P_f]``, ``delta_basis [n, P_x]`` in FOREIGN currency (per trade), ``agg_pv``, ``agg_delta_foreign``,
    ``agg_delta_basis`` (book).  Raises `LibError` (ADR_ERR_UNSUPPORTED) for books the launch does not take."""
    n, Pf, Px = legs.n_trades, foreign_curve.n_pillars, xccy_curve.n_pillars
    mask = (REQ_VALUE if want_value else 0) | (REQ_DELTA if want_delta else 0)
    pv = np.empty(n) if (per_trade and want_value) else None
    df = np.empty((n, Pf)) if (per_trade and want_delta) else None
    dx = np.empty((n, Px)) if (per_trade and want_delta) else None
    af = np.empty(1 + Pf + Pf * Pf) if aggregate else None
    ax = np.empty(1 + Px + Px * Px) if aggregate else None
    _check(load().adr_price_xccy_foreign(ctx._h, foreign_curve._h, xccy_curve._h, legs._h, mask, _ptr(pv), _ptr(df), _ptr(dx),
                                         _ptr(af), _ptr(ax)), "adr_price_xccy_foreign")
    out = {}
    if pv is not None:
        out["pv"] = pv
    if df is not None:
        out["delta_foreign"], out["delta_basis"] = df, dx
    if aggregate:
        out["agg_pv"] = float(af[0])
        out["agg_delta_foreign"], out["agg_delta_basis"] = af[1:1 + Pf].copy(), ax[1:1 + Px].copy()
    return out


def price_xccy_foreign_dev(ctx: Context, foreign_curve: DeviceCurve, xccy_curve: DeviceCurve, legs: DeviceTrades, mask: int, pv_ptr=0,
                           delta_foreign_ptr=0, delta_basis_ptr=0, agg_foreign_ptr=0, agg_basis_ptr=0, stream=0):
    """Non-blocking form (adr_price_xccy_foreign_dev): device pointers as integers."""
    _check(load().adr_price_xccy_foreign_dev(ctx._h, foreign_curve._h, xccy_curve._h, legs._h, int(mask), _vp(pv_ptr or None),
                                             _vp(delta_foreign_ptr or None), _vp(delta_basis_ptr or None),
                                             _vp(agg_foreign_ptr or None), _vp(agg_basis_ptr or None), _vp(stream or None)),
           "adr_price_xccy_foreign_dev")


def price_dev(ctx: Context, curve: DeviceCurve, trades: DeviceTrades, mask: int, pv_ptr=0, delta_ptr=0,
              gamma_ptr=0, agg_ptr=0, stream=0):
    """Non-blocking pricing into caller-owned device buffers (adr_price_dev);
    pointers are integers (e.g. ``tensor.data_ptr()``), ``stream`` a hipStream_t
    handle (0 = the context's own stream)."""
    _check(load().adr_price_dev(ctx._h, curve._h, trades._h, int(mask), _vp(pv_ptr or None), _vp(delta_ptr or None),
                                _vp(gamma_ptr or None), _vp(agg_ptr or None), _vp(stream or None)), "adr_price_dev")


def curve_df(ctx: Context, curve: DeviceCurve, t):
    """Discount factors at the times ``t`` off an uploaded curve, evaluated on the GPU (adr_curve_df): the batched
    `InterpolatorAd.simple_interpolate`.  Scalars in, scalar out."""
    tt = _f64(np.atleast_1d(t))
    out = np.empty_like(tt)
    _check(load().adr_curve_df(ctx._h, curve._h, tt.size, _ptr(tt), _ptr(out)), "adr_curve_df")
    return float(out[0]) if np.ndim(t) == 0 else out.reshape(np.shape(t))


def curve_tables_host(times, dfs, jac, hess=None):
    """Log-space tables of the reachable knots, computed by the library's host
    code (no GPU needed) - used by the CPU tests."""
    times, dfs, jac = _f64(times), _f64(dfs), _f64(jac)
    K, P = jac.shape
    hess_c = None if hess is None else _f64(hess)
    lib = load()
    kc = _check(lib.adr_curve_tables_host(K, P, _ptr(times), _ptr(dfs), _ptr(jac), _ptr(hess_c),
                                          None, None, None, None), "adr_curve_tables_host")
    idx = np.empty(kc, dtype=np.int32)
    log_df = np.empty(kc)
    lj = np.empty((kc, P))
    lc = np.empty((kc, P, P)) if hess is not None else None
    _check(lib.adr_curve_tables_host(K, P, _ptr(times), _ptr(dfs), _ptr(jac), _ptr(hess_c),
                                     _ptr(idx, _i32p), _ptr(log_df), _ptr(lj), _ptr(lc)), "adr_curve_tables_host")
    return dict(knot_index=idx, log_df=log_df, lj=lj, lc=lc)


def curve_layout_host(times, dfs, jac, hess=None):
    """LDS layout the fast kernels would use for this curve (diagnostic; no GPU needed)."""
    times, dfs, jac = _f64(times), _f64(dfs), _f64(jac)
    K, P = jac.shape
    hess_c = None if hess is None else _f64(hess)
    info = np.zeros(16, dtype=np.int64)
    _check(load().adr_curve_layout_host(K, P, _ptr(times), _ptr(dfs), _ptr(jac), _ptr(hess_c), _ptr(info, _i64p)),
           "adr_curve_layout_host")
    keys = ("packed_ok", "core_pillars", "core_pairs", "packed_entries", "entries_per_lane", "core_rows",
            "mini_knots", "lds_bytes", "general_lds_bytes", "general_lds_rows", "core_slots_per_lane", "hub_layout",
            "wide_chunks", "wide_lds_bytes", "wide_max_knot_chunks", "reserved")
    return dict(zip(keys, (int(v) for v in info)))


def leg_times_host(effective, termination, months_per_period, payment_lag, bd_value, weekend_calendar, denominator,
                   value_serial, payment_denominator=None):
    """Coupon schedules of many legs (adr_leg_counts_host + adr_leg_times_host; threads on the host, no GPU):
    ``(off, tp, ts, te, alpha, plain)`` - CSR offsets over the coupons, payment / accrual start / accrual end times as year
    fractions from ``value_serial``, accrual fractions, and the mask of legs with strictly increasing dates."""
    i64 = lambda a, n: np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.int64), (n,)))
    eff = np.ascontiguousarray(effective, dtype=np.int64)
    n = eff.shape[0]
    term, mpp, lag = i64(termination, n), i64(months_per_period, n), i64(payment_lag, n)
    den = np.ascontiguousarray(np.broadcast_to(np.asarray(denominator, dtype=np.float64), (n,)))
    counts = np.empty(n, dtype=np.int64)
    lib = load()
    _check(lib.adr_leg_counts_host(n, _ptr(eff, _i64p), _ptr(term, _i64p), _ptr(mpp, _i64p), _ptr(counts, _i64p)),
           "adr_leg_counts_host")
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    m = int(off[-1])
    tp, ts, te, al = (np.empty(m) for _ in range(4))
    plain = np.empty(n, dtype=np.uint8)
    _check(lib.adr_leg_times_host(n, _ptr(eff, _i64p), _ptr(term, _i64p), _ptr(mpp, _i64p), _ptr(lag, _i64p), int(bd_value),
                                  1 if weekend_calendar else 0, _ptr(den), int(value_serial),
                                  float(payment_denominator or 0.0), _ptr(off, _i64p), _ptr(tp), _ptr(ts), _ptr(te), _ptr(al),
                                  _ptr(plain, C.POINTER(C.c_uint8))), "adr_leg_times_host")
    return off, tp, ts, te, al, plain.astype(bool)


def exchange_flows_host(exch_t, notional, on, sign, scale):
    """Notional exchanges of n legs (adr_exchange_flows_host): ``(off, flow_tp, flow_pay, pv_const)``."""
    exch_t = _f64(exch_t).reshape(-1)
    notional, sign = _f64(notional), _f64(sign)
    n = notional.shape[0]
    on = np.ascontiguousarray(on, dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.int64)
    tp, pay, const = np.empty(2 * n), np.empty(2 * n), np.empty(n)
    _check(load().adr_exchange_flows_host(n, _ptr(exch_t), _ptr(notional), _ptr(on, C.POINTER(C.c_uint8)), _ptr(sign), float(scale),
                                          _ptr(off, _i64p), _ptr(tp), _ptr(pay), _ptr(const)), "adr_exchange_flows_host")
    k = int(off[-1])
    return off, tp[:k], pay[:k], const


def xccy_assemble_host(for_off, tp_x, ts, te, alpha, disc, growth, for_n, for_spread, for_sign, spot, exch_t, exch_on,
                       pv_const):
    """Foreign-leg batches of a cross-currency book (adr_xccy_assemble_host): ``(rates_off, rates_ts, rates_te, rates_alpha,
    rates_weight, flows_off, flows_tp, flows_pay, pv_const)``; ``pv_const`` comes in with the domestic constants.
    ``disc`` [m + 1]: D_x at the payment times, then at the value time; ``growth`` [2 m]: D_f at the accrual starts, then ends."""
    for_off = np.ascontiguousarray(for_off, dtype=np.int64)
    n, m = for_off.shape[0] - 1, int(for_off[-1])
    cols = [_f64(a) for a in (tp_x, ts, te, alpha, disc, growth, for_n, for_spread, for_sign)]
    exch_t = _f64(exch_t).reshape(-1)
    exch_on = np.ascontiguousarray(exch_on, dtype=np.uint8)
    pv = np.array(pv_const, dtype=np.float64)
    r_off, f_off = np.empty(n + 1, dtype=np.int64), np.empty(n + 1, dtype=np.int64)
    r_ts, r_te, r_al, r_w = (np.empty(m) for _ in range(4))
    f_tp, f_pay = np.empty(m + 2 * n), np.empty(m + 2 * n)
    u8 = C.POINTER(C.c_uint8)
    _check(load().adr_xccy_assemble_host(n, _ptr(for_off, _i64p), *(_ptr(a) for a in cols), float(spot), _ptr(exch_t),
                                         _ptr(exch_on, u8), _ptr(r_off, _i64p), _ptr(r_ts), _ptr(r_te), _ptr(r_al), _ptr(r_w),
                                         _ptr(f_off, _i64p), _ptr(f_tp), _ptr(f_pay), _ptr(pv)), "adr_xccy_assemble_host")
    if n == 0:
        r_off[:] = 0; f_off[:] = 0
    kr, kf = int(r_off[-1]), int(f_off[-1])
    return r_off, r_ts[:kr], r_te[:kr], r_al[:kr], r_w[:kr], f_off, f_tp[:kf], f_pay[:kf], pv


_default_ctx = {}


def set_default_context(ctx: Context, device: int | None = None) -> None:
    """Make ``ctx`` the context `default_context` hands out (for hosts that create their own, e.g. one per rank)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
        n = load().adr_device_count()
        if n > 0:
            device %= n
    _default_ctx[device] = ctx


def default_context(device: int | None = None) -> Context:
    """Process-wide context for ``device`` (default: LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
        n = load().adr_device_count()
        if n > 0:
            device %= n
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def build_identity() -> dict:
    """What is running: sha256 of the loaded shared library and of the sources it is built from (adrates_amd/csrc/* and
    include/adrates.h, names and contents, sorted).  Evidence files under profiles/ carry the same two hashes
    (tools/profile_summary.py), and bench.py reports a counter-derived figure only when the source hash matches."""
    import glob
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    src = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "csrc", "*.hip")) + glob.glob(os.path.join(root, "csrc", "*.hpp")) +
                   glob.glob(os.path.join(root, "csrc", "*.cpp")) + [os.path.join(root, "csrc", "Makefile"),
                                                                    os.path.join(os.path.dirname(root), "include", "adrates.h")])
    for f in files:
        if os.path.exists(f):
            src.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                src.update(fh.read())
    lib = None
    if os.path.exists(_LIB_PATH):
        with open(_LIB_PATH, "rb") as fh:
            lib = hashlib.sha256(fh.read()).hexdigest()
    return {"source_sha256": src.hexdigest(), "lib_sha256": lib, "lib": os.path.relpath(_LIB_PATH, os.path.dirname(root))}


ROUTE_FAMILIES = ("lite", "lite_lag", "fast", "fast_chained", "fast_lag", "fast_lag_chained", "general", "wide", "tiled", "knot",
                  "knot_lag")
ROUTE_SETS = ("lite", "lite_lag", "rows", "chained", "lagged", "lagged_chained", "general", "general_b", "rest", "nonlite",
              "nonlite_b", "all")


def route_host(interp_method: int, times, dfs, jac, hess, batch, req_mask: int, per_trade=True, aggregate=False, n_cu=256,
               curve_flags=0):
    """The launch plan adr_price_dev would replay for this curve, batch and request, and how often it prices each trade
    (adr_route_host; no GPU needed).  Returns ``(launches, cover)``: launches = [(family, set, items, blocks)], names from
    ROUTE_FAMILIES / ROUTE_SETS; cover [n] int32."""
    times, dfs, jac = _f64(times), _f64(dfs), _f64(jac)
    K, P = jac.shape
    hess_c = None if hess is None else _f64(hess)
    n = batch.n_trades
    fo, lo = np.ascontiguousarray(batch.fix_off, dtype=np.int64), np.ascontiguousarray(batch.flt_off, dtype=np.int64)
    tp, te, al = _f64(batch.flt_tp), _f64(batch.flt_te), _f64(batch.flt_alpha)
    w = None if batch.flt_weight is None else _f64(batch.flt_weight)
    cover = np.zeros(max(n, 1), dtype=np.int32)
    rows = np.zeros((64, 4), dtype=np.int32)          # (a 256-pillar curve with GAMMA: 36 tile-pair launches + the knot passes)
    got = _check(load().adr_route_host(int(interp_method), K, P, _ptr(times), _ptr(dfs), _ptr(jac), _ptr(hess_c), int(curve_flags), n,
                                       _ptr(fo, _i64p), _ptr(lo, _i64p), _ptr(tp), _ptr(te), _ptr(al), _ptr(w), int(req_mask),
                                       1 if per_trade else 0, 1 if aggregate else 0, int(n_cu),
                                       cover.ctypes.data_as(C.POINTER(C.c_int32)), rows.ctypes.data_as(C.POINTER(C.c_int32)), 64),
                 "adr_route_host")
    launches = [(ROUTE_FAMILIES[f], ROUTE_SETS[s_], int(items), int(blocks)) for f, s_, items, blocks in rows[:min(got, 64)]]
    return launches, cover[:n]
