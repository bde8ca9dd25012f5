// C-ABI of libadrates_hip.so (declarations and reference citations: include/adrates.h).
#include <hip/hip_runtime.h>
#include <cmath>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <functional>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "../../include/adrates.h"
#include "curve_tables.hpp"
#include "host_pool.hpp"
#include "kernels.hpp"
#include "route.hpp"

namespace {

thread_local std::string g_last_error;

constexpr size_t kLdsBudget = 160 * 1024;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int fail_hip(hipError_t e, const char* what) {
    return fail(ADR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define ADR_HIP(call)                                      \
    do {                                                   \
        hipError_t e__ = (call);                           \
        if (e__ != hipSuccess) return fail_hip(e__, #call); \
    } while (0)

template <typename T>
hipError_t upload(const std::vector<T>& host, T** dev) {
    *dev = nullptr;
    if (host.empty()) return hipSuccess;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dev), host.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace

struct adr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_cu = 0;
    size_t lds_limit = 0;
    double* partials = nullptr;     // [max_blocks][kAggStride] scratch for the aggregate
    double* dump = nullptr;                 // [32*32] store sink (kernels.hpp, OutputsDev::dump)
    unsigned long long* stamps = nullptr;   // diagnostic builds: [max_blocks*16][8]
    int max_blocks = 0;
    // aggregate-only mode (kernels_knot.hip): block records [knot_blocks][1 + 3 Kc] of the knot-space kernel and their sum
    double* knot_partials = nullptr;
    double* knot_reduced = nullptr;
    double* knot_overflow = nullptr;        // [kKnotLagMaxKc^2] pairs of knots beyond the bands (payment-lag rows)
    int knot_blocks = 0;
};

namespace {
constexpr int kKnotMaxKc = 640;             // reachable knots a curve can have (its dense 32-wide Jacobian must fit the LDS)
constexpr int kKnotLagMaxKc = 256;          // ... for the knot pass over payment-lag rows (16 pair bands per knot, a dense overflow matrix)
constexpr int kKnotStrideMax = std::max(1 + 3 * kKnotMaxKc, 1 + (2 + adr::kKnotBand) * kKnotLagMaxKc);
}

struct adr_curve {
    adr_ctx* ctx = nullptr;
    adr::CurveDev dev{};
    std::vector<void*> allocations;
};

// Rate-independent description of one knot grid: its bootstrap scan and the table layout of the base curve.
struct adr_curve_plan {
    adr_ctx* ctx = nullptr;
    int interp = 0;
    bool has_hess = false;
    adr::CurveTables base;               // host tables of the base curve (structure + base values)
    adr::CurveBuildPlanDev dev{};
    adr::CurveDev shared{};              // the structural device arrays every built curve points at
    std::vector<void*> allocations;
};

// Curves built together on the device; `curves` are views into the set's slabs.
struct adr_curve_set {
    adr_ctx* ctx = nullptr;
    const adr_curve_plan* plan = nullptr;
    int n = 0;
    double *dfs = nullptr, *jac = nullptr, *hess = nullptr;   // dense [n][K], [n][K][P], [n][K][P][P]
    std::vector<adr_curve> curves;
    std::vector<void*> allocations;
};

struct adr_trades {
    adr_ctx* ctx = nullptr;
    adr::TradesDev dev{};
    int64_t n_fix_flows = 0, n_flt_flows = 0;
    // Trades with a coupon whose accrual end differs from its payment time (payment lag) need the general
    // kernel (list_general, null when there are none), and so do legs of more than kMaxChain * 32 coupons.
    // list_fast holds the trades of at most 32 coupons per leg sorted by coupon
    // count, so that the trades sharing a wavefront in the fast kernel have similar lengths.
    int64_t n_fast = 0, n_long = 0, n_general = 0;
    const int32_t* list_general = nullptr;
    adr::TradesDev chained{};        // row table of the longer trades as chains of 32-coupon rows (fast kernel, LONG)
    int chained_blocks = 0;          // the grid the chains were laid out for
    // delta / PV-only requests: the trades without payment lag and at most 32 coupons per leg as 16-slot rows of the
    // lite kernel (the trades of the 32-slot row table); list_nonlite = every other trade (for curves without a packed layout)
    // gamma requests on curves with the packed layout: trades with payment lag or per-coupon notionals and at most 32
    // coupons per leg as rows of the payment-lag variant of the fast kernel; list_rest = the general list without them
    adr::TradesDev lagged{};
    adr::TradesDev lagged_chained{};   // ... those of 33-128 coupons per leg as chains of rows (LONG + LAG), laid out for
    int lagged_chained_blocks = 0;     // this grid
    int64_t n_lagged = 0, n_lagged_long = 0, n_rest = 0;
    // per-wave stash of the payment-lag variant (kernels.hpp, OutputsDev::lag_scratch), sized for a grid of lag_blocks
    // blocks.  It belongs to the BATCH (not to the ctx): two batches priced on two streams never share it.
    double* lag_scratch = nullptr;
    int lag_blocks = 0;
    const int32_t* list_rest = nullptr;
    adr::LiteRowsDev lite{};
    int64_t n_lite = 0, n_nonlite = 0;
    const int32_t* list_nonlite = nullptr;
    // ... and the trades with payment lag or per-coupon notionals and at most 135 coupons per leg as rows of the lite
    // kernel's payment-lag variant; list_general_b / list_nonlite_b = list_general / list_nonlite without them
    adr::LiteRowsDev lite_lag{};
    int64_t n_lite_lag = 0, n_general_b = 0, n_nonlite_b = 0;
    const int32_t* list_general_b = nullptr;
    const int32_t* list_nonlite_b = nullptr;
    std::vector<void*> allocations;

    // what the launch plan needs to know about the batch (route.hpp)
    adr::route::TradeCounts counts() const {
        adr::route::TradeCounts c;
        c.n = dev.n;
        c.rows = dev.n_rows; c.chained_rows = chained.n_rows; c.lagged_rows = lagged.n_rows; c.lagged_chained_rows = lagged_chained.n_rows;
        c.lite_units = lite.n_units; c.lite_lag_units = lite_lag.n_units;
        c.n_general = n_general; c.n_general_b = n_general_b; c.n_rest = n_rest; c.n_nonlite = n_nonlite; c.n_nonlite_b = n_nonlite_b;
        c.chained_blocks = chained_blocks; c.lagged_chained_blocks = lagged_chained_blocks; c.lag_blocks = lag_blocks;
        c.lag_scratch = lag_scratch != nullptr;
        return c;
    }
    // the plan of the last (curve class, request) this batch was priced with: built on first use, replayed afterwards
    struct PlanKey { int v[20]; };
    mutable std::mutex plan_mutex;
    mutable PlanKey plan_key{};
    mutable bool plan_valid = false;
    mutable adr::route::Plan plan;
    const adr::route::Plan& plan_for(const adr::CurveDev& cv, bool want_delta, bool want_gamma, bool per_trade, bool has_agg,
                                     const adr_ctx& c) const {
        const PlanKey key{{cv.K, cv.Kc, cv.P, cv.T, cv.wide_nch, cv.packed_ok, cv.method, cv.epg, cv.cpg, cv.Ec, cv.Kcore, cv.n_mini,
                           cv.n_lut, cv.n_fringe, cv.pc_pad, cv.Eu, want_delta ? 1 : 0, want_gamma ? 1 : 0, per_trade ? 1 : 0, has_agg ? 1 : 0}};
        std::lock_guard<std::mutex> lock(plan_mutex);
        if (!plan_valid || std::memcmp(&key, &plan_key, sizeof key) != 0) {
            plan = adr::route::make_plan(cv, counts(), want_delta, want_gamma, per_trade, has_agg, c.n_cu, c.max_blocks, c.knot_blocks,
                                         kKnotMaxKc, kKnotLagMaxKc);
            plan_key = key;
            plan_valid = true;
        }
        return plan;
    }
};

// the host-only translation units (book_host.cpp) report errors through the same per-thread message
int adr_set_error(int status, const std::string& msg) { return fail(status, msg); }

extern "C" {

int adr_version(void) { return 100; }

const char* adr_last_error(void) { return g_last_error.c_str(); }

int adr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int adr_init(int device_ordinal, adr_ctx** out) {
    if (!out) return fail(ADR_ERR_INVALID, "adr_init: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(ADR_ERR_HIP, "adr_init: no HIP device available (this library has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(ADR_ERR_INVALID, "adr_init: bad device ordinal");
    ADR_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    ADR_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    adr_ctx* ctx = new (std::nothrow) adr_ctx();
    if (!ctx) return fail(ADR_ERR_NOMEM, "adr_init: out of memory");
    ctx->device = device_ordinal;
    ctx->n_cu = prop.multiProcessorCount;
    ctx->lds_limit = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor
                                                           : prop.sharedMemPerBlock;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return fail_hip(e, "hipStreamCreate"); }
    ctx->max_blocks = std::max(1, ctx->n_cu) * 16;
    e = hipMalloc(reinterpret_cast<void**>(&ctx->partials),
                  sizeof(double) * static_cast<size_t>(ctx->max_blocks) * adr::kAggStride);
    if (e != hipSuccess) { hipStreamDestroy(ctx->stream); delete ctx; return fail_hip(e, "hipMalloc(partials)"); }
    e = hipMalloc(reinterpret_cast<void**>(&ctx->dump), sizeof(double) * adr::kPillarPad * adr::kPillarPad);
    if (e != hipSuccess) {
        hipFree(ctx->partials); hipStreamDestroy(ctx->stream); delete ctx;
        return fail_hip(e, "hipMalloc(dump)");
    }
    ctx->knot_blocks = std::max(1, ctx->n_cu) * adr::kLiteWavesPerSimd * 4 * 64 / adr::kLiteThreads;   // blocks resident at once
    e = hipMalloc(reinterpret_cast<void**>(&ctx->knot_partials),
                  sizeof(double) * (static_cast<size_t>(ctx->knot_blocks + 1) * kKnotStrideMax + static_cast<size_t>(kKnotLagMaxKc) * kKnotLagMaxKc + 1));
    if (e != hipSuccess) {
        hipFree(ctx->dump); hipFree(ctx->partials); hipStreamDestroy(ctx->stream); delete ctx;
        return fail_hip(e, "hipMalloc(knot partials)");
    }
    ctx->knot_reduced = ctx->knot_partials + static_cast<size_t>(ctx->knot_blocks) * kKnotStrideMax;
    ctx->knot_overflow = ctx->knot_reduced + kKnotStrideMax;
    // Dynamic-LDS ceiling of every kernel instantiation, once per device: the whole 160 KiB of a CU.  (Setting it per
    // uploaded curve to that curve's need would LOWER it below what an earlier, larger curve's launches request.)
    e = adr::set_kernel_lds_limits(kLdsBudget, kLdsBudget);
    if (e != hipSuccess) {
        hipFree(ctx->knot_partials); hipFree(ctx->dump); hipFree(ctx->partials); hipStreamDestroy(ctx->stream); delete ctx;
        return fail_hip(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
#ifdef ADR_STAMPS
    hipMalloc(reinterpret_cast<void**>(&ctx->stamps), sizeof(unsigned long long) * ctx->max_blocks * 16 * 8);
    hipMemset(ctx->stamps, 0, sizeof(unsigned long long) * ctx->max_blocks * 16 * 8);
#endif
    *out = ctx;
    return ADR_OK;
}

#ifdef ADR_STAMPS
extern "C" int adr_debug_stamps(adr_ctx* ctx, unsigned long long* host, int n_waves) {
    hipDeviceSynchronize();
    return hipMemcpy(host, ctx->stamps, sizeof(unsigned long long) * n_waves * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif

void adr_free_ctx(adr_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->partials) hipFree(ctx->partials);
    if (ctx->dump) hipFree(ctx->dump);
    if (ctx->knot_partials) hipFree(ctx->knot_partials);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int adr_sync(adr_ctx* ctx) {
    if (!ctx) return fail(ADR_ERR_INVALID, "adr_sync: ctx is null");
    ADR_HIP(hipStreamSynchronize(ctx->stream));
    return ADR_OK;
}

// ------------------------------------------------------------------------------------------- curve
int adr_curve_tables_host(int K, int P, const double* times, const double* dfs, const double* jac,
                          const double* hess, int32_t* knot_index, double* log_df, double* lj, double* lc) {
    adr::CurveTables t;
    const std::string err = adr::build_curve_tables(K, P, times, dfs, jac, hess, t);
    if (!err.empty()) return fail(ADR_ERR_INVALID, "adr_curve_tables_host: " + err);
    if (knot_index) std::copy(t.knot_index.begin(), t.knot_index.end(), knot_index);
    if (log_df) std::copy(t.log_df.begin(), t.log_df.end(), log_df);
    if (lj)
        for (int c = 0; c < t.Kc; ++c)
            for (int p = 0; p < P; ++p)
                lj[static_cast<size_t>(c) * P + p] =
                    t.lj[(static_cast<size_t>(p / adr::kPillarPad) * t.Kc + c) * adr::kPillarPad + p % adr::kPillarPad];
    if (lc && t.has_hess) std::copy(t.lc.begin(), t.lc.end(), lc);
    return t.Kc;
}

int adr_curve_layout_host(int K, int P, const double* times, const double* dfs, const double* jac,
                          const double* hess, int64_t* info) {
    if (!info) return fail(ADR_ERR_INVALID, "adr_curve_layout_host: info is null");
    adr::CurveTables t;
    const std::string err = adr::build_curve_tables(K, P, times, dfs, jac, hess, t);
    if (!err.empty()) return fail(ADR_ERR_INVALID, "adr_curve_layout_host: " + err);
    adr::CurveDev d{};
    d.K = t.K; d.Kc = t.Kc; d.Kcore = t.Kcore; d.pc_pad = t.pc_pad; d.Ec = t.Ec; d.Eu = t.Eu; d.epg = t.epg; d.cpg = t.cpg; d.hub = t.hub ? 1 : 0; d.n_lut = static_cast<int>(t.lut.size() / 2); d.n_mini = t.n_mini;
    info[0] = t.packed_ok ? 1 : 0; info[1] = t.Pc; info[2] = t.Ec; info[3] = t.Eu; info[4] = t.epg;
    info[5] = t.Kcore; info[6] = t.n_mini;
    info[7] = t.packed_ok ? static_cast<int64_t>(adr::fast_kernel_lds_bytes(d, t.has_hess)) : 0;
    {   // the general kernel's variant with LDS-resident convexity rows (it serves what the fast kernels do not take)
        const size_t g = (t.packed_ok && t.has_hess && t.T == 1)
            ? adr::general_lds_kernel_lds_bytes_for(t.K, t.Kc, t.Kcore, t.Ec, t.n_mini, static_cast<int>(t.lut.size() / 2), true) : 0;
        info[8] = static_cast<int64_t>(g);
        info[9] = adr::general_lds_rows_fit(g, t.Ec, t.n_fringe) ? 1 : 0;
    }
    info[10] = t.cpg;
    info[11] = t.hub ? 1 : 0;
    info[12] = t.wide_nch;
    info[13] = t.wide_nch > 0 ? static_cast<int64_t>(adr::wide_kernel_lds_bytes(t.K, t.Kc, t.wide_nch, t.has_hess)) : 0;
    info[14] = 0;
    for (uint32_t m : t.wide_knot_chunks) info[14] = std::max<int64_t>(info[14], __builtin_popcount(m));
    info[15] = 0;
    return ADR_OK;
}

void adr_free_curve(adr_curve* curve) {
    if (!curve) return;
    if (curve->ctx) hipSetDevice(curve->ctx->device);
    for (void* p : curve->allocations) hipFree(p);
    delete curve;
}

int adr_curve_pillars(const adr_curve* curve) { return curve ? curve->dev.P : 0; }

int adr_curve_upload(adr_ctx* ctx, int interp_method, int K, int P, const double* times, const double* dfs,
                     const double* jac, const double* hess, adr_curve** out) {
    return adr_curve_upload_ex(ctx, interp_method, K, P, times, dfs, jac, hess, 0u, out);
}

int adr_curve_upload_ex(adr_ctx* ctx, int interp_method, int K, int P, const double* times, const double* dfs,
                        const double* jac, const double* hess, uint32_t flags, adr_curve** out) {
    if (!ctx || !out) return fail(ADR_ERR_INVALID, "adr_curve_upload: null ctx/out");
    *out = nullptr;
    if (flags & ~static_cast<uint32_t>(ADR_CURVE_PILLAR_TILES))
        return fail(ADR_ERR_INVALID, "adr_curve_upload_ex: unknown flag bits");
    if (interp_method != ADR_INTERP_FLAT_FWD_RATES && interp_method != ADR_INTERP_LINEAR_ZERO_RATES &&
        interp_method != ADR_INTERP_LINEAR_FWD_RATES)
        return fail(ADR_ERR_UNSUPPORTED, "adr_curve_upload: only FLAT_FWD_RATES (1), LINEAR_FWD_RATES (2) and "
                                         "LINEAR_ZERO_RATES (4) are implemented");
    if (P > ADR_MAX_PILLARS)
        return fail(ADR_ERR_UNSUPPORTED, "adr_curve_upload: more than ADR_MAX_PILLARS (256) pillars");
    adr::CurveTables t;
    const std::string err = adr::build_curve_tables(K, P, times, dfs, jac, hess, t);
    if (!err.empty()) return fail(ADR_ERR_INVALID, "adr_curve_upload: " + err);

    const size_t lds = adr::general_kernel_lds_bytes(t.K, t.Kc, t.T > 1);
    if (lds > kLdsBudget)
        return fail(ADR_ERR_UNSUPPORTED, "adr_curve_upload: curve tables exceed the 160 KiB LDS of a CU");

    ADR_HIP(hipSetDevice(ctx->device));
    adr_curve* c = new (std::nothrow) adr_curve();
    if (!c) return fail(ADR_ERR_NOMEM, "adr_curve_upload: out of memory");
    c->ctx = ctx;
    std::vector<int16_t> first16(t.first_of.begin(), t.first_of.end());
    std::vector<int16_t> comp16(t.compact_of.begin(), t.compact_of.end());
    double *d_x = nullptr, *d_log = nullptr, *d_invx = nullptr, *d_lj = nullptr, *d_lc = nullptr;
    double *d_ljc = nullptr, *d_lcc = nullptr;
    int16_t *d_first = nullptr, *d_comp = nullptr, *d_class = nullptr, *d_p2c = nullptr, *d_omap = nullptr;
    int16_t* d_smap = nullptr;
    uint8_t* d_pq = nullptr;
    int16_t *d_cpos = nullptr, *d_lpos = nullptr;
    adr::MiniKnot* d_mini = nullptr;
    hipError_t e = hipSuccess;
    auto track = [&](hipError_t r, void* p) { if (p) c->allocations.push_back(p); if (e == hipSuccess) e = r; };
    track(upload(t.x, &d_x), d_x);
    track(upload(t.log_df, &d_log), d_log);
    track(upload(t.inv_x, &d_invx), d_invx);
    track(upload(t.lj, &d_lj), d_lj);
    track(upload(t.lc_lanes, &d_lc), d_lc);
    unsigned long long* d_lcmask = nullptr;
    {
        std::vector<unsigned long long> m(t.lc_block_mask.begin(), t.lc_block_mask.end());
        track(upload(m, &d_lcmask), d_lcmask);
    }
    track(upload(first16, &d_first), d_first);
    track(upload(comp16, &d_comp), d_comp);
    int16_t* d_lut = nullptr;
    track(upload(t.lut, &d_lut), d_lut);
    // wide layout (33-64 pillars): the whole ladder in one launch when its LDS image fits, else one launch per tile pair
    double *d_lj64 = nullptr, *d_lcflat = nullptr;
    uint32_t *d_went = nullptr, *d_wchunks = nullptr, *d_wsmap = nullptr;
    int32_t *d_wpos = nullptr, *d_worder = nullptr;
    const bool wide = t.wide_nch > 0 && t.wide_nch <= adr::kWideMaxChunks && !(flags & ADR_CURVE_PILLAR_TILES) &&
                      adr::wide_kernel_lds_bytes(t.K, t.Kc, t.wide_nch, t.has_hess) <= kLdsBudget;
    if (wide) {
        track(upload(t.lj64, &d_lj64), d_lj64);
        track(upload(t.wide_ent, &d_went), d_went);
        track(upload(t.wide_store_map, &d_wsmap), d_wsmap);
        track(upload(t.wide_pos, &d_wpos), d_wpos);
        track(upload(t.wide_order, &d_worder), d_worder);
        if (t.has_hess) {
            track(upload(t.lcflat, &d_lcflat), d_lcflat);
            track(upload(t.wide_knot_chunks, &d_wchunks), d_wchunks);
        }
    }
    if (t.packed_ok) {
        track(upload(t.ljc, &d_ljc), d_ljc);
        track(upload(t.lcc, &d_lcc), d_lcc);
        track(upload(t.knot_class, &d_class), d_class);
        track(upload(t.pillar_to_core, &d_p2c), d_p2c);
        track(upload(t.out_map, &d_omap), d_omap);
        track(upload(t.store_map, &d_smap), d_smap);
        track(upload(t.ent_pq, &d_pq), d_pq);
        track(upload(t.core_pos, &d_cpos), d_cpos);
        track(upload(t.lcc_pos, &d_lpos), d_lpos);
        track(upload(t.mini, &d_mini), d_mini);
    }
    if (e != hipSuccess) { adr_free_curve(c); return fail_hip(e, "adr_curve_upload: copying tables"); }
    c->dev.K = t.K; c->dev.Kc = t.Kc; c->dev.P = t.P; c->dev.method = interp_method;
    c->dev.T = t.T; c->dev.tile_i = c->dev.tile_j = 0;
    c->dev.x = d_x; c->dev.log_df = d_log; c->dev.inv_x = d_invx; c->dev.lj = d_lj; c->dev.lc_lanes = d_lc; c->dev.lc_block_mask = d_lcmask;
    c->dev.first_of = d_first; c->dev.compact_of = d_comp; c->dev.lut = d_lut; c->dev.n_lut = static_cast<int>(t.lut.size() / 2);
    c->dev.wide_nch = wide ? t.wide_nch : 0;
    c->dev.lj64 = d_lj64; c->dev.wide_ent = d_went; c->dev.lcflat = d_lcflat; c->dev.wide_knot_chunks = d_wchunks; c->dev.wide_store_map = d_wsmap; c->dev.wide_pos = d_wpos; c->dev.wide_order = d_worder;
    // the fast kernels store the [P][P] matrices as 16-byte pairs of the flat array: P must be even
    // LINEAR_FWD_RATES is linear in the knot DFs, not in their logs: only the general kernel carries the extra
    // Hessian term (kernels_general.hip, `Lookup`)
    c->dev.packed_ok = t.packed_ok ? 1 : 0;
    c->dev.odd_last = t.odd_last;
    c->dev.Pc = t.Pc; c->dev.pc_pad = t.pc_pad; c->dev.Ec = t.Ec; c->dev.Eu = t.Eu; c->dev.epg = t.epg; c->dev.cpg = t.cpg; c->dev.hub = t.hub ? 1 : 0;
    c->dev.Kcore = t.Kcore; c->dev.n_mini = t.n_mini; c->dev.fringe_start = t.fringe_start; c->dev.n_fringe = t.n_fringe; c->dev.fringe_own = t.fringe_own ? 1 : 0;
    c->dev.ljc = d_ljc; c->dev.lcc = d_lcc; c->dev.mini = d_mini; c->dev.knot_class = d_class;
    c->dev.pillar_to_core = d_p2c; c->dev.out_map = d_omap; c->dev.store_map = d_smap; c->dev.ent_pq = d_pq; c->dev.core_pos = d_cpos; c->dev.lcc_pos = d_lpos;
    // the packed tables must fit the LDS of a CU next to the search arrays, else the general kernel serves all
    size_t fast_lds = 0;
    if (c->dev.packed_ok) {
        fast_lds = adr::fast_kernel_lds_bytes(c->dev, t.has_hess);
        if (fast_lds > kLdsBudget) { c->dev.packed_ok = 0; fast_lds = 0; }
    }
    *out = c;
    return ADR_OK;
}

// ------------------------------------------------------------------------------ batched curve lookups
int adr_curve_df_dev(adr_ctx* ctx, const adr_curve* curve, int64_t n, const double* t_dev, double* df_dev, void* stream_v) {
    if (!ctx || !curve) return fail(ADR_ERR_INVALID, "adr_curve_df: null ctx/curve");
    if (curve->ctx != ctx) return fail(ADR_ERR_INVALID, "adr_curve_df: the curve was uploaded through another ctx");
    if (n < 0 || (n > 0 && (!t_dev || !df_dev))) return fail(ADR_ERR_INVALID, "adr_curve_df: bad count / null array");
    ADR_HIP(hipSetDevice(ctx->device));
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : ctx->stream;
    ADR_HIP(adr::launch_curve_df(curve->dev, n, t_dev, df_dev, ctx->n_cu, stream));
    return ADR_OK;
}

int adr_curve_df(adr_ctx* ctx, const adr_curve* curve, int64_t n, const double* t, double* df) {
    if (!ctx || !curve) return fail(ADR_ERR_INVALID, "adr_curve_df: null ctx/curve");
    if (n < 0 || (n > 0 && (!t || !df))) return fail(ADR_ERR_INVALID, "adr_curve_df: bad count / null array");
    if (n == 0) return ADR_OK;
    for (int64_t i = 0; i < n; ++i)
        if (!std::isfinite(t[i])) return fail(ADR_ERR_INVALID, "adr_curve_df: times must be finite");
    ADR_HIP(hipSetDevice(ctx->device));
    double *d_t = nullptr, *d_df = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_t), sizeof(double) * static_cast<size_t>(n));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_df), sizeof(double) * static_cast<size_t>(n));
    if (e == hipSuccess) e = hipMemcpyAsync(d_t, t, sizeof(double) * static_cast<size_t>(n), hipMemcpyHostToDevice, ctx->stream);
    int rc = ADR_OK;
    if (e == hipSuccess) rc = adr_curve_df_dev(ctx, curve, n, d_t, d_df, nullptr);
    if (e == hipSuccess && rc == ADR_OK)
        e = hipMemcpyAsync(df, d_df, sizeof(double) * static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == ADR_OK) e = hipStreamSynchronize(ctx->stream);
    hipFree(d_t); hipFree(d_df);
    if (rc != ADR_OK) return rc;
    if (e != hipSuccess) return fail_hip(e, "adr_curve_df");
    return ADR_OK;
}

// ------------------------------------------------------------------------------ device curve builder
void adr_free_curve_plan(adr_curve_plan* plan) {
    if (!plan) return;
    if (plan->ctx) hipSetDevice(plan->ctx->device);
    for (void* p : plan->allocations) hipFree(p);
    delete plan;
}

int adr_curve_plan_create(adr_ctx* ctx, int interp_method, int K, int P, const double* times, const double* acc,
                          const int32_t* pillar, const int32_t* prev_idx, const double* base_dfs,
                          const double* base_jac, const double* base_hess, adr_curve_plan** out) {
    if (!ctx || !out) return fail(ADR_ERR_INVALID, "adr_curve_plan_create: null ctx/out");
    *out = nullptr;
    if (interp_method != ADR_INTERP_FLAT_FWD_RATES && interp_method != ADR_INTERP_LINEAR_ZERO_RATES &&
        interp_method != ADR_INTERP_LINEAR_FWD_RATES)
        return fail(ADR_ERR_UNSUPPORTED, "adr_curve_plan_create: only FLAT_FWD_RATES (1), LINEAR_FWD_RATES (2) and "
                                         "LINEAR_ZERO_RATES (4) are implemented");
    if (P > ADR_MAX_PLAN_PILLARS)
        return fail(ADR_ERR_UNSUPPORTED, "adr_curve_plan_create: more than ADR_MAX_PLAN_PILLARS (64) pillars");
    if (!acc || !pillar || !prev_idx) return fail(ADR_ERR_INVALID, "adr_curve_plan_create: null scan arrays");
    for (int k = 0; k < K; ++k) {
        if (pillar[k] < 0 || pillar[k] >= P) return fail(ADR_ERR_INVALID, "adr_curve_plan_create: pillar index out of range");
        if (prev_idx[k] < -1 || prev_idx[k] >= K) return fail(ADR_ERR_INVALID, "adr_curve_plan_create: prev_idx out of range");
    }
    adr_curve_plan* plan = new (std::nothrow) adr_curve_plan();
    if (!plan) return fail(ADR_ERR_NOMEM, "adr_curve_plan_create: out of memory");
    plan->ctx = ctx;
    plan->interp = interp_method;
    plan->has_hess = base_hess != nullptr;
    adr::CurveTables& t = plan->base;
    const std::string err = adr::build_curve_tables(K, P, times, base_dfs, base_jac, base_hess, t);
    if (!err.empty()) { delete plan; return fail(ADR_ERR_INVALID, "adr_curve_plan_create: " + err); }
    // more than 32 pillars: the built curves carry the wide layout's tables only (no tiled route for them)
    const bool wide = t.T > 1;
    const size_t lds = wide ? adr::wide_kernel_lds_bytes(t.K, t.Kc, t.wide_nch, plan->has_hess) : adr::general_kernel_lds_bytes(t.K, t.Kc);
    // the PV01 gradients of the scan ([K][P] doubles) stay in LDS when they fit, else they go through a scratch buffer
    const bool dpv_global = adr::bootstrap_kernel_lds_bytes(K, P, false) > kLdsBudget;
    if (lds > kLdsBudget || adr::bootstrap_kernel_lds_bytes(K, P, dpv_global) > kLdsBudget) {
        delete plan;
        return fail(ADR_ERR_UNSUPPORTED, "adr_curve_plan_create: curve tables exceed the 160 KiB LDS of a CU");
    }
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { delete plan; return fail_hip(e, "hipSetDevice"); }

    std::vector<double> acc_v(acc, acc + K);
    std::vector<int32_t> pil_v(pillar, pillar + K), prev_v(prev_idx, prev_idx + K);
    std::vector<int16_t> first16(t.first_of.begin(), t.first_of.end()), comp16(t.compact_of.begin(), t.compact_of.end());
    std::vector<int32_t> core_pillars;
    for (int p = 0; p < P; ++p)
        if (t.packed_ok && t.pillar_to_core[p] < t.Pc) core_pillars.push_back(p);
    double *d_acc = nullptr, *d_x = nullptr, *d_invx = nullptr;
    int32_t *d_pil = nullptr, *d_prev = nullptr, *d_kidx = nullptr, *d_core = nullptr;
    int16_t *d_first = nullptr, *d_comp = nullptr, *d_class = nullptr, *d_p2c = nullptr, *d_omap = nullptr;
    int16_t* d_smap = nullptr;
    uint8_t *d_pq = nullptr, *d_lccpq = nullptr;
    int16_t *d_cpos = nullptr, *d_lpos = nullptr;
    auto track = [&](hipError_t r, void* p) { if (p) plan->allocations.push_back(p); if (e == hipSuccess) e = r; };
    track(upload(acc_v, &d_acc), d_acc);
    track(upload(pil_v, &d_pil), d_pil);
    track(upload(prev_v, &d_prev), d_prev);
    track(upload(t.knot_index, &d_kidx), d_kidx);
    unsigned long long* d_lcmask = nullptr;
    {
        std::vector<unsigned long long> m(t.lc_block_mask.begin(), t.lc_block_mask.end());
        track(upload(m, &d_lcmask), d_lcmask);
    }
    track(upload(t.x, &d_x), d_x);
    track(upload(t.inv_x, &d_invx), d_invx);
    track(upload(first16, &d_first), d_first);
    track(upload(comp16, &d_comp), d_comp);
    int16_t* d_lut = nullptr;
    track(upload(t.lut, &d_lut), d_lut);
    uint32_t *d_went = nullptr, *d_wchunks = nullptr, *d_wsmap = nullptr;
    int32_t *d_wpos = nullptr, *d_worder = nullptr;
    uint8_t* d_wpq = nullptr;
    if (wide) {
        track(upload(t.wide_ent, &d_went), d_went);
        track(upload(t.wide_store_map, &d_wsmap), d_wsmap);
        track(upload(t.wide_pos, &d_wpos), d_wpos);
        track(upload(t.wide_order, &d_worder), d_worder);
        track(upload(t.wide_pq, &d_wpq), d_wpq);
        if (plan->has_hess) track(upload(t.wide_knot_chunks, &d_wchunks), d_wchunks);
    }
    if (t.packed_ok) {
        track(upload(core_pillars, &d_core), d_core);
        track(upload(t.knot_class, &d_class), d_class);
        track(upload(t.pillar_to_core, &d_p2c), d_p2c);
        track(upload(t.out_map, &d_omap), d_omap);
        track(upload(t.store_map, &d_smap), d_smap);
        track(upload(t.ent_pq, &d_pq), d_pq);
        track(upload(t.lcc_pq, &d_lccpq), d_lccpq);
        track(upload(t.core_pos, &d_cpos), d_cpos);
        track(upload(t.lcc_pos, &d_lpos), d_lpos);
    }
    if (e != hipSuccess) { adr_free_curve_plan(plan); return fail_hip(e, "adr_curve_plan_create: copying tables"); }

    adr::CurveBuildPlanDev& d = plan->dev;
    d.K = K; d.P = P; d.Kc = t.Kc; d.acc = d_acc; d.pillar = d_pil; d.prev_idx = d_prev; d.knot_index = d_kidx;
    d.packed_ok = t.packed_ok ? 1 : 0;   // as in adr_curve_upload
    d.Pc = t.Pc; d.pc_pad = t.pc_pad; d.Ec = t.Ec; d.Kcore = t.Kcore; d.n_mini = t.n_mini;
    d.knot_class = d_class; d.core_pillars = d_core; d.lcc_pq = d_lccpq;
    d.wide_nch = wide ? t.wide_nch : 0; d.wide_pq = d_wpq; d.dpv_global = dpv_global ? 1 : 0;

    adr::CurveDev& c = plan->shared;
    c.K = t.K; c.Kc = t.Kc; c.P = t.P; c.method = interp_method; c.T = t.T; c.tile_i = c.tile_j = 0;
    c.wide_nch = d.wide_nch; c.wide_ent = d_went; c.wide_store_map = d_wsmap; c.wide_pos = d_wpos; c.wide_order = d_worder;
    c.wide_knot_chunks = d_wchunks;
    c.x = d_x; c.inv_x = d_invx; c.first_of = d_first; c.compact_of = d_comp; c.lc_block_mask = d_lcmask;
    c.lut = d_lut; c.n_lut = static_cast<int>(t.lut.size() / 2);
    c.packed_ok = d.packed_ok;
    c.odd_last = t.odd_last;
    c.Pc = t.Pc; c.pc_pad = t.pc_pad; c.Ec = t.Ec; c.Eu = t.Eu; c.epg = t.epg; c.cpg = t.cpg; c.hub = t.hub ? 1 : 0; c.Kcore = t.Kcore; c.n_mini = t.n_mini; c.fringe_start = t.fringe_start; c.n_fringe = t.n_fringe; c.fringe_own = t.fringe_own ? 1 : 0;
    c.knot_class = d_class; c.pillar_to_core = d_p2c; c.out_map = d_omap; c.store_map = d_smap; c.ent_pq = d_pq; c.core_pos = d_cpos; c.lcc_pos = d_lpos;
    size_t fast_lds = 0;
    if (c.packed_ok) {
        fast_lds = adr::fast_kernel_lds_bytes(c, plan->has_hess);
        if (fast_lds > kLdsBudget) { c.packed_ok = 0; d.packed_ok = 0; fast_lds = 0; }
    }
    *out = plan;
    return ADR_OK;
}

void adr_free_curve_set(adr_curve_set* set) {
    if (!set) return;
    if (set->ctx) hipSetDevice(set->ctx->device);
    for (void* p : set->allocations) hipFree(p);
    delete set;
}

int adr_curve_set_size(const adr_curve_set* set) { return set ? set->n : 0; }

const adr_curve* adr_curve_set_get(const adr_curve_set* set, int i) {
    if (!set || i < 0 || i >= set->n) { fail(ADR_ERR_INVALID, "adr_curve_set_get: index out of range"); return nullptr; }
    return &set->curves[static_cast<size_t>(i)];
}

int adr_curve_set_build(adr_ctx* ctx, const adr_curve_plan* plan, int n_scen, const double* rates,
                        adr_curve_set** out) {
    if (!ctx || !plan || !out) return fail(ADR_ERR_INVALID, "adr_curve_set_build: null ctx/plan/out");
    *out = nullptr;
    if (n_scen < 0 || (n_scen > 0 && !rates)) return fail(ADR_ERR_INVALID, "adr_curve_set_build: bad scenario count / null rates");
    const adr::CurveTables& t = plan->base;
    const size_t S = static_cast<size_t>(n_scen), K = t.K, P = t.P, Kc = t.Kc;
    for (size_t i = 0; i < S * P; ++i)
        if (!std::isfinite(rates[i])) return fail(ADR_ERR_INVALID, "adr_curve_set_build: par rates must be finite");
    ADR_HIP(hipSetDevice(ctx->device));
    adr_curve_set* set = new (std::nothrow) adr_curve_set();
    if (!set) return fail(ADR_ERR_NOMEM, "adr_curve_set_build: out of memory");
    set->ctx = ctx; set->plan = plan; set->n = n_scen;
    if (n_scen == 0) { *out = set; return ADR_OK; }

    hipError_t e = hipSuccess;
    auto alloc = [&](size_t bytes, bool zero) -> void* {
        if (bytes == 0 || e != hipSuccess) return nullptr;
        void* p = nullptr;
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return nullptr;
        set->allocations.push_back(p);
        if (zero) e = hipMemsetAsync(p, 0, bytes, ctx->stream);
        return p;
    };
    const bool hs = plan->has_hess, packed = plan->dev.packed_ok != 0;
    double* d_rates = static_cast<double*>(alloc(sizeof(double) * S * P, false));
    set->dfs = static_cast<double*>(alloc(sizeof(double) * S * K, false));
    set->jac = static_cast<double*>(alloc(sizeof(double) * S * K * P, false));
    set->hess = hs ? static_cast<double*>(alloc(sizeof(double) * S * K * P * P, false)) : nullptr;
    adr::CurvePackOut po{};
    po.log_df = static_cast<double*>(alloc(sizeof(double) * S * Kc, false));
    const bool wide = plan->dev.wide_nch > 0;
    const size_t wrow = static_cast<size_t>(plan->dev.wide_nch) * adr::kWideChunk;
    if (wide) {
        po.lj64 = static_cast<double*>(alloc(sizeof(double) * S * Kc * adr::kWidePad, false));
        po.lcflat = hs ? static_cast<double*>(alloc(sizeof(double) * S * Kc * wrow, true)) : nullptr;
    } else {
        po.lj = static_cast<double*>(alloc(sizeof(double) * S * Kc * adr::kPillarPad, false));
        po.lc_lanes = hs ? static_cast<double*>(alloc(sizeof(double) * S * Kc * 64 * adr::kGammaPerLane, false)) : nullptr;
    }
    const size_t ljc_n = static_cast<size_t>(t.Kcore + 1) * t.pc_pad, lcc_n = static_cast<size_t>(t.Kcore + 1) * (t.Ec + 1);
    if (packed) {
        po.ljc = static_cast<double*>(alloc(sizeof(double) * S * ljc_n, true));
        po.lcc = hs ? static_cast<double*>(alloc(sizeof(double) * S * lcc_n, true)) : nullptr;
        po.mini = static_cast<adr::MiniKnot*>(alloc(sizeof(adr::MiniKnot) * S * std::max(1, t.n_mini), false));
    }
    // scratch of the scan's second-derivative state; released once the build has run
    double *d_scratch = nullptr, *d_dpv = nullptr;
    if (hs && e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_scratch), sizeof(double) * S * K * P * P);
    if (plan->dev.dpv_global && e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_dpv), sizeof(double) * S * K * P);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rates, rates, sizeof(double) * S * P, hipMemcpyHostToDevice, ctx->stream);
    if (packed && t.n_mini > 0)
        for (size_t s = 0; s < S && e == hipSuccess; ++s)   // pillar / entry fields of the short-end records
            e = hipMemcpyAsync(po.mini + s * t.n_mini, t.mini.data(), sizeof(adr::MiniKnot) * t.n_mini,
                               hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = adr::launch_curve_build(plan->dev, n_scen, d_rates, set->dfs, set->jac, set->hess, d_scratch, d_dpv, po, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_scratch) hipFree(d_scratch);
    if (d_dpv) hipFree(d_dpv);
    if (e != hipSuccess) { adr_free_curve_set(set); return fail_hip(e, "adr_curve_set_build"); }

    set->curves.resize(S);
    for (size_t s = 0; s < S; ++s) {
        adr_curve& c = set->curves[s];
        c.ctx = ctx;
        c.dev = plan->shared;
        c.dev.log_df = po.log_df + s * Kc;
        if (wide) {
            c.dev.lj64 = po.lj64 + s * Kc * adr::kWidePad;
            c.dev.lcflat = hs ? po.lcflat + s * Kc * wrow : nullptr;
        } else {
            c.dev.lj = po.lj + s * Kc * adr::kPillarPad;
            c.dev.lc_lanes = hs ? po.lc_lanes + s * Kc * 64 * adr::kGammaPerLane : nullptr;
        }
        if (packed) {
            c.dev.ljc = po.ljc + s * ljc_n;
            c.dev.lcc = hs ? po.lcc + s * lcc_n : nullptr;
            c.dev.mini = po.mini + s * t.n_mini;
        }
    }
    *out = set;
    return ADR_OK;
}

int adr_curve_set_download(const adr_curve_set* set, int i, double* dfs, double* jac, double* hess) {
    if (!set || i < 0 || i >= set->n) return fail(ADR_ERR_INVALID, "adr_curve_set_download: index out of range");
    const size_t K = set->plan->base.K, P = set->plan->base.P, s = static_cast<size_t>(i);
    ADR_HIP(hipSetDevice(set->ctx->device));
    if (dfs) ADR_HIP(hipMemcpy(dfs, set->dfs + s * K, sizeof(double) * K, hipMemcpyDeviceToHost));
    if (jac) ADR_HIP(hipMemcpy(jac, set->jac + s * K * P, sizeof(double) * K * P, hipMemcpyDeviceToHost));
    if (hess) {
        if (!set->hess) return fail(ADR_ERR_INVALID, "adr_curve_set_download: the plan was created without hess");
        ADR_HIP(hipMemcpy(hess, set->hess + s * K * P * P, sizeof(double) * K * P * P, hipMemcpyDeviceToHost));
    }
    return ADR_OK;
}

// ------------------------------------------------------------------------------------------ trades
void adr_free_trades(adr_trades* t) {
    if (!t) return;
    if (t->ctx) hipSetDevice(t->ctx->device);
    for (void* p : t->allocations) hipFree(p);
    delete t;
}

int64_t adr_trades_count(const adr_trades* t) { return t ? t->dev.n : 0; }

int64_t adr_trades_input_bytes(const adr_trades* t) {
    if (!t) return 0;
    return 16 * t->n_fix_flows + 32 * t->n_flt_flows + 40 * t->dev.n;
}

int adr_trades_upload(adr_ctx* ctx, int64_t n, const int64_t* fix_off, const int64_t* flt_off, const double* fix_tp,
                      const double* fix_pay, const double* flt_tp, const double* flt_ts, const double* flt_te,
                      const double* flt_alpha, const double* notional, const double* spread, const double* fix_sign,
                      const double* flt_sign, adr_trades** out) {
    return adr_trades_upload_weighted(ctx, n, fix_off, flt_off, fix_tp, fix_pay, flt_tp, flt_ts, flt_te, flt_alpha,
                                      nullptr, notional, spread, fix_sign, flt_sign, out);
}

int adr_trades_upload_weighted(adr_ctx* ctx, int64_t n, const int64_t* fix_off, const int64_t* flt_off,
                               const double* fix_tp, const double* fix_pay, const double* flt_tp, const double* flt_ts,
                               const double* flt_te, const double* flt_alpha, const double* flt_weight,
                               const double* notional, const double* spread, const double* fix_sign,
                               const double* flt_sign, adr_trades** out) {
    if (!ctx || !out) return fail(ADR_ERR_INVALID, "adr_trades_upload: null ctx/out");
    *out = nullptr;
    if (n < 0) return fail(ADR_ERR_INVALID, "adr_trades_upload: negative trade count");
    if (n > 0 && (!fix_off || !flt_off || !notional || !spread || !fix_sign || !flt_sign))
        return fail(ADR_ERR_INVALID, "adr_trades_upload: null per-trade array");
    if (n > INT32_MAX) return fail(ADR_ERR_UNSUPPORTED, "adr_trades_upload: more than 2^31 trades in one batch");
    const int64_t n_fix = n ? fix_off[n] : 0, n_flt = n ? flt_off[n] : 0;
    if (n > 0 && (fix_off[0] != 0 || flt_off[0] != 0))
        return fail(ADR_ERR_INVALID, "adr_trades_upload: offsets must start at 0");

    // Host side of the upload: validation, the routing class of every trade and the row orders.  All of it is a
    // single pass over the caller's arrays, cut into contiguous trade ranges for a pool of threads; the tables
    // themselves are gathered on the device (trades_build.hip).
    const int n_threads = adr::pool_threads(n, 4096);
    auto parallel_ranges = [&](auto&& body) {          // body(range, first trade, one past the last trade); host_pool.hpp
        adr::parallel_ranges(n, n_threads, body);
    };
    // the offsets index the caller's arrays: check them before anything walks those arrays
    {
        std::vector<char> bad(static_cast<size_t>(n_threads), 0);
        parallel_ranges([&](int k, int64_t t0, int64_t t1) {
            for (int64_t t = t0; t < t1; ++t) {
                const int64_t mf = fix_off[t + 1] - fix_off[t], ml = flt_off[t + 1] - flt_off[t];
                if (mf < 0 || ml < 0 || mf > INT16_MAX || ml > INT16_MAX) { bad[static_cast<size_t>(k)] = 1; return; }
            }
        });
        for (char b : bad)
            if (b) return fail(ADR_ERR_INVALID, "adr_trades_upload: offsets must be non-decreasing, <= 32767 flows per leg");
    }
    if (n_fix > INT32_MAX || n_flt > INT32_MAX)
        return fail(ADR_ERR_UNSUPPORTED, "adr_trades_upload: more than 2^31 cash flows in one batch; shard the portfolio");
    if ((n_fix > 0 && (!fix_tp || !fix_pay)) || (n_flt > 0 && (!flt_tp || !flt_ts || !flt_te || !flt_alpha)))
        return fail(ADR_ERR_INVALID, "adr_trades_upload: null cash-flow array");

    // NaN / infinite inputs would only produce NaN outputs (every table index in the kernels is clamped), but a
    // batch that contains them is a caller error: say so here instead of returning a ladder of NaNs.
    // Class of a trade: bit 0 = a coupon accrues to a date other than its payment date (payment lag: ratio terms) or
    // carries a notional multiplier != 1.
    // (rows per trade in the chained tables - route.hpp, kMaxChain / kMaxChainLag: plain legs of up to 384 coupons (a 30Y
    // monthly leg is 360, cavour/utils/frequency.py:46); payment-lag legs of up to 128 (the variant's per-trade stash))
    std::vector<uint8_t> lagged_of(static_cast<size_t>(n), 0);
    {
        std::vector<char> bad(static_cast<size_t>(n_threads), 0);       // 1: not finite, 2: bad sign
        parallel_ranges([&](int k, int64_t t0, int64_t t1) {
            char err = 0;
            auto finite = [](const double* a, int64_t lo, int64_t hi) {
                bool ok = true;
                for (int64_t i = lo; i < hi; ++i) ok &= std::isfinite(a[i]);
                return ok;
            };
            if (t1 > t0) {
                const int64_t f0 = fix_off[t0], f1 = fix_off[t1], l0 = flt_off[t0], l1 = flt_off[t1];
                if (!finite(fix_tp, f0, f1) || !finite(fix_pay, f0, f1) || !finite(flt_tp, l0, l1) || !finite(flt_ts, l0, l1) ||
                    !finite(flt_te, l0, l1) || !finite(flt_alpha, l0, l1) || (flt_weight && !finite(flt_weight, l0, l1)) ||
                    !finite(notional, t0, t1) || !finite(spread, t0, t1))
                    err = 1;
            }
            for (int64_t t = t0; t < t1 && !err; ++t) {
                if (!(fix_sign[t] == 1.0 || fix_sign[t] == -1.0) || !(flt_sign[t] == 1.0 || flt_sign[t] == -1.0)) { err = 2; break; }
                bool lag = false;
                for (int64_t j = flt_off[t]; j < flt_off[t + 1] && !lag; ++j)
                    lag = (flt_alpha[j] > 0.0 && flt_te[j] != flt_tp[j]) || (flt_weight && flt_weight[j] != 1.0);
                lagged_of[static_cast<size_t>(t)] = lag ? 1 : 0;
            }
            bad[static_cast<size_t>(k)] = err;
        });
        for (char b : bad)
            if (b == 1) return fail(ADR_ERR_INVALID, "adr_trades_upload: times, amounts, accruals, notionals and spreads must be finite");
        for (char b : bad)
            if (b == 2) return fail(ADR_ERR_INVALID, "adr_trades_upload: leg signs must be +1 or -1");
    }

    auto coupons_of = [&](int64_t t) { return flt_off[t + 1] - flt_off[t]; };
    auto rows_of = [&](int64_t t) {
        const int64_t m = std::max(flt_off[t + 1] - flt_off[t], fix_off[t + 1] - fix_off[t]);
        return std::max<int64_t>(1, (m + adr::kRowSlots - 1) / adr::kRowSlots);
    };
    // stable order by float-coupon count, longest first (the trades sharing a wavefront then have similar lengths):
    // a counting sort for the one-row tables (at most 32 coupons), std::stable_sort for the short lists of longer trades
    auto sort_by_coupons = [&](std::vector<int32_t>& list) {
        bool small = true;
        for (int32_t t : list) small &= coupons_of(t) <= 64;
        if (!small) {
            std::stable_sort(list.begin(), list.end(), [&](int32_t a, int32_t b) { return coupons_of(a) > coupons_of(b); });
            return;
        }
        size_t count[66] = {0};
        for (int32_t t : list) ++count[64 - coupons_of(t) + 1];
        for (int b = 1; b < 66; ++b) count[b] += count[b - 1];
        std::vector<int32_t> sorted(list.size());
        for (int32_t t : list) sorted[count[64 - coupons_of(t)]++] = t;
        list.swap(sorted);
    };
    // which table or list every trade lands in (route.hpp: the same classification the launch plan's test walks)
    adr::route::TradeClasses cls;
    adr::route::classify_trades(n, fix_off, flt_off, lagged_of.data(), cls);
    std::vector<int32_t>&list_fast = cls.list_fast, &list_long = cls.list_long, &list_general = cls.list_general,
                        &list_lagged = cls.list_lagged, &list_lagged_long = cls.list_lagged_long, &list_rest = cls.list_rest;

    ADR_HIP(hipSetDevice(ctx->device));
    adr_trades* tr = new (std::nothrow) adr_trades();
    if (!tr) return fail(ADR_ERR_NOMEM, "adr_trades_upload: out of memory");
    tr->ctx = ctx;
    tr->n_fix_flows = n_fix;
    tr->n_flt_flows = n_flt;
    hipError_t e = hipSuccess;
    hipStream_t stream = ctx->stream;
    auto alloc = [&](size_t bytes) -> void* {
        if (bytes == 0 || e != hipSuccess) return nullptr;
        // over-allocate one header's worth so that the kernels' neighbour reads never leave the buffer
        void* p = nullptr;
        e = hipMalloc(&p, bytes + 64);
        if (e != hipSuccess) return nullptr;
        tr->allocations.push_back(p);
        return p;
    };
    auto put = [&](const void* src, size_t bytes) -> void* {
        void* p = alloc(bytes);
        if (p && e == hipSuccess) e = hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, stream);
        return p;
    };
    // (host vectors handed to `put` must outlive the asynchronous copies: they are kept until the final synchronisation)
    std::vector<std::vector<int32_t>> keep32;
    std::vector<std::vector<uint8_t>> keep8;
    auto put32 = [&](std::vector<int32_t>&& v) -> const int32_t* {
        keep32.push_back(std::move(v));
        return static_cast<const int32_t*>(put(keep32.back().data(), keep32.back().size() * sizeof(int32_t)));
    };
    auto put8 = [&](std::vector<uint8_t>&& v) -> const uint8_t* {
        keep8.push_back(std::move(v));
        return static_cast<const uint8_t*>(put(keep8.back().data(), keep8.back().size()));
    };

    // the caller's arrays, once
    adr::CsrDev csr{};
    csr.n = n;
    csr.fix_off = static_cast<const int64_t*>(put(fix_off, (n ? n + 1 : 0) * sizeof(int64_t)));
    csr.flt_off = static_cast<const int64_t*>(put(flt_off, (n ? n + 1 : 0) * sizeof(int64_t)));
    csr.fix_tp = static_cast<const double*>(put(fix_tp, n_fix * sizeof(double)));
    csr.fix_pay = static_cast<const double*>(put(fix_pay, n_fix * sizeof(double)));
    csr.flt_tp = static_cast<const double*>(put(flt_tp, n_flt * sizeof(double)));
    csr.flt_ts = static_cast<const double*>(put(flt_ts, n_flt * sizeof(double)));
    csr.flt_te = static_cast<const double*>(put(flt_te, n_flt * sizeof(double)));
    csr.flt_alpha = static_cast<const double*>(put(flt_alpha, n_flt * sizeof(double)));
    csr.flt_weight = flt_weight ? static_cast<const double*>(put(flt_weight, n_flt * sizeof(double))) : nullptr;
    csr.notional = static_cast<const double*>(put(notional, n * sizeof(double)));
    csr.spread = static_cast<const double*>(put(spread, n * sizeof(double)));
    csr.fix_sign = static_cast<const double*>(put(fix_sign, n * sizeof(double)));
    csr.flt_sign = static_cast<const double*>(put(flt_sign, n * sizeof(double)));

    tr->dev.n = n;
    tr->dev.any_ratio = 0;
    for (uint8_t lg : lagged_of) if (lg) { tr->dev.any_ratio = 1; break; }
    {
        adr::TradeHeader* hdr = static_cast<adr::TradeHeader*>(alloc(static_cast<size_t>(n) * sizeof(adr::TradeHeader)));
        if (e == hipSuccess && n > 0) e = adr::launch_build_headers(csr, hdr, stream);
        tr->dev.header = hdr;
    }
    tr->dev.fix_tp = csr.fix_tp; tr->dev.fix_pay = csr.fix_pay; tr->dev.flt_tp = csr.flt_tp; tr->dev.flt_ts = csr.flt_ts;
    tr->dev.flt_te = csr.flt_te; tr->dev.flt_alpha = csr.flt_alpha; tr->dev.flt_weight = csr.flt_weight;
    tr->dev.list = nullptr;
    tr->dev.n_list = n;
    tr->n_fast = static_cast<int64_t>(list_fast.size());
    tr->n_long = static_cast<int64_t>(list_long.size());
    tr->n_general = static_cast<int64_t>(list_general.size());
    sort_by_coupons(list_fast);
    tr->list_general = static_cast<const int32_t*>(put(list_general.data(), list_general.size() * sizeof(int32_t)));

    // Row tables of the fast kernel (kernels.hpp): 32 zero-padded slots per row and array, gathered on the device
    // from the work list (trade or -1 for an empty row, first coupon of the piece, "the trade continues" flag).
    struct Piece { int64_t trade; int64_t first; bool more; };
    auto build_rows = [&](std::vector<int32_t>&& piece_trade, std::vector<int32_t>&& piece_first, std::vector<uint8_t>&& piece_more,
                          adr::TradesDev& dst, bool lagged = false) {
        const size_t rows = piece_trade.size(), S = adr::kRowSlots;
        const bool chained = !piece_first.empty();
        adr::RowBuildDev rb{};
        rb.rows = static_cast<int64_t>(rows);
        rb.piece_trade = put32(std::move(piece_trade));
        rb.piece_first = chained ? put32(std::move(piece_first)) : nullptr;
        rb.piece_more = chained ? put8(std::move(piece_more)) : nullptr;
        rb.row_tp = static_cast<double*>(alloc(rows * S * sizeof(double)));
        rb.row_ts = static_cast<double*>(alloc(rows * S * sizeof(double)));
        rb.row_alpha = static_cast<double*>(alloc(rows * S * sizeof(double)));
        rb.row_xtp = static_cast<double*>(alloc(rows * S * sizeof(double)));
        rb.row_xpay = static_cast<double*>(alloc(rows * S * sizeof(double)));
        rb.row_te = lagged ? static_cast<double*>(alloc(rows * S * sizeof(double))) : nullptr;
        rb.row_w = (lagged && flt_weight) ? static_cast<double*>(alloc(rows * S * sizeof(double))) : nullptr;
        rb.row_notional = static_cast<double*>(alloc(rows * sizeof(double)));
        rb.row_spread = static_cast<double*>(alloc(rows * sizeof(double)));
        rb.row_meta = static_cast<int32_t*>(alloc(rows * sizeof(int32_t)));
        rb.row_trade = static_cast<int32_t*>(alloc(rows * sizeof(int32_t)));
        if (e == hipSuccess) e = adr::launch_build_rows(csr, rb, stream);
        dst.n_rows = static_cast<int64_t>(rows);
        dst.row_tp = rb.row_tp; dst.row_ts = rb.row_ts; dst.row_alpha = rb.row_alpha; dst.row_xtp = rb.row_xtp; dst.row_xpay = rb.row_xpay;
        dst.row_notional = rb.row_notional; dst.row_spread = rb.row_spread; dst.row_meta = rb.row_meta; dst.row_trade = rb.row_trade;
        dst.rows_lagged = lagged ? 1 : 0;
        dst.row_te = rb.row_te; dst.row_w = rb.row_w;
    };
    {   // plain table: one row per trade, sorted by coupon count
        tr->dev.rows_chained = 0;
        build_rows(std::vector<int32_t>(list_fast), {}, {}, tr->dev);
    }
    // Chained tables.  The kernel's wave w walks units w, w + W, w + 2W, ... (W = waves of the launch), so the
    // rows of a pair of trades (one per group of a wave) go to consecutive "rounds" of one wave column;
    // pairs are dealt to the columns longest first, always to the shortest column.
    auto build_chained = [&](std::vector<int32_t>& list, adr::TradesDev& dst, int blocks, int waves_per_block, bool lagged) {
        std::stable_sort(list.begin(), list.end(), [&](int32_t a, int32_t b) { return rows_of(a) > rows_of(b); });
        const int G = adr::fast_kernel_groups();
        const int64_t W = static_cast<int64_t>(blocks) * waves_per_block;
        // pass 1: the wave column and first round of every pair - always the shortest column, the lowest-numbered one among
        // equals (a heap of (height, column): with tens of thousands of pairs and thousands of columns a linear search
        // per pair was most of the upload's host time for books of long legs)
        const size_t n_pairs = (list.size() + static_cast<size_t>(G) - 1) / static_cast<size_t>(G);
        std::vector<int64_t> pair_col(n_pairs), pair_round(n_pairs);
        typedef std::pair<int64_t, int64_t> HW;                                  // (height, column)
        std::priority_queue<HW, std::vector<HW>, std::greater<HW>> heap;
        for (int64_t w = 0; w < W; ++w) heap.push(HW(0, w));
        int64_t rounds = 0;
        for (size_t pi = 0; pi < n_pairs; ++pi) {
            const HW top = heap.top();
            heap.pop();
            const int64_t len = rows_of(list[pi * static_cast<size_t>(G)]);     // the longest of the pair (sorted)
            pair_col[pi] = top.second; pair_round[pi] = top.first;
            heap.push(HW(top.first + len, top.second));
            rounds = std::max(rounds, top.first + len);
        }
        // pass 2: the rows
        const size_t total = static_cast<size_t>(rounds * W * G);
        std::vector<int32_t> p_trade(total, -1), p_first(total, 0);
        std::vector<uint8_t> p_more(total, 0);
        for (size_t pi = 0; pi < n_pairs; ++pi) {
            const size_t i = pi * static_cast<size_t>(G);
            const int64_t len = rows_of(list[i]);
            for (int64_t j = 0; j < len; ++j)
                for (int g = 0; g < G; ++g) {
                    const size_t at = static_cast<size_t>(((pair_round[pi] + j) * W + pair_col[pi]) * G + g);
                    p_more[at] = j + 1 < len ? 1 : 0;
                    if (i + static_cast<size_t>(g) >= list.size()) continue;          // an odd trade out: the slot stays empty
                    const int64_t t = list[i + static_cast<size_t>(g)];
                    p_first[at] = static_cast<int32_t>(j * adr::kRowSlots);
                    // results are written after the chain's last row, by the row's trade index: an empty padding row of
                    // the shorter trade still has to carry that index
                    if (j < rows_of(t) || j + 1 == len) p_trade[at] = static_cast<int32_t>(t);
                }
        }
        build_rows(std::move(p_trade), std::move(p_first), std::move(p_more), dst, lagged);
        dst.rows_chained = 1;
    };
    tr->chained = tr->dev;
    tr->chained.n_rows = 0;
    if (!list_long.empty()) {
        tr->chained_blocks = std::max(1, ctx->n_cu);
        build_chained(list_long, tr->chained, tr->chained_blocks, adr::kFastThreads / 64, false);
    }
    tr->lagged = tr->dev;
    tr->lagged.n_rows = 0;
    tr->lagged_chained = tr->dev;
    tr->lagged_chained.n_rows = 0;
    tr->n_lagged_long = static_cast<int64_t>(list_lagged_long.size());
    tr->n_lagged = static_cast<int64_t>(list_lagged.size());
    tr->n_rest = static_cast<int64_t>(list_rest.size());
    tr->list_rest = static_cast<const int32_t*>(put(list_rest.data(), list_rest.size() * sizeof(int32_t)));
    if (!list_lagged_long.empty()) {   // payment-lag legs of 33-128 coupons: chains of rows for the grid of the variant
        tr->lagged_chained_blocks = std::max(1, ctx->n_cu);
        build_chained(list_lagged_long, tr->lagged_chained, tr->lagged_chained_blocks,
                      adr::fast_kernel_threads(true) / 64, true);
    }
    if (!list_lagged.empty() || !list_lagged_long.empty()) {
        // sized for the grids these rows can be launched on: the chained rows' fixed grid, or as many blocks as the
        // one-row trades fill (557 KB per block: a batch with a handful of such trades must not pin 143 MB)
        const int waves = adr::fast_kernel_threads(true) / 64, G = adr::fast_kernel_groups();
        const int64_t units = (static_cast<int64_t>(list_lagged.size()) + G - 1) / G;
        const int need = static_cast<int>(std::min<int64_t>(std::max(1, ctx->n_cu), (units + waves - 1) / waves));
        const int blocks = std::max({1, need, list_lagged_long.empty() ? 0 : tr->lagged_chained_blocks});
        if (e == hipSuccess) {
            // per-wave scratch of the payment-lag variant: its special nodes' stash
            void* p = nullptr;
            const size_t bytes = adr::fast_kernel_lag_scratch_bytes(blocks);
            e = hipMalloc(&p, bytes);
            if (e == hipSuccess) { tr->allocations.push_back(p); tr->lag_scratch = static_cast<double*>(p); tr->lag_blocks = blocks; }
            if (e == hipSuccess) e = hipMemsetAsync(p, 0, bytes, stream);
        }
    }
    if (!list_lagged.empty()) {   // payment-lag rows: one row per trade, sorted by coupon count like the plain table
        sort_by_coupons(list_lagged);
        tr->lagged.rows_chained = 0;
        build_rows(std::vector<int32_t>(list_lagged), {}, {}, tr->lagged, true);
    }
    bool too_many_rows = false;
    {   // lite tables (kernels.hpp, LiteRowsDev): segments of equal row count, longest coupon counts first -
        // one for the trades of the 32-slot row table, one (with accrual ends and notional multipliers) for trades with
        // payment lag or per-coupon notionals of at most 390 coupons per leg (26 rows)
        constexpr int S = adr::kLiteSlots, G = 64 / adr::kLiteSlots;
        // rows per trade, rounded up to one of kLiteSegments row counts (the kernel keeps one segment per distinct count):
        // 1, 2, 3, 4, 6, 8, 12, 16, 26 rows = up to 390 coupons per leg; kLiteSegments = too long for the table
        const int64_t (&kRowBuckets)[adr::kLiteSegments] = adr::route::kLiteRowBuckets;
        // plain: the same trades as the 32-slot row table holds (at most 32 coupons per leg, i.e. up to 3 lite rows); longer
        // ones keep their chained rows.  (seg_*[k] holds bucket kLiteSegments - 1 - k: longest first; route.hpp)
        std::vector<int32_t> (&seg_plain)[adr::kLiteSegments] = cls.seg_plain, (&seg_lag)[adr::kLiteSegments] = cls.seg_lag;
        std::vector<int32_t>&nonlite = cls.nonlite, &nonlite_b = cls.nonlite_b, &general_b = cls.general_b;
        tr->n_nonlite = static_cast<int64_t>(nonlite.size());
        tr->list_nonlite = put32(std::move(nonlite));
        tr->n_nonlite_b = static_cast<int64_t>(nonlite_b.size());
        tr->list_nonlite_b = put32(std::move(nonlite_b));
        tr->n_general_b = static_cast<int64_t>(general_b.size());
        tr->list_general_b = put32(std::move(general_b));
        auto build_lite = [&](std::vector<int32_t> (&seg_trades)[adr::kLiteSegments], adr::LiteRowsDev& lt, int64_t& n_out,
                              bool with_te) {
            int64_t units = 0, rows = 0;
            int used[adr::kLiteSegments];                 // the non-empty row counts, longest first, packed to the front
            lt.n_seg = 0;
            for (int k = 0; k < adr::kLiteSegments; ++k) {
                lt.seg_rows[k] = 1; lt.seg_unit0[k] = 0; lt.seg_row0[k] = 0;
                if (seg_trades[k].empty()) continue;
                used[lt.n_seg++] = k;
            }
            for (int j = 0; j < lt.n_seg; ++j) {
                const int k = used[j];
                sort_by_coupons(seg_trades[k]);
                n_out += static_cast<int64_t>(seg_trades[k].size());
                lt.seg_rows[j] = static_cast<int>(kRowBuckets[adr::kLiteSegments - 1 - k]);
                lt.seg_unit0[j] = units;
                lt.seg_row0[j] = rows;
                const int64_t seg_units = (static_cast<int64_t>(seg_trades[k].size()) + G - 1) / G;
                units += seg_units;
                rows += seg_units * G * lt.seg_rows[j];
            }
            for (int j = lt.n_seg; j < adr::kLiteSegments; ++j) { lt.seg_unit0[j] = units; lt.seg_row0[j] = rows; }   // (never reached)
            lt.n_units = units;
            if (rows * S > static_cast<int64_t>(UINT32_MAX)) { too_many_rows = true; return; }   // the kernel indexes with 32 bits
            const size_t n_slots = static_cast<size_t>(units) * G, n_rows = static_cast<size_t>(rows);
            std::vector<int32_t> slot_trade(n_slots, -1), row_slot(n_rows, -1);
            std::vector<uint8_t> row_piece(n_rows, 0);
            for (int j = 0; j < lt.n_seg; ++j) {
                const int k = used[j];
                const int R = lt.seg_rows[j];
                const size_t slot0 = static_cast<size_t>(lt.seg_unit0[j]) * G, row0 = static_cast<size_t>(lt.seg_row0[j]);
                for (size_t i = 0; i < seg_trades[k].size(); ++i) slot_trade[slot0 + i] = seg_trades[k][i];
                const size_t slots_here = ((seg_trades[k].size() + G - 1) / G) * G;
                for (size_t i = 0; i < slots_here; ++i)
                    for (int r = 0; r < R; ++r) {
                        row_slot[row0 + i * static_cast<size_t>(R) + static_cast<size_t>(r)] = static_cast<int32_t>(slot0 + i);
                        row_piece[row0 + i * static_cast<size_t>(R) + static_cast<size_t>(r)] = static_cast<uint8_t>(r);
                    }
            }
            adr::LiteBuildDev lb{};
            lb.rows = static_cast<int64_t>(n_rows); lb.n_slots = static_cast<int64_t>(n_slots);
            lb.slot_trade = put32(std::move(slot_trade));
            lb.row_slot = put32(std::move(row_slot));
            lb.row_piece = put8(std::move(row_piece));
            lb.tp_ts = static_cast<double*>(alloc(n_rows * S * 2 * sizeof(double)));
            lb.al_xtp = static_cast<double*>(alloc(n_rows * S * 2 * sizeof(double)));
            lb.xpay = static_cast<double*>(alloc(n_rows * S * sizeof(double)));
            lb.te_w = with_te ? static_cast<double*>(alloc(n_rows * S * 2 * sizeof(double))) : nullptr;
            lb.slot = static_cast<adr::LiteTrade*>(alloc(n_slots * sizeof(adr::LiteTrade)));
            if (e == hipSuccess) e = adr::launch_build_lite(csr, lb, stream);
            lt.tp_ts = lb.tp_ts; lt.al_xtp = lb.al_xtp; lt.xpay = lb.xpay; lt.te_w = lb.te_w; lt.slot = lb.slot;
        };
        build_lite(seg_plain, tr->lite, tr->n_lite, false);
        if (!too_many_rows && tr->n_nonlite > tr->n_nonlite_b) build_lite(seg_lag, tr->lite_lag, tr->n_lite_lag, true);
    }
    // the copies and the table builders run on the ctx's stream: the batch is usable once they are done
    {
        const hipError_t es = hipStreamSynchronize(stream);        // (also on errors: the copies read this function's vectors)
        if (e == hipSuccess) e = es;
    }
    if (too_many_rows) {
        adr_free_trades(tr);
        return fail(ADR_ERR_UNSUPPORTED, "adr_trades_upload: more than 2^28 rows in the delta-only table; shard the portfolio");
    }
    if (e != hipSuccess) { adr_free_trades(tr); return fail_hip(e, "adr_trades_upload: copying trades"); }
    *out = tr;
    return ADR_OK;
}

// ------------------------------------------------------------------------------------------- price
int adr_price_dev(adr_ctx* ctx, const adr_curve* curve, const adr_trades* trades, uint32_t req_mask, double* pv_dev,
                  double* delta_dev, double* gamma_dev, double* agg_dev, void* stream_v) {
    if (!ctx || !curve || !trades) return fail(ADR_ERR_INVALID, "adr_price: null ctx/curve/trades");
    if (curve->ctx != ctx || trades->ctx != ctx)
        return fail(ADR_ERR_INVALID, "adr_price: curve/trades were uploaded through another ctx");
    const bool want_gamma = (req_mask & ADR_REQ_GAMMA) != 0;
    const bool want_delta = want_gamma || (req_mask & ADR_REQ_DELTA) != 0;
    if (want_gamma && !curve->dev.lc_lanes && !curve->dev.lcflat)
        return fail(ADR_ERR_INVALID, "adr_price: GAMMA requested but the curve was uploaded without hess");
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : ctx->stream;
    const int P = curve->dev.P;
    const int64_t n = trades->dev.n;
    const size_t agg_bytes = sizeof(double) * (1 + P + static_cast<size_t>(P) * P);

    ADR_HIP(hipSetDevice(ctx->device));
    if (n == 0) {   // empty portfolio: the aggregate is all zeros, nothing else to write
        if (agg_dev) ADR_HIP(hipMemsetAsync(agg_dev, 0, agg_bytes, stream));
        return ADR_OK;
    }

    adr::OutputsDev o{};
    o.stamps = ctx->stamps;
    o.dump = ctx->dump;
    o.pv = (req_mask & ADR_REQ_VALUE) ? pv_dev : nullptr;
    o.delta = (req_mask & ADR_REQ_DELTA) ? delta_dev : nullptr;
    o.gamma = want_gamma ? gamma_dev : nullptr;
    o.lag_scratch = trades->lag_scratch;
    o.knot_partials = ctx->knot_partials;
    o.knot_overflow = ctx->knot_overflow;

    // The launch plan (route.hpp): which kernel family takes which of the batch's tables / lists.  It depends on the curve's
    // class, the batch's table sizes and the request only; the batch keeps the last one (a book is priced again and again on
    // scenario curves of one class).
    const bool per_trade = o.pv || o.delta || o.gamma;
    const adr::route::Plan& plan = trades->plan_for(curve->dev, want_delta, want_gamma, per_trade, agg_dev != nullptr, *ctx);
    if (plan.error) return fail(ADR_ERR_INVALID, std::string("adr_price: ") + plan.error);

    namespace R = adr::route;
    const int stride = plan.wide ? adr::wide_partial_doubles(curve->dev.wide_nch) : adr::kAggStride;
    auto list_view = [&](int set) {              // the general / wide / tiled kernels walk a trade list
        adr::TradesDev v = trades->dev;
        v.list = nullptr; v.n_list = n;
        switch (set) {
            case R::S_GENERAL: v.list = trades->list_general; v.n_list = trades->n_general; break;
            case R::S_GENERAL_B: v.list = trades->list_general_b; v.n_list = trades->n_general_b; break;
            case R::S_REST: v.list = trades->list_rest; v.n_list = trades->n_rest; break;
            case R::S_NONLITE: v.list = trades->list_nonlite; v.n_list = trades->n_nonlite; break;
            case R::S_NONLITE_B: v.list = trades->list_nonlite_b; v.n_list = trades->n_nonlite_b; break;
            default: break;                      // S_ALL: the identity list
        }
        return v;
    };
    if (plan.tiled && agg_dev)   // tiles no launch covers (no GAMMA: the off-diagonal ones; PV alone: every delta tile) stay zero
        ADR_HIP(hipMemsetAsync(agg_dev, 0, agg_bytes, stream));
    const R::Launch *knot = nullptr, *knot_lag = nullptr;
    for (const R::Launch& L : plan.launches) {
        o.block_partials = agg_dev ? ctx->partials + static_cast<size_t>(L.first_block) * stride : nullptr;
        switch (L.family) {
            case R::F_LITE: ADR_HIP(adr::launch_price_lite(curve->dev, trades->lite, o, want_delta, L.blocks, stream)); break;
            case R::F_LITE_LAG: ADR_HIP(adr::launch_price_lite(curve->dev, trades->lite_lag, o, want_delta, L.blocks, stream)); break;
            case R::F_FAST: ADR_HIP(adr::launch_price_fast(curve->dev, trades->dev, o, want_delta, want_gamma, L.blocks, stream)); break;
            case R::F_FAST_CHAINED: ADR_HIP(adr::launch_price_fast(curve->dev, trades->chained, o, want_delta, want_gamma, L.blocks, stream)); break;
            case R::F_FAST_LAG: ADR_HIP(adr::launch_price_fast(curve->dev, trades->lagged, o, want_delta, want_gamma, L.blocks, stream)); break;
            case R::F_FAST_LAG_CHAINED: ADR_HIP(adr::launch_price_fast(curve->dev, trades->lagged_chained, o, want_delta, want_gamma, L.blocks, stream)); break;
            case R::F_GENERAL: ADR_HIP(adr::launch_price_general(curve->dev, list_view(L.set), o, want_delta, want_gamma, L.blocks, stream)); break;
            case R::F_WIDE: ADR_HIP(adr::launch_price_wide(curve->dev, list_view(L.set), o, want_delta, want_gamma, L.blocks, stream)); break;
            case R::F_TILED: {
                // each launch writes its tile of the ladders; its partials are reduced into its tile of the aggregate
                adr::CurveDev cv = curve->dev;
                cv.tile_i = L.tile_i; cv.tile_j = L.tile_j;
                const size_t pair_tile = static_cast<size_t>(curve->dev.Kc) * 64 * adr::kGammaPerLane;
                if (cv.lc_lanes) cv.lc_lanes += static_cast<size_t>(adr::tile_pair(L.tile_i, L.tile_j)) * pair_tile;
                if (cv.lc_block_mask) cv.lc_block_mask += static_cast<size_t>(adr::tile_pair(L.tile_i, L.tile_j)) * curve->dev.Kc;
                ADR_HIP(adr::launch_price_general(cv, list_view(L.set), o, want_delta, want_gamma, L.blocks, stream));
                if (agg_dev)
                    ADR_HIP(adr::launch_reduce_partials(o.block_partials, L.blocks, P, want_gamma, agg_dev, stream, L.tile_i, L.tile_j));
                break;
            }
            case R::F_KNOT: knot = &L; break;
            case R::F_KNOT_LAG: knot_lag = &L; break;
            default: return fail(ADR_ERR_INVALID, "adr_price: unknown kernel family in the launch plan");
        }
    }
    if (agg_dev && !plan.tiled) {
        if (plan.total_blocks == 0) ADR_HIP(hipMemsetAsync(agg_dev, 0, agg_bytes, stream));
        else if (plan.wide) ADR_HIP(adr::launch_reduce_wide(curve->dev, ctx->partials, plan.total_blocks, want_delta, want_gamma, agg_dev, stream));
        else ADR_HIP(adr::launch_reduce_partials(ctx->partials, plan.total_blocks, P, want_gamma, agg_dev, stream));
    }
    if (knot) {
        // aggregate-only request (agg and no per-trade output - Portfolio.compute's single ladder): the lite table's trades
        // are summed in KNOT space and projected once (kernels_lite.hip KNOT instantiations, kernels_knot.hip); the
        // projection ADDS to what the other families' reduction wrote above
        ADR_HIP(adr::launch_price_knot(curve->dev, trades->lite, o, want_gamma, knot->blocks, stream));
        ADR_HIP(adr::launch_knot_project(curve->dev, ctx->knot_partials, knot->blocks, ctx->knot_reduced, want_delta, want_gamma, 1, nullptr, agg_dev, stream));
    }
    if (knot_lag) {
        // ... and the payment-lag rows' ratio nodes: pair bands per wave, pairs farther apart in the launch's overflow matrix
        const size_t kc = static_cast<size_t>(curve->dev.Kc);
        if (want_gamma) ADR_HIP(hipMemsetAsync(ctx->knot_overflow, 0, sizeof(double) * (kc * kc + 1), stream));   // (+ the "in use" flag)
        ADR_HIP(adr::launch_price_knot(curve->dev, trades->lite_lag, o, want_gamma, knot_lag->blocks, stream));
        ADR_HIP(adr::launch_knot_project(curve->dev, ctx->knot_partials, knot_lag->blocks, ctx->knot_reduced, want_delta, want_gamma,
                                         adr::kKnotBand, ctx->knot_overflow, agg_dev, stream));
    }
    return ADR_OK;
}


int adr_price(adr_ctx* ctx, const adr_curve* curve, const adr_trades* trades, uint32_t req_mask, double* pv,
              double* delta, double* gamma, double* agg) {
    if (!ctx || !curve || !trades) return fail(ADR_ERR_INVALID, "adr_price: null ctx/curve/trades");
    const int P = curve->dev.P;
    const size_t n = static_cast<size_t>(trades->dev.n);
    const size_t n_agg = 1 + P + static_cast<size_t>(P) * P;
    ADR_HIP(hipSetDevice(ctx->device));
    double *d_pv = nullptr, *d_delta = nullptr, *d_gamma = nullptr, *d_agg = nullptr;
    int rc = ADR_OK;
    hipError_t e = hipSuccess;
    auto cleanup = [&]() { hipFree(d_pv); hipFree(d_delta); hipFree(d_gamma); hipFree(d_agg); };
    if (pv && (req_mask & ADR_REQ_VALUE) && n) e = hipMalloc(reinterpret_cast<void**>(&d_pv), n * sizeof(double));
    if (e == hipSuccess && delta && (req_mask & ADR_REQ_DELTA) && n)
        e = hipMalloc(reinterpret_cast<void**>(&d_delta), n * P * sizeof(double));
    if (e == hipSuccess && gamma && (req_mask & ADR_REQ_GAMMA) && n)
        e = hipMalloc(reinterpret_cast<void**>(&d_gamma), n * P * P * sizeof(double));
    if (e == hipSuccess && agg) e = hipMalloc(reinterpret_cast<void**>(&d_agg), n_agg * sizeof(double));
    if (e != hipSuccess) { cleanup(); return fail_hip(e, "adr_price: allocating outputs"); }
    rc = adr_price_dev(ctx, curve, trades, req_mask, d_pv, d_delta, d_gamma, d_agg, nullptr);
    if (rc != ADR_OK) { cleanup(); return rc; }
    e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && d_pv) e = hipMemcpy(pv, d_pv, n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && d_delta) e = hipMemcpy(delta, d_delta, n * P * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && d_gamma) e = hipMemcpy(gamma, d_gamma, n * P * P * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && d_agg) e = hipMemcpy(agg, d_agg, n_agg * sizeof(double), hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail_hip(e, "adr_price: running kernels / copying results");
    return ADR_OK;
}

// ------------------------------------------------------------------------ cross-currency foreign leg, two curves
int adr_price_xccy_foreign_dev(adr_ctx* ctx, const adr_curve* foreign_curve, const adr_curve* xccy_curve, const adr_trades* legs,
                               uint32_t req_mask, double* pv_dev, double* delta_foreign_dev, double* delta_basis_dev,
                               double* agg_foreign_dev, double* agg_basis_dev, void* stream_v) {
    if (!ctx || !foreign_curve || !xccy_curve || !legs) return fail(ADR_ERR_INVALID, "adr_price_xccy_foreign: null ctx/curve/legs");
    if (foreign_curve->ctx != ctx || xccy_curve->ctx != ctx || legs->ctx != ctx)
        return fail(ADR_ERR_INVALID, "adr_price_xccy_foreign: curves/legs were uploaded through another ctx");
    if (req_mask & ADR_REQ_GAMMA)
        return fail(ADR_ERR_UNSUPPORTED, "adr_price_xccy_foreign: GAMMA takes the three-batch route (adr_trades_upload_weighted + adr_price)");
    const adr::CurveDev &cf = foreign_curve->dev, &cx = xccy_curve->dev;
    if (cf.T > 1 || cx.T > 1 || (cf.method == ADR_INTERP_LINEAR_FWD_RATES) != (cx.method == ADR_INTERP_LINEAR_FWD_RATES))
        return fail(ADR_ERR_UNSUPPORTED, "adr_price_xccy_foreign: curves of up to 32 pillars, both on LINEAR_FWD_RATES or neither");
    if (adr::lite_xc_kernel_lds_bytes(cf, cx) > kLdsBudget)
        return fail(ADR_ERR_UNSUPPORTED, "adr_price_xccy_foreign: the two curves' tables exceed the LDS of a CU");
    const int64_t n = legs->dev.n;
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : ctx->stream;
    ADR_HIP(hipSetDevice(ctx->device));
    const int Pf = cf.P, Px = cx.P;
    const size_t agg_f_bytes = sizeof(double) * (1 + Pf + static_cast<size_t>(Pf) * Pf), agg_x_bytes = sizeof(double) * (1 + Px + static_cast<size_t>(Px) * Px);
    if (n == 0) {
        if (agg_foreign_dev) ADR_HIP(hipMemsetAsync(agg_foreign_dev, 0, agg_f_bytes, stream));
        if (agg_basis_dev) ADR_HIP(hipMemsetAsync(agg_basis_dev, 0, agg_x_bytes, stream));
        return ADR_OK;
    }
    // every leg must sit in the lite kernel's payment-lag rows (accrual end != payment time on some coupon, <= 390 coupons)
    if (legs->lite.n_units > 0 || legs->n_nonlite_b > 0 || legs->lite_lag.n_units == 0)
        return fail(ADR_ERR_UNSUPPORTED, "adr_price_xccy_foreign: a leg is outside the payment-lag row table (more than 390 coupons, "
                                         "or no coupon whose accrual end differs from its payment time)");
    // one block per CU: its registers leave room for the block's own three waves per SIMD
    const int blocks = adr::route::blocks_for(legs->lite_lag.n_units, adr::lite_xc_kernel_threads() / 64, static_cast<int64_t>(ctx->n_cu));
    if (2 * blocks > ctx->max_blocks) return fail(ADR_ERR_INVALID, "adr_price_xccy_foreign: grid exceeds scratch");
    const bool want_agg = agg_foreign_dev || agg_basis_dev;
    adr::OutputsDev o{};
    o.pv = (req_mask & ADR_REQ_VALUE) ? pv_dev : nullptr;
    o.delta = (req_mask & ADR_REQ_DELTA) ? delta_foreign_dev : nullptr;
    o.delta2 = (req_mask & ADR_REQ_DELTA) ? delta_basis_dev : nullptr;
    o.block_partials = want_agg ? ctx->partials : nullptr;
    o.block_partials2 = want_agg ? ctx->partials + static_cast<size_t>(blocks) * adr::kAggStride : nullptr;
    ADR_HIP(adr::launch_price_lite_xc(cf, cx, legs->lite_lag, o, blocks, stream));
    if (agg_foreign_dev) ADR_HIP(adr::launch_reduce_partials(o.block_partials, blocks, Pf, false, agg_foreign_dev, stream));
    if (agg_basis_dev) ADR_HIP(adr::launch_reduce_partials(o.block_partials2, blocks, Px, false, agg_basis_dev, stream));
    return ADR_OK;
}

int adr_price_xccy_foreign(adr_ctx* ctx, const adr_curve* foreign_curve, const adr_curve* xccy_curve, const adr_trades* legs,
                           uint32_t req_mask, double* pv, double* delta_foreign, double* delta_basis, double* agg_foreign,
                           double* agg_basis) {
    if (!ctx || !foreign_curve || !xccy_curve || !legs) return fail(ADR_ERR_INVALID, "adr_price_xccy_foreign: null ctx/curve/legs");
    const size_t n = static_cast<size_t>(legs->dev.n), Pf = foreign_curve->dev.P, Px = xccy_curve->dev.P;
    ADR_HIP(hipSetDevice(ctx->device));
    double *d_pv = nullptr, *d_f = nullptr, *d_x = nullptr, *d_af = nullptr, *d_ax = nullptr;
    hipError_t e = hipSuccess;
    auto cleanup = [&]() { hipFree(d_pv); hipFree(d_f); hipFree(d_x); hipFree(d_af); hipFree(d_ax); };
    auto get = [&](double** p, size_t count) { if (e == hipSuccess && count) e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(double)); };
    if (pv && (req_mask & ADR_REQ_VALUE)) get(&d_pv, n);
    if (delta_foreign && (req_mask & ADR_REQ_DELTA)) get(&d_f, n * Pf);
    if (delta_basis && (req_mask & ADR_REQ_DELTA)) get(&d_x, n * Px);
    if (agg_foreign) get(&d_af, 1 + Pf + Pf * Pf);
    if (agg_basis) get(&d_ax, 1 + Px + Px * Px);
    if (e != hipSuccess) { cleanup(); return fail_hip(e, "adr_price_xccy_foreign: allocating outputs"); }
    const int rc = adr_price_xccy_foreign_dev(ctx, foreign_curve, xccy_curve, legs, req_mask, d_pv, d_f, d_x, d_af, d_ax, nullptr);
    if (rc != ADR_OK) { cleanup(); return rc; }
    e = hipStreamSynchronize(ctx->stream);
    auto back = [&](double* dst, double* src, size_t count) { if (e == hipSuccess && src) e = hipMemcpy(dst, src, count * sizeof(double), hipMemcpyDeviceToHost); };
    back(pv, d_pv, n); back(delta_foreign, d_f, n * Pf); back(delta_basis, d_x, n * Px);
    back(agg_foreign, d_af, 1 + Pf + Pf * Pf); back(agg_basis, d_ax, 1 + Px + Px * Px);
    cleanup();
    if (e != hipSuccess) return fail_hip(e, "adr_price_xccy_foreign: running the kernel / copying results");
    return ADR_OK;
}

// ---------------------------------------------------------------------------------------- launch plan, host side
int adr_route_host(int interp_method, int K, int P, const double* times, const double* dfs, const double* jac, const double* hess,
                   uint32_t curve_flags, int64_t n, const int64_t* fix_off, const int64_t* flt_off, const double* flt_tp,
                   const double* flt_te, const double* flt_alpha, const double* flt_weight, uint32_t req_mask, int per_trade,
                   int aggregate, int n_cu, int32_t* cover, int32_t* launches, int max_launches) {
    if (n < 0 || (n > 0 && (!fix_off || !flt_off || !cover)) || !launches || max_launches < 1 || n_cu < 1)
        return fail(ADR_ERR_INVALID, "adr_route_host: bad argument");
    adr::CurveTables t;
    const std::string err = adr::build_curve_tables(K, P, times, dfs, jac, hess, t);
    if (!err.empty()) return fail(ADR_ERR_INVALID, "adr_route_host: " + err);
    // the curve's class, as adr_curve_upload_ex decides it (integer fields only: no table is read by the plan)
    adr::CurveDev cv{};
    cv.K = t.K; cv.Kc = t.Kc; cv.P = t.P; cv.method = interp_method; cv.T = t.T;
    cv.Pc = t.Pc; cv.pc_pad = t.pc_pad; cv.Ec = t.Ec; cv.Eu = t.Eu; cv.epg = t.epg; cv.cpg = t.cpg; cv.hub = t.hub ? 1 : 0;
    cv.Kcore = t.Kcore; cv.n_mini = t.n_mini; cv.n_fringe = t.n_fringe; cv.n_lut = static_cast<int>(t.lut.size() / 2);
    const bool wide = t.wide_nch > 0 && t.wide_nch <= adr::kWideMaxChunks && !(curve_flags & ADR_CURVE_PILLAR_TILES) &&
                      adr::wide_kernel_lds_bytes(t.K, t.Kc, t.wide_nch, t.has_hess) <= kLdsBudget;
    cv.wide_nch = wide ? t.wide_nch : 0;
    cv.packed_ok = (t.packed_ok && adr::fast_kernel_lds_bytes(cv, t.has_hess) <= kLdsBudget) ? 1 : 0;
    if ((req_mask & ADR_REQ_GAMMA) && !t.has_hess) return fail(ADR_ERR_INVALID, "adr_route_host: GAMMA requested but hess is null");
    // the trades' classes, as adr_trades_upload decides them
    std::vector<uint8_t> lagged_of(static_cast<size_t>(n), 0);
    for (int64_t tr = 0; tr < n; ++tr) {
        bool lag = false;
        for (int64_t j = flt_off[tr]; j < flt_off[tr + 1] && !lag; ++j)
            lag = (flt_alpha[j] > 0.0 && flt_te[j] != flt_tp[j]) || (flt_weight && flt_weight[j] != 1.0);
        lagged_of[static_cast<size_t>(tr)] = lag ? 1 : 0;
    }
    namespace R = adr::route;
    R::TradeClasses cls;
    R::classify_trades(n, fix_off, flt_off, lagged_of.data(), cls);
    R::TradeCounts tc;
    tc.n = n;
    tc.rows = static_cast<int64_t>(cls.list_fast.size());
    tc.chained_rows = static_cast<int64_t>(cls.list_long.size());               // (non-zero is all the plan asks of the chained tables)
    tc.lagged_rows = static_cast<int64_t>(cls.list_lagged.size());
    tc.lagged_chained_rows = static_cast<int64_t>(cls.list_lagged_long.size());
    tc.lite_units = cls.lite_units; tc.lite_lag_units = cls.lite_lag_units;
    tc.n_general = static_cast<int64_t>(cls.list_general.size()); tc.n_general_b = static_cast<int64_t>(cls.general_b.size());
    tc.n_rest = static_cast<int64_t>(cls.list_rest.size());
    tc.n_nonlite = static_cast<int64_t>(cls.nonlite.size()); tc.n_nonlite_b = static_cast<int64_t>(cls.nonlite_b.size());
    tc.chained_blocks = cls.list_long.empty() ? 0 : n_cu;
    tc.lagged_chained_blocks = cls.list_lagged_long.empty() ? 0 : n_cu;
    tc.lag_scratch = !cls.list_lagged.empty() || !cls.list_lagged_long.empty();
    tc.lag_blocks = tc.lag_scratch ? n_cu : 0;
    const bool want_gamma = (req_mask & ADR_REQ_GAMMA) != 0, want_delta = want_gamma || (req_mask & ADR_REQ_DELTA) != 0;
    const R::Plan plan = R::make_plan(cv, tc, want_delta, want_gamma, per_trade != 0, aggregate != 0, n_cu, n_cu * 16,
                                      n_cu * adr::kLiteWavesPerSimd * 4 * 64 / adr::kLiteThreads, kKnotMaxKc, kKnotLagMaxKc);
    if (plan.error) return fail(ADR_ERR_INVALID, std::string("adr_route_host: ") + plan.error);
    // who is covered how often: the tile launches of one pass count once
    for (int64_t i = 0; i < n; ++i) cover[i] = 0;
    auto add = [&](const std::vector<int32_t>& list) { for (int32_t tr : list) ++cover[tr]; };
    int n_out = 0;
    for (const R::Launch& L : plan.launches) {
        if (n_out < max_launches) {
            int32_t* row = launches + 4 * n_out;
            row[0] = L.family; row[1] = L.set; row[2] = static_cast<int32_t>(std::min<int64_t>(L.items, INT32_MAX)); row[3] = L.blocks;
        }
        ++n_out;
        if (L.family == R::F_TILED && !(L.tile_i == 0 && L.tile_j == 0)) continue;
        switch (L.set) {
            case R::S_LITE: for (auto& v : cls.seg_plain) add(v); break;
            case R::S_LITE_LAG: for (auto& v : cls.seg_lag) add(v); break;
            case R::S_ROWS: add(cls.list_fast); break;
            case R::S_CHAINED: add(cls.list_long); break;
            case R::S_LAGGED: add(cls.list_lagged); break;
            case R::S_LAGGED_CHAINED: add(cls.list_lagged_long); break;
            case R::S_GENERAL: add(cls.list_general); break;
            case R::S_GENERAL_B: add(cls.general_b); break;
            case R::S_REST: add(cls.list_rest); break;
            case R::S_NONLITE: add(cls.nonlite); break;
            case R::S_NONLITE_B: add(cls.nonlite_b); break;
            default: for (int64_t i = 0; i < n; ++i) ++cover[i]; break;
        }
    }
    return n_out;
}

// ---------------------------------------------------------------------------------------- multi-GPU
int adr_allreduce_agg(adr_ctx* ctx, void* rccl_comm, double* agg_dev, int count, void* stream_v) {
    if (!ctx || !rccl_comm || !agg_dev || count <= 0) return fail(ADR_ERR_INVALID, "adr_allreduce_agg: bad argument");
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : ctx->stream;
    ncclResult_t r = ncclAllReduce(agg_dev, agg_dev, static_cast<size_t>(count), ncclDouble, ncclSum,
                                   static_cast<ncclComm_t>(rccl_comm), stream);
    if (r != ncclSuccess) return fail(ADR_ERR_RCCL, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    return ADR_OK;
}

int adr_rccl_unique_id(void* id_out) {
    if (!id_out) return fail(ADR_ERR_INVALID, "adr_rccl_unique_id: null output");
    static_assert(sizeof(ncclUniqueId) == ADR_RCCL_ID_BYTES, "ADR_RCCL_ID_BYTES must equal sizeof(ncclUniqueId)");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(ADR_ERR_RCCL, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    std::memcpy(id_out, &id, sizeof id);
    return ADR_OK;
}

int adr_rccl_comm_init(adr_ctx* ctx, const void* id, int n_ranks, int rank, void** comm_out) {
    if (!ctx || !id || !comm_out || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(ADR_ERR_INVALID, "adr_rccl_comm_init: bad argument");
    *comm_out = nullptr;
    ADR_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = ncclCommInitRank(&comm, n_ranks, uid, rank);
    if (r != ncclSuccess) return fail(ADR_ERR_RCCL, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    *comm_out = comm;
    return ADR_OK;
}

void adr_rccl_comm_destroy(void* rccl_comm) {
    if (rccl_comm) ncclCommDestroy(static_cast<ncclComm_t>(rccl_comm));
}

}  // extern "C"
