// Launch plan of a pricing call: which kernel family takes which set of trades, on what grid, and where its block partials
// go.  Pure host logic (no HIP call): adr_price_dev builds the plan from the curve's class and the batch's table sizes and
// replays it; adr_route_host exposes the same function for a CPU test that checks, over the cross product of trade
// classes, curve classes, schemes and requests, that every trade is priced exactly once (tests/test_route_table.py).
//
// The reference has one route - Engine._compute_ois_natural walks a trade's legs whatever they look like
// (cavour/market/position/engine.py:153-215); everything here is about which specialised kernel gives the same numbers fastest.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "kernels.hpp"

namespace adr {
namespace route {

// kernel families
enum Family : int {
    F_LITE = 0,           // lite kernel: PV / PV + delta, 16-slot rows, trades without payment lag (<= 384 coupons per leg)
    F_LITE_LAG = 1,       // ... its payment-lag rows (<= 360 coupons, log-linear schemes)
    F_FAST = 2,           // fast kernel on the 32-slot row table
    F_FAST_CHAINED = 3,   // ... chains of rows (33-384 coupons per leg)
    F_FAST_LAG = 4,       // payment-lag variant of the fast kernel, one-row trades (GAMMA, packed layout, log-linear, even P)
    F_FAST_LAG_CHAINED = 5,   // ... chains of rows (33-128 coupons)
    F_GENERAL = 6,        // general kernel over a trade list
    F_WIDE = 7,           // wide variants of the general kernel (33-64 pillars, one launch)
    F_TILED = 8,          // general kernel once per pair of 32-pillar tiles (curves whose wide tables do not fit / PILLAR_TILES)
    F_KNOT = 9,           // aggregate-only: knot-space sums of the lite table's trades + one projection
    F_KNOT_LAG = 10       // ... of the lite kernel's payment-lag rows (ratio nodes: pair bands; log-linear schemes)
};
// trade sets (adr_trades: tables and lists built by adr_trades_upload)
enum Set : int {
    S_LITE = 0, S_LITE_LAG = 1, S_ROWS = 2, S_CHAINED = 3, S_LAGGED = 4, S_LAGGED_CHAINED = 5,
    S_GENERAL = 6,        // payment lag / weights, or more than 384 coupons per leg
    S_GENERAL_B = 7,      // ... without the trades of the lite kernel's payment-lag rows
    S_REST = 8,           // ... without the trades of the payment-lag variant's tables
    S_NONLITE = 9,        // every trade outside the lite table
    S_NONLITE_B = 10,     // ... and outside the lite kernel's payment-lag rows
    S_ALL = 11
};

struct Launch {
    int family, set;
    int64_t items;        // rows / units / list entries the launch covers
    int blocks;
    int first_block;      // where its block partials start (in blocks of the plan's partial stride)
    int tile_i, tile_j;   // F_TILED only
};

// sizes of a batch's tables and lists (adr_trades), all the plan needs to know about the trades
struct TradeCounts {
    int64_t n = 0;
    int64_t rows = 0, chained_rows = 0, lagged_rows = 0, lagged_chained_rows = 0;
    int64_t lite_units = 0, lite_lag_units = 0;
    int64_t n_general = 0, n_general_b = 0, n_rest = 0, n_nonlite = 0, n_nonlite_b = 0;
    int chained_blocks = 0, lagged_chained_blocks = 0, lag_blocks = 0;
    bool lag_scratch = false;
};

struct Plan {
    std::vector<Launch> launches;    // in launch order; F_KNOT (if any) last: it ADDS to the aggregate the others' reduction wrote
    int total_blocks = 0;            // blocks that write partials (all families but F_KNOT, F_TILED keeps its own layout)
    bool wide = false;               // partials in the wide layout (launch_reduce_wide)
    bool tiled = false;              // one reduction per tile launch
    bool knot = false, knot_lag = false;
    const char* error = nullptr;     // "grid exceeds scratch"
};

constexpr size_t kLds = 160 * 1024;

inline int blocks_for(int64_t units, int waves, int64_t cap) {
    const int64_t need = (units + waves - 1) / waves;
    return static_cast<int>(std::min<int64_t>(need, cap));
}

// cv: only its integer fields are read (a CurveDev built from CurveTables on the host serves as well); has_hess: the curve
// carries second derivatives.  per_trade: some per-trade output pointer is non-null; has_agg: agg is requested.
inline Plan make_plan(const CurveDev& cv, const TradeCounts& tc, bool want_delta, bool want_gamma, bool per_trade, bool has_agg,
                      int n_cu, int max_blocks, int knot_blocks, int knot_kc_max, int knot_lag_kc_max) {
    Plan plan;
    const int64_t n = tc.n;
    if (n == 0) return plan;
    const bool lite_fits = lite_kernel_lds_bytes(cv, want_delta, true) <= kLds;
    const bool log_linear = cv.method != 2;
    // aggregate-only request: the lite table's trades in knot space
    plan.knot = has_agg && !per_trade && want_delta && tc.lite_units > 0 && cv.Kc <= knot_kc_max &&
                knot_kernel_lds_bytes(cv, want_gamma) <= kLds;
    // ... and the payment-lag rows' (ratio nodes, any scheme; the pair bands of 16 want more LDS and scratch per knot)
    plan.knot_lag = has_agg && !per_trade && want_delta && tc.lite_lag_units > 0 && cv.Kc <= knot_lag_kc_max &&
                    (plan.knot || tc.lite_units == 0) && knot_kernel_lds_bytes(cv, want_gamma, true) <= kLds;
    // the trades no knot pass takes: everything / outside the lite table / and outside its payment-lag rows
    const int rest_of_knot = plan.knot_lag ? S_NONLITE_B : (plan.knot ? S_NONLITE : S_ALL);
    const int64_t rest_of_knot_n = plan.knot_lag ? tc.n_nonlite_b : (plan.knot ? tc.n_nonlite : n);
    auto lite_blocks = [&](int64_t units, bool lag = false) {
        const size_t lds = lite_kernel_lds_bytes(cv, want_delta, lag);
        const int threads = lite_kernel_threads(cv, want_delta, lag);
        // (blocks of more than 512 threads are built for ONE per CU: their registers allow no second)
        const int per_cu = threads > kLiteThreads ? 1 : static_cast<int>(std::max<size_t>(1, std::min<size_t>(2, kLds / lds)));
        return blocks_for(units, threads / 64, static_cast<int64_t>(n_cu) * per_cu);
    };
    auto push = [&](int family, int set, int64_t items, int blocks, int ti = 0, int tj = 0) {
        if (items <= 0 || blocks <= 0) return;
        plan.launches.push_back(Launch{family, set, items, blocks, plan.total_blocks, ti, tj});
        if (family != F_KNOT) plan.total_blocks += blocks;
    };
    auto knot_launch = [&]() {
        for (int lag = 0; lag < 2; ++lag) {
            if (!(lag ? plan.knot_lag : plan.knot)) continue;
            const size_t lds = knot_kernel_lds_bytes(cv, want_gamma, lag != 0);
            const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(2, kLds / lds)));
            const int64_t units = lag ? tc.lite_lag_units : tc.lite_units;
            const int blocks = blocks_for(units, knot_kernel_threads() / 64, std::min<int64_t>(static_cast<int64_t>(n_cu) * per_cu, knot_blocks));
            plan.launches.push_back(Launch{lag ? F_KNOT_LAG : F_KNOT, lag ? S_LITE_LAG : S_LITE, units, blocks, 0, 0, 0});
        }
    };

    if (cv.T > 1 && cv.wide_nch > 0) {
        // 33-64 pillars, one launch for the whole ladder.  GAMMA: the wide variants; PV / PV + delta: the lite kernel's 64-pillar
        // instantiations for the trades of its tables, the wide kernel for the rest.
        plan.wide = true;
        const bool lite_elsewhere = (!want_gamma && lite_fits && tc.lite_units > 0) || plan.knot;
        const bool use_lite_lag = (!want_gamma && lite_fits && tc.lite_lag_units > 0) || plan.knot_lag;   // (priced elsewhere)
        int rest_set = S_ALL;
        int64_t rest_n = n;
        if (lite_elsewhere || use_lite_lag) {     // (no plain lite rows means no trade is outside list_nonlite)
            rest_set = use_lite_lag ? S_NONLITE_B : S_NONLITE;
            rest_n = use_lite_lag ? tc.n_nonlite_b : tc.n_nonlite;
        }
        if (lite_elsewhere && !plan.knot) push(F_LITE, S_LITE, tc.lite_units, lite_blocks(tc.lite_units));
        if (use_lite_lag && !plan.knot_lag) push(F_LITE_LAG, S_LITE_LAG, tc.lite_lag_units, lite_blocks(tc.lite_lag_units, true));
        if (rest_n > 0) {
            const size_t lds = wide_kernel_lds_bytes(cv.K, cv.Kc, cv.wide_nch, want_gamma);
            const int threads = wide_kernel_threads(cv.wide_nch, want_gamma);
            push(F_WIDE, rest_set, rest_n, blocks_for(rest_n, threads / 64, static_cast<int64_t>(n_cu) * wide_kernel_blocks_per_cu(lds, threads)));
        }
        if (static_cast<size_t>(plan.total_blocks) * wide_partial_doubles(cv.wide_nch) > static_cast<size_t>(max_blocks) * kAggStride)
            plan.error = "grid exceeds scratch";
        knot_launch();
        return plan;
    }
    if (cv.T > 1) {
        // ... or once per pair of pillar tiles (tile_i <= tile_j): each launch writes its tile of the ladders
        plan.tiled = true;
        const int T = cv.T;
        const int set = rest_of_knot;
        const int64_t items = rest_of_knot_n;
        const int blocks = blocks_for(items, kGeneralThreads / 64, static_cast<int64_t>(n_cu) * 4);
        for (int tj = 0; tj < T; ++tj)
            for (int ti = 0; ti <= tj; ++ti) {
                if (!want_gamma && ti != tj) continue;               // PV / delta live on the diagonal tiles
                if (!want_delta && tj > 0) continue;                 // PV alone: tile (0, 0) has it
                push(F_TILED, set, items, blocks, ti, tj);
            }
        // (every tile launch is followed by its own reduction: the launches of a pass share one range of the scratch)
        for (Launch& L : plan.launches) L.first_block = 0;
        plan.total_blocks = plan.launches.empty() ? 0 : blocks;
        if (plan.total_blocks > max_blocks) plan.error = "grid exceeds scratch";
        knot_launch();
        return plan;
    }

    // Up to 32 pillars.  With GAMMA: trades without payment lag take the fast kernel when the curve has the packed layout
    // (more than 32 coupons per leg: chains of rows), payment-lag / weighted trades its payment-lag variant (log-linear
    // schemes, even pillar count), the general kernel what is left.  Without GAMMA: the lite kernel takes the trades of its
    // two tables, the chained fast kernel (packed layout) or the general kernel the rest.
    const bool use_fast = cv.packed_ok != 0;
    const bool use_lite = !want_gamma && lite_fits && tc.lite_units > 0;
    const bool lite_elsewhere = use_lite || plan.knot;
    const bool use_lite_lag = (!want_gamma && lite_fits && tc.lite_lag_units > 0) || plan.knot_lag;   // (priced elsewhere)
    const bool use_lag = want_gamma && use_fast && (tc.lagged_rows > 0 || tc.lagged_chained_rows > 0) && tc.lag_scratch &&
                         log_linear && cv.P % 2 == 0 && tc.lagged_chained_blocks <= tc.lag_blocks && !plan.knot_lag;
    int64_t rows = tc.rows, chained = tc.chained_rows;
    int general_set = S_ALL;
    int64_t general_n = n;
    if (!want_gamma && (lite_elsewhere || use_lite_lag)) {
        if (lite_elsewhere) { rows = 0; chained = 0; }        // the lite table holds the trades of both 32-slot row tables
        if (use_fast) {                                       // (without lite rows: long trades keep their chained rows)
            general_set = use_lite_lag ? S_GENERAL_B : S_GENERAL;
            general_n = use_lite_lag ? tc.n_general_b : tc.n_general;
        } else {                                              // no packed layout: long trades join the general list
            rows = 0; chained = 0;
            general_set = (lite_elsewhere && !use_lite_lag) ? S_NONLITE : S_NONLITE_B;
            general_n = (lite_elsewhere && !use_lite_lag) ? tc.n_nonlite : tc.n_nonlite_b;
        }
    } else if (use_fast) {
        general_set = plan.knot_lag ? S_GENERAL_B : (use_lag ? S_REST : S_GENERAL);
        general_n = plan.knot_lag ? tc.n_general_b : (use_lag ? tc.n_rest : tc.n_general);
        if (plan.knot) { rows = 0; chained = 0; }
    } else {
        rows = 0; chained = 0;                                // the general kernel walks every trade no knot pass takes
        general_set = rest_of_knot; general_n = rest_of_knot_n;
    }
    if (use_lite && !plan.knot) push(F_LITE, S_LITE, tc.lite_units, lite_blocks(tc.lite_units));
    if (rows > 0) {
        const size_t lds = fast_kernel_lds_bytes(cv, want_gamma);
        const int per_cu = static_cast<int>(std::max<size_t>(1, std::min<size_t>(2, kLds / lds)));
        const int64_t units = (rows + fast_kernel_groups() - 1) / fast_kernel_groups();
        push(F_FAST, S_ROWS, rows, blocks_for(units, kFastThreads / 64, static_cast<int64_t>(n_cu) * per_cu));
    }
    if (chained > 0) push(F_FAST_CHAINED, S_CHAINED, chained, tc.chained_blocks);        // the chains are laid out for this grid
    if (general_n > 0) {
        const int threads = general_kernel_threads(cv, want_gamma);     // 512: LDS-resident convexity rows
        push(F_GENERAL, general_set, general_n,
             blocks_for(general_n, threads / 64, static_cast<int64_t>(n_cu) * (threads == kGeneralThreads ? 4 : 2)));
    }
    if (use_lag && tc.lagged_rows > 0) {
        const int64_t units = (tc.lagged_rows + fast_kernel_groups() - 1) / fast_kernel_groups();
        push(F_FAST_LAG, S_LAGGED, tc.lagged_rows, blocks_for(units, fast_kernel_threads(true) / 64, std::min(tc.lag_blocks, n_cu)));
    }
    if (use_lite_lag && !plan.knot_lag) push(F_LITE_LAG, S_LITE_LAG, tc.lite_lag_units, lite_blocks(tc.lite_lag_units, true));
    if (use_lag && tc.lagged_chained_rows > 0) push(F_FAST_LAG_CHAINED, S_LAGGED_CHAINED, tc.lagged_chained_rows, tc.lagged_chained_blocks);
    if (plan.total_blocks > max_blocks) plan.error = "grid exceeds scratch";
    knot_launch();
    return plan;
}

// ------------------------------------------------------------------------------------------------------------------
// Classes of the trades of a batch: which table or list each trade lands in (host side of adr_trades_upload).
// ------------------------------------------------------------------------------------------------------------------
constexpr int64_t kMaxChain = 12;                                       // rows per trade in the chained table: legs of up to 384 coupons
constexpr int64_t kMaxChainLag = kLagScratchNodes / kRowSlots;          // ... payment-lag legs: 128 (the variant's per-trade stash)
static const int64_t kLiteRowBuckets[kLiteSegments] = {1, 2, 3, 4, 6, 8, 12, 16, 26};   // lite rows per trade, rounded up (26 x 15 >= 384)

struct TradeClasses {
    std::vector<int32_t> list_fast, list_long, list_general, list_lagged, list_lagged_long, list_rest;
    std::vector<int32_t> seg_plain[kLiteSegments], seg_lag[kLiteSegments], nonlite, nonlite_b, general_b;
    int64_t lite_units = 0, lite_lag_units = 0;
};

// lagged_of[t] != 0: a coupon of trade t accrues to a date other than its payment date, or carries a weight != 1
inline void classify_trades(int64_t n, const int64_t* fix_off, const int64_t* flt_off, const uint8_t* lagged_of, TradeClasses& out) {
    auto rows_of = [&](int64_t t) {
        const int64_t m = std::max(flt_off[t + 1] - flt_off[t], fix_off[t + 1] - fix_off[t]);
        return std::max<int64_t>(1, (m + kRowSlots - 1) / kRowSlots);
    };
    auto lite_bucket = [&](int64_t t) {
        const int64_t m = std::max(flt_off[t + 1] - flt_off[t], fix_off[t + 1] - fix_off[t]);
        const int64_t rows = std::max<int64_t>(1, (m + kLiteCoupons - 1) / kLiteCoupons);
        int b = 0;
        while (b < kLiteSegments && kLiteRowBuckets[b] < rows) ++b;
        return b;
    };
    std::vector<char> lite_lag(static_cast<size_t>(n), 0);
    for (int64_t t = 0; t < n; ++t) {
        const int64_t rows = rows_of(t);
        const bool lagged = lagged_of[static_cast<size_t>(t)] != 0;
        const bool general = rows > kMaxChain || lagged;
        (general ? out.list_general : rows > 1 ? out.list_long : out.list_fast).push_back(static_cast<int32_t>(t));
        if (general) (rows == 1 ? out.list_lagged : rows <= kMaxChainLag ? out.list_lagged_long : out.list_rest).push_back(static_cast<int32_t>(t));
        // lite tables: plain = the trades of the 32-slot row tables, one-row and chained (legs of up to 384 coupons = 26 lite
        // rows of 15); with payment lag / weights: as many rows as the buckets allow
        const int bucket = lite_bucket(t);
        if (!general) { out.seg_plain[kLiteSegments - 1 - bucket].push_back(static_cast<int32_t>(t)); continue; }
        out.nonlite.push_back(static_cast<int32_t>(t));
        if (lagged && bucket < kLiteSegments) {
            out.seg_lag[kLiteSegments - 1 - bucket].push_back(static_cast<int32_t>(t));
            lite_lag[static_cast<size_t>(t)] = 1;
        } else {
            out.nonlite_b.push_back(static_cast<int32_t>(t));
        }
    }
    for (int32_t t : out.list_general) if (!lite_lag[static_cast<size_t>(t)]) out.general_b.push_back(t);
    constexpr int G = 64 / kLiteSlots;
    for (int k = 0; k < kLiteSegments; ++k) {
        out.lite_units += (static_cast<int64_t>(out.seg_plain[k].size()) + G - 1) / G;
        out.lite_lag_units += (static_cast<int64_t>(out.seg_lag[k].size()) + G - 1) / G;
    }
}

}  // namespace route
}  // namespace adr
