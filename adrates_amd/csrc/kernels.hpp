// Device-side data layout and kernel launchers shared by the .hip files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve_tables.hpp"

#ifndef ADR_FAST_THREADS
#define ADR_FAST_THREADS 768
#endif
#ifndef ADR_OUT_PRIO
#define ADR_OUT_PRIO 3      // fast kernel: wave priority during the output phase (next unit's loads, expansion, stores): -2 % on the bench pass
#endif
#ifndef ADR_WALK_PRIO
#define ADR_WALK_PRIO 3     // ... during the node walk outside the rank-one update (record, Jacobian rows, v, hand-off: the
                            // latency chain of a node); with the output phase at 3 and the rest at 0 / 1: another -3 %
#endif
#ifndef ADR_LITE_SWEEP_PRIO
#define ADR_LITE_SWEEP_PRIO 1     // lite kernel: wave priority during the entry sweeps (-3 % on the PV + delta pass)
#endif
#ifndef ADR_LITE_OUT_PRIO
#define ADR_LITE_OUT_PRIO 0
#endif
#ifndef ADR_BUILD_PRIO
#define ADR_BUILD_PRIO 0    // ... during folding, lookups and exponentials
#endif
#ifndef ADR_RANK_PRIO
#define ADR_RANK_PRIO 1     // ... inside the rank-one update (the gathers and FMAs: the walk's throughput part)
#endif
#ifndef ADR_FAST_BATCH
#define ADR_FAST_BATCH 4    // packed entries whose LDS operands are fetched together
#endif

namespace adr {

constexpr int kAggStride = 1 + kPillarPad + kPillarPad * kPillarPad; // padded [pv, delta, gamma] record
constexpr int kGeneralThreads = 256;                                 // general kernel: 4 wavefronts per block
constexpr int kGeneralLdsThreads = 512;                              // ... 8 with LDS-resident convexity rows (1 block per CU)
constexpr int kFastThreads = ADR_FAST_THREADS;
constexpr int kRowSlots = 32;                                        // cash-flow slots per row of the fast table
constexpr int kLagStashDoubles = 34;                                 // payment-lag variant: {v[32], omega, pad} per special node
constexpr int kLagScratchNodes = 128;                               // ... special nodes a trade can leave: one per coupon, 4 rows of 32
// lite kernel (kernels_lite.hip): 4 trades per wavefront, rows of 16 slots = 15 coupons + a spare lane
constexpr int kLiteThreads = 512;
constexpr int kLiteWavesPerSimd = 4;                                 // 2 blocks of 8 waves per CU (105 VGPRs)
constexpr int kLiteSlots = 16;
constexpr int kLiteCoupons = 15;
constexpr int kKnotBand = 16;                                        // aggregate-only mode, payment-lag rows: pair bands kept per wave (knots up to 16 apart)
constexpr int kLiteSegments = 9;                                     // distinct row counts per table: 1, 2, 3, 4, 6, 8, 12, 16, 26 rows
                                                                     // (route.hpp) = up to 390 coupons per leg

// Per-trade header, 32 bytes, read once per trade with scalar loads.
struct TradeHeader {
    double notional;
    double spread;
    int32_t flt_begin;   // first float cash flow in the flt_* arrays
    int32_t fix_begin;   // first fixed cash flow in the fix_* arrays
    int16_t n_flt;
    int16_t n_fix;
    int8_t fix_sign;     // +1 receive, -1 pay
    int8_t flt_sign;
    int16_t pad;
};
static_assert(sizeof(TradeHeader) == 32, "TradeHeader must stay 32 bytes");

// Trade arrays in HBM: struct-of-arrays over the flattened (trade x cash flow) axis, so that the
// lanes of a wavefront read consecutive doubles.
struct TradesDev {
    int64_t n;                   // trades in the batch
    const TradeHeader* header;   // [n]
    const double* fix_tp;        // [sum n_fix]
    const double* fix_pay;
    const double* flt_tp;        // [sum n_flt]
    const double* flt_ts;
    const double* flt_te;
    const double* flt_alpha;
    const double* flt_weight;    // per-coupon multiplier of the notional, or null (= 1); general kernel only
    int any_ratio;               // some trade has a coupon with te != tp or a notional multiplier != 1 (ratio nodes)
    // General kernel: the launch covers n_list trades: list[i] when list != null, else trade i.
    const int32_t* list;
    int64_t n_list;
    // Fast kernel: the eligible trades (no payment lag, at most kRowSlots coupons per leg) as a table of
    // n_rows rows sorted by coupon count, kRowSlots zero-padded slots per row and array.
    int64_t n_rows;
    int rows_chained;            // rows are chains of 32-coupon pieces of longer trades (meta bit 18; see the kernel)
    int rows_lagged;             // rows of trades with payment lag / per-coupon notionals (row_te, row_w; LAG kernel variant)
    const double* row_te;        // [n_rows][kRowSlots] accrual end times (lagged rows only)
    const double* row_w;         //                     per-coupon notional multipliers, or null (= 1)
    const double* row_tp;        // [n_rows][kRowSlots] float payment times
    const double* row_ts;        //                     accrual start times
    const double* row_alpha;     //                     accrual fractions
    const double* row_xtp;       //                     fixed payment times
    const double* row_xpay;      //                     fixed payment amounts
    const double* row_notional;  // [n_rows]
    const double* row_spread;    // [n_rows]
    const int32_t* row_meta;     // [n_rows] n_flt | n_fix << 8 | (float leg pays) << 16 | (fixed leg pays) << 17 |
                                 //          (the trade continues in the group's next row) << 18
    const int32_t* row_trade;    // [n_rows] index of the trade in the batch (where its results go)
};

// Row table of the lite kernel: the trades of the 32-slot row table (no payment lag, at most 32 coupons per leg) - or,
// with `te_w`, trades with payment lag / per-coupon notionals of at most 135 coupons per leg -, grouped into
// segments of equal row count (3, 2, 1 rows per trade; inside a segment sorted by coupon count), every segment padded
// to a multiple of 4 trades (one unit = the 4 trades of a wavefront) with empty slots (trade = -1).
struct LiteTrade {                    // 32 bytes: one 16-byte and one 8-byte load per lane
    double notional;
    double spread;
    int32_t meta;                     // n_flt | n_fix << 9 | (float leg pays) << 18 | (fixed leg pays) << 19 (whole trade)
    int32_t trade;                    // index of the trade in the batch, -1 for an empty slot
    int64_t pad;
};
static_assert(sizeof(LiteTrade) == 32, "LiteTrade is read as 32-byte records");

struct LiteRowsDev {
    int64_t n_units;                  // units of 4 trade slots
    int64_t seg_unit0[kLiteSegments]; // first unit of segment k
    int64_t seg_row0[kLiteSegments];  // first row of segment k
    int seg_rows[kLiteSegments];      // rows per trade in segment k
    int n_seg;                        // segments in use (the non-empty ones, longest rows first): entries 0 .. n_seg - 1 above
    // row arrays, interleaved in pairs so that a lane fetches 16 bytes per load (8-byte-per-lane streams reach only
    // 0.5-0.7 of the 16-byte rate on gfx950, MI355X guide)
    const double* tp_ts;              // [n_rows][kLiteSlots][2] float payment time, accrual start time
    const double* al_xtp;             // [n_rows][kLiteSlots][2] accrual fraction, fixed payment time
    const double* xpay;               // [n_rows][kLiteSlots]    fixed payment amount
    const double* te_w;               // [n_rows][kLiteSlots][2] accrual end time, notional multiplier - payment-lag rows only (else null)
    const LiteTrade* slot;            // [4 * n_units] per-trade scalars
};

// The caller's CSR batch on the device (adr_trades_upload) and the work lists of the table builders (trades_build.hip).
struct CsrDev {
    int64_t n;
    const int64_t* fix_off;      // [n + 1]
    const int64_t* flt_off;
    const double *fix_tp, *fix_pay, *flt_tp, *flt_ts, *flt_te, *flt_alpha;
    const double* flt_weight;    // or null
    const double *notional, *spread, *fix_sign, *flt_sign;      // [n]
};
struct RowBuildDev {
    int64_t rows;
    const int32_t* piece_trade;  // [rows] trade of the row, -1 for an empty row
    const int32_t* piece_first;  // [rows] first coupon of the piece, or null (= 0)
    const uint8_t* piece_more;   // [rows] the trade continues in the group's next row, or null
    double *row_tp, *row_ts, *row_alpha, *row_xtp, *row_xpay, *row_te, *row_w, *row_notional, *row_spread;   // row_te / row_w may be null
    int32_t *row_meta, *row_trade;
};
struct LiteBuildDev {
    int64_t rows, n_slots;
    const int32_t* slot_trade;   // [n_slots] trade in the slot, -1 for an empty one
    const int32_t* row_slot;     // [rows]
    const uint8_t* row_piece;    // [rows] which 15-coupon piece of the slot's trade
    double *tp_ts, *al_xtp, *xpay, *te_w;    // te_w may be null
    LiteTrade* slot;
};
hipError_t launch_build_headers(const CsrDev& csr, TradeHeader* out, hipStream_t stream);
hipError_t launch_build_rows(const CsrDev& csr, const RowBuildDev& rb, hipStream_t stream);
hipError_t launch_build_lite(const CsrDev& csr, const LiteBuildDev& lb, hipStream_t stream);

// Curve tables in HBM.
struct CurveDev {
    int K, Kc, P, method;
    int T;                       // pillar tiles of 32 (curve_tables.hpp); more than one: general kernel, one launch per tile pair
    int tile_i, tile_j;          // the tile pair (tile_i <= tile_j) a general-kernel launch works on; 0, 0 for P <= 32
    const double* x;             // [K]
    const double* log_df;        // [Kc]
    const double* inv_x;         // [Kc]
    const int16_t* lut;          // [n_lut][2] knot-search table (curve_tables.hpp)
    int n_lut;
    const int16_t* first_of;     // [K]
    const int16_t* compact_of;   // [K]
    // general kernel: dense 32-wide tables
    const double* lj;            // [T][Kc][32]
    const double* lc_lanes;      // [pairs][Kc][64][16], null without gamma
    const unsigned long long* lc_block_mask;  // [pairs][Kc] lanes whose 4x4 block of that tile of LC_k is not structurally zero
    // wide kernel (33-64 pillars, one launch for the whole ladder; curve_tables.hpp, wide layout): valid when wide_nch > 0
    int wide_nch;                // chunks of 128 packed gamma entries (two per lane), 0: no wide tables
    const double* lj64;          // [Kc][64]
    const uint32_t* wide_ent;    // [wide_nch][64] the lane's pair of entries: row | column << 8 | inside-the-triangle bits
    const int32_t* wide_pos;     // [64] position of pillar p in the packing's pillar order
    const int32_t* wide_order;   // [P] its inverse
    const double* lcflat;        // [Kc][wide_nch * 128], null without gamma
    const uint32_t* wide_knot_chunks;   // [Kc] chunks of the knot's row with a structural non-zero
    const uint32_t* wide_store_map;     // [bands][64] packed entries of elements 128 band + 2 lane, + 1 of the P x P matrix
    // fast kernels: packed layout (curve_tables.hpp), valid when packed_ok
    int packed_ok, Pc, pc_pad, Ec, Eu, epg, cpg, hub, Kcore, n_mini;
    int fringe_start;            // first packed entry of the fringe pairs (entries fringe_start .. Eu - 1)
    int n_fringe, fringe_own;    // number of fringe pairs; 1: each sits in the group lane of its first pillar (that lane's own v)
    const double* ljc;           // [Kcore][pc_pad]
    const double* lcc;           // [Kcore][Ec + 1] (last entry of every row is 0), null without gamma
    const MiniKnot* mini;        // [n_mini]
    const int16_t* knot_class;   // [Kc]
    const int16_t* pillar_to_core;  // [32]
    const int16_t* out_map;      // [32*32] packed entry of gamma[r][c], rows 32 wide (aggregate)
    const int16_t* store_map;    // [32*32] packed entry by flat index r*P + c of the caller's matrix; -2 beyond P*P
    int odd_last;                // odd P: packed entry of the last element, stored on its own (curve_tables.hpp); -3 for even P
    const uint8_t* ent_pq;       // [Eu][2]
    const int16_t* core_pos;     // [32*cpg] hub layout only: row position of core entry e
    const int16_t* lcc_pos;      // [32*32] flat index of pair (r, c) for the general kernel's LDS rows: its position in a
                                 //         lcc row (core pairs), Ec + 1 + ordinal (fringe pairs), -1 otherwise
};

struct OutputsDev {
    double* pv;              // [n] or null
    double* delta;           // [n*P] or null
    double* gamma;           // [n*P*P] or null
    double* block_partials;  // [grid][kAggStride] or null
    double* block_partials2; // two-curve launch (XC): the second ladder's block records [grid][kAggStride]
    double* delta2;          // ... its per-trade ladder [n * P2] or null
    double* dump;            // [32*32] sink for the gamma stores of the idle trade slot of a wave's last unit
    double* lag_scratch;     // payment-lag variant: [grid waves][2 groups][kLagScratchNodes][kLagStashDoubles]
    double* knot_partials;   // aggregate-only mode: [grid][record] block sums {pv, w[Kc], D[Kc], P[bands][Kc]} of the knot-space kernel
    double* knot_overflow;   // ... [Kc][Kc] pairs of knots farther apart than the bands (payment-lag rows; zeroed per launch)
    unsigned long long* stamps;  // diagnostic builds only (ADR_STAMPS): [grid*waves][8] cycle sums per phase
};

// Device-side curve builder (curve_build.hip): the scan description of one knot grid plus the gather maps of
// its table layout (both independent of the par rates), and where a batch of scenarios' tables go.
struct CurveBuildPlanDev {
    int K, P, Kc;
    const double* acc;            // [K] accrual fraction of the knot's coupon period
    const int32_t* pillar;        // [K] calibration swap (= par rate) of the knot
    const int32_t* prev_idx;      // [K] knot holding the PV01 the knot builds on, -1 for none
    const int32_t* knot_index;    // [Kc] knots a query can reach (curve_tables.hpp)
    int packed_ok, Pc, pc_pad, Ec, Kcore, n_mini;
    const int16_t* knot_class;    // [Kc]
    const int32_t* core_pillars;  // [Pc]
    const uint8_t* lcc_pq;        // [Ec][2] pillars of the pair at each position of a lcc row
    // 33-64 pillars: the wide layout of the base curve (curve_tables.hpp); no tiled tables are written
    int wide_nch;                 // chunks per row of the packed triangle, 0 for P <= 32
    const uint8_t* wide_pq;       // [wide_nch * 128][2] pillars of the packed entry, 255 where the slot is padding
    int dpv_global;               // the PV01 gradients go through `dpv_scratch` instead of LDS (K * P doubles do not fit)
};

struct CurvePackOut {             // every array has a leading scenario axis
    double* log_df;               // [S][Kc]
    double* lj;                   // [S][Kc][32]
    double* lc_lanes;             // [S][Kc][64][16] or null
    double* ljc;                  // [S][Kcore + 1][pc_pad], zero-initialised
    double* lcc;                  // [S][Kcore + 1][Ec + 1], zero-initialised, or null
    MiniKnot* mini;               // [S][n_mini], pillar / entry fields pre-filled
    double* lj64;                 // [S][Kc][64] (wide layout)
    double* lcflat;               // [S][Kc][wide_nch * 128], zero-initialised, or null
};

size_t bootstrap_kernel_lds_bytes(int K, int P, bool dpv_global = false);
hipError_t launch_curve_build(const CurveBuildPlanDev& plan, int n_scen, const double* rates_dev, double* dfs,
                              double* jac, double* hess, double* d2pv_scratch, double* dpv_scratch,
                              const CurvePackOut& out, hipStream_t stream);

size_t general_kernel_lds_bytes(int K, int Kc, bool two_tiles = false);
int general_kernel_threads(const CurveDev& cv, bool gamma);          // block size of the variant launch_price_general picks
size_t fast_kernel_lds_bytes(const CurveDev& cv, bool gamma, bool lagged = false);
size_t general_lds_kernel_lds_bytes_for(int K, int Kc, int Kcore, int Ec, int n_mini, int n_lut, bool gamma);
bool general_lds_rows_fit(size_t lds_bytes, int Ec, int n_fringe);
int fast_kernel_threads(bool lagged);
int fast_kernel_groups();
size_t fast_kernel_lag_scratch_bytes(int n_blocks);
hipError_t set_kernel_lds_limits(size_t general_bytes, size_t fast_bytes);
hipError_t launch_price_general(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, bool want_delta,
                                bool want_gamma, int n_blocks, hipStream_t stream);
// wide variants of the general kernel: curves of 33-64 pillars, every trade, all three schemes, one launch
size_t wide_kernel_lds_bytes(int K, int Kc, int nch, bool gamma);
int wide_kernel_threads(int nch, bool gamma);
int wide_kernel_blocks_per_cu(size_t lds_bytes, int threads);
hipError_t launch_price_wide(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, bool want_delta,
                             bool want_gamma, int n_blocks, hipStream_t stream);
// block partials of the wide kernel ([n_blocks][wide_partial_doubles(nch)]) -> agg[1 + P + P*P], fixed order
int wide_partial_doubles(int nch);
hipError_t launch_reduce_wide(const CurveDev& cv, const double* partials, int n_blocks, bool has_delta, bool has_gamma,
                              double* agg, hipStream_t stream);
hipError_t launch_price_fast(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, bool want_delta,
                             bool want_gamma, int n_blocks, hipStream_t stream);
hipError_t launch_curve_df(const CurveDev& cv, int64_t n, const double* t_dev, double* df_dev, int n_cu, hipStream_t stream);
size_t lite_kernel_lds_bytes(const CurveDev& cv, bool delta, bool lag = false);
int lite_kernel_threads(const CurveDev& cv, bool delta, bool lag);
hipError_t launch_price_lite(const CurveDev& cv, const LiteRowsDev& tr, const OutputsDev& out, bool want_delta,
                             int n_blocks, hipStream_t stream);
// the foreign leg of cross-currency swaps on two curves (payment-lag rows; cv: foreign OIS curve, cx: XCCY curve): PV and two
// delta ladders (out.delta on cv's pillars, out.delta2 on cx's) from one read of the coupons
int lite_xc_kernel_threads();
size_t lite_xc_kernel_lds_bytes(const CurveDev& cv, const CurveDev& cx);
hipError_t launch_price_lite_xc(const CurveDev& cv, const CurveDev& cx, const LiteRowsDev& tr, const OutputsDev& out, int n_blocks,
                                hipStream_t stream);
// Aggregate-only mode (kernels_lite.hip KNOT instantiations + kernels_knot.hip): the book's knot-space sums from the rows of
// the lite table, reduced over the blocks in a fixed order, projected once to the pillar ladders and ADDED to agg
// ([pv, delta[P], gamma[P*P]]; the caller has zeroed it or another kernel family's reduction has written it).
size_t knot_kernel_lds_bytes(const CurveDev& cv, bool gamma, bool lag = false);
int knot_record_doubles(const CurveDev& cv, bool lag);      // doubles per block record
int knot_kernel_threads();
hipError_t launch_price_knot(const CurveDev& cv, const LiteRowsDev& tr, const OutputsDev& out, bool want_gamma, int n_blocks,
                             hipStream_t stream);
// partials [n_blocks][record] -> reduced [record] (fixed order), then agg += projection; `reduced` is scratch; bands: 1, or
// kKnotBand with `overflow` ([Kc][Kc], may hold pairs beyond the bands) for the payment-lag rows
hipError_t launch_knot_project(const CurveDev& cv, const double* partials, int n_blocks, double* reduced, bool want_delta,
                               bool want_gamma, int bands, const double* overflow, double* agg, hipStream_t stream);
// has_gamma == false: the gamma part of the partials was not written (no gamma requested); agg's gamma part is zeroed
// (tile_i, tile_j): the pillar tile pair the partials belong to (0, 0 for P <= 32); off-diagonal tiles are also written
// transposed; pv comes from tile (0, 0), delta from the diagonal tiles
hipError_t launch_reduce_partials(const double* partials, int n_blocks, int P, bool has_gamma, double* agg,
                                  hipStream_t stream, int tile_i = 0, int tile_j = 0);

}  // namespace adr
