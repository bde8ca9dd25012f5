// Device-side data layout and kernel launchers shared by kernels.hip and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve_tables.hpp"

namespace adr {

constexpr int kBlockThreads = 256;                                   // 4 wavefronts, one trade each
constexpr int kAggStride = 1 + kPillarPad + kPillarPad * kPillarPad; // padded [pv, delta, gamma] record

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
    int64_t n;
    const TradeHeader* header;   // [n]
    const double* fix_tp;        // [sum n_fix]
    const double* fix_pay;
    const double* flt_tp;        // [sum n_flt]
    const double* flt_ts;
    const double* flt_te;
    const double* flt_alpha;
};

// Curve tables in HBM (copied to LDS by every block, except lc_lanes which streams from L2).
struct CurveDev {
    int K, Kc, P, method;
    const double* x;             // [K]
    const double* log_df;        // [Kc]
    const double* inv_x;         // [Kc]
    const double* lj;            // [Kc][32]
    const double* lc_lanes;      // [Kc][64][16], null without gamma
    const int16_t* first_of;     // [K]
    const int16_t* compact_of;   // [K]
};

struct OutputsDev {
    double* pv;              // [n] or null
    double* delta;           // [n*P] or null
    double* gamma;           // [n*P*P] or null
    double* block_partials;  // [grid][kAggStride] or null
};

size_t price_kernel_lds_bytes(int K, int Kc);
hipError_t set_price_kernel_lds_limit(size_t bytes);
hipError_t launch_price(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, bool want_delta,
                        bool want_gamma, int n_blocks, hipStream_t stream);
hipError_t launch_reduce_partials(const double* partials, int n_blocks, int P, double* agg, hipStream_t stream);

}  // namespace adr
