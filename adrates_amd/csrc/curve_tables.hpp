// Host-side construction of the on-chip curve tables.
//
// Input is exactly what the reference keeps in its curve cache
// (cavour/market/position/engine.py:2405-2411): knot times, knot discount factors and their first and
// second derivatives w.r.t. the par rates.  Output is the log-space form the kernels consume
// ("Equivalent log-space form"):
//
//   L_k   = ln d_k
//   LJ_k  = J_k / d_k                               (d ln d_k / d r)
//   LC_k  = C_k / d_k - J_k J_k^T / d_k^2           (d2 ln d_k / d r2)
//
// restricted to the knots a query of InterpolatorAd.simple_interpolate
// (cavour/market/curves/interpolator_ad.py:186-249) can ever reference: with duplicate knot times, the
// left neighbour of a query is always the LAST knot of a run of equal times, the right neighbour and
// every snap target the FIRST one.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace adr {

constexpr int kPillarPad = 32;                                  // ladders are padded to this on chip
constexpr int kGammaPerLane = kPillarPad * kPillarPad / 64;     // 16 gamma entries per lane (4x4 block)

struct CurveTables {
    int K = 0;    // knots of the caller's grid
    int P = 0;    // pillars
    int Kc = 0;   // knots kept
    bool has_hess = false;
    std::vector<double> x;             // [K]   knot times (full grid, for the search)
    std::vector<int32_t> first_of;     // [K]   first index of the run of equal times containing k
    std::vector<int32_t> compact_of;   // [K]   row of knot k in the compact tables, -1 if unreachable
    std::vector<int32_t> knot_index;   // [Kc]  inverse of compact_of
    std::vector<double> log_df;        // [Kc]
    std::vector<double> inv_x;         // [Kc]  1 / max(x_k, 1e-15)   (linear-zero-rate weights)
    std::vector<double> lj;            // [Kc][kPillarPad], zero padded
    std::vector<double> lc;            // [Kc][P][P] row-major (plain layout, for checking)
    std::vector<double> lc_lanes;      // [Kc][64][16] lane-major layout read by the gamma kernel
};

// Row/column of gamma entry e (0..15) held by lane l (0..63): a 4x4 block at (4*(l/8), 4*(l%8)).
inline int gamma_row(int lane, int e) { return 4 * (lane >> 3) + (e >> 2); }
inline int gamma_col(int lane, int e) { return 4 * (lane & 7) + (e & 3); }

// Returns an empty string on success, otherwise the reason the inputs were rejected.
std::string build_curve_tables(int K, int P, const double* times, const double* dfs, const double* jac,
                               const double* hess, CurveTables& out);

}  // namespace adr
