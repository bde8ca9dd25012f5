#include "curve_tables.hpp"

#include <algorithm>
#include <cmath>

namespace adr {

std::string build_curve_tables(int K, int P, const double* times, const double* dfs, const double* jac,
                               const double* hess, CurveTables& out) {
    if (K < 2) return "curve needs at least two knots";
    if (P < 1 || P > kPillarPad) return "pillar count must be in 1.." + std::to_string(kPillarPad);
    if (K > 32767) return "too many knots (int16 index tables)";
    if (!times || !dfs || !jac) return "times, dfs and jac must not be null";
    for (int k = 0; k < K; ++k) {
        if (!(dfs[k] > 0.0) || !std::isfinite(dfs[k])) return "discount factors must be positive and finite";
        if (!std::isfinite(times[k])) return "knot times must be finite";
        if (k > 0 && times[k] < times[k - 1]) return "knot times must be non-decreasing";
    }

    out = CurveTables();
    out.K = K;
    out.P = P;
    out.has_hess = hess != nullptr;
    out.x.assign(times, times + K);
    out.first_of.resize(K);
    out.compact_of.assign(K, -1);

    // runs of exactly equal times: keep the first and the last knot of each run
    for (int k = 0; k < K; ++k) {
        out.first_of[k] = (k > 0 && times[k] == times[k - 1]) ? out.first_of[k - 1] : k;
    }
    for (int k = 0; k < K; ++k) {
        const bool first = out.first_of[k] == k;
        const bool last = (k == K - 1) || (times[k + 1] != times[k]);
        if (first || last) {
            out.compact_of[k] = static_cast<int32_t>(out.knot_index.size());
            out.knot_index.push_back(k);
        }
    }
    const int Kc = out.Kc = static_cast<int>(out.knot_index.size());

    out.log_df.resize(Kc);
    out.inv_x.resize(Kc);
    out.lj.assign(static_cast<size_t>(Kc) * kPillarPad, 0.0);
    if (out.has_hess) {
        out.lc.assign(static_cast<size_t>(Kc) * P * P, 0.0);
        out.lc_lanes.assign(static_cast<size_t>(Kc) * 64 * kGammaPerLane, 0.0);
    }

    for (int c = 0; c < Kc; ++c) {
        const int k = out.knot_index[c];
        const double d = dfs[k];
        out.log_df[c] = std::log(d);
        out.inv_x[c] = 1.0 / std::max(times[k], 1e-15);
        double* ljrow = &out.lj[static_cast<size_t>(c) * kPillarPad];
        for (int p = 0; p < P; ++p) ljrow[p] = jac[static_cast<size_t>(k) * P + p] / d;
        if (!out.has_hess) continue;
        const double* hk = hess + static_cast<size_t>(k) * P * P;
        double* lck = &out.lc[static_cast<size_t>(c) * P * P];
        for (int p = 0; p < P; ++p)
            for (int q = 0; q < P; ++q) lck[p * P + q] = hk[p * P + q] / d - ljrow[p] * ljrow[q];
        double* lanes = &out.lc_lanes[static_cast<size_t>(c) * 64 * kGammaPerLane];
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < kGammaPerLane; ++e) {
                const int r = gamma_row(lane, e), q = gamma_col(lane, e);
                lanes[lane * kGammaPerLane + e] = (r < P && q < P) ? lck[r * P + q] : 0.0;
            }
    }
    return std::string();
}

}  // namespace adr
