// Curve lookup shared by the fast and the lite kernels and by the projection kernels: InterpolatorAd.simple_interpolate
// (cavour/market/curves/interpolator_ad.py:186-249) for one query time, in weight form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve_tables.hpp"

namespace adr {

struct Lookup {
    int ka, kb;        // compact knots
    double ba, bb;     // D = exp(ba*L[ka] + bb*L[kb]) (LINEAR_FWD_RATES: D = ba*d[ka] + bb*d[kb]); bb == 0: single knot
};

// InterpolatorAd.simple_interpolate for one time (interpolator_ad.py:210-243) in weight form.
// ``C`` exposes the search arrays (any address space): x[K], lut[n_lut][2], first_of[K], compact_of[K], inv_x[Kc],
// and K, n_lut, method (1 FLAT_FWD_RATES, 4 LINEAR_ZERO_RATES; 2 LINEAR_FWD_RATES gets the plain linear weights of
// FLAT_FWD_RATES here and the caller applies them to the discount factors instead of their logarithms).
// INV_DX: ``C`` also has inv_dx[K] (1 / (x[i] - x[i-1]), 0 where jnp.interp's dx guard applies) and the weight is a
// multiplication instead of a division (a double-precision divide is about twenty vector instructions; the weight
// differs from the divided one by an ulp at most, the discount factor by ~1e-16 relative).
// j = index of the first knot later than t (K when there is none): the bucketed binary search.
template <class C>
__device__ __forceinline__ int curve_first_later(const C& c, double t) {
    const double tb = t * kLutPerYear;
    const int bucket = tb > 0.0 ? (tb < static_cast<double>(c.n_lut) ? static_cast<int>(tb) : c.n_lut - 1) : 0;
    int lo = c.lut[2 * bucket], hi = c.lut[2 * bucket + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c.x[mid] > t) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// true when j IS the first knot later than t (x[j-1] <= t < x[j]): a neighbouring time's result can then be reused
// instead of searching again - exactly the same j, hence exactly the same lookup.
template <class C>
__device__ __forceinline__ bool curve_first_later_is(const C& c, double t, int j) {
    return (j <= 0 || c.x[j - 1] <= t) && (j >= c.K || c.x[j] > t);
}

// The lookup proper, given j = curve_first_later(c, t).
template <bool INV_DX = false, class C>
__device__ __forceinline__ Lookup curve_lookup_at(const C& c, double t, int j) {
    const int K = c.K;
    // nearest knot; the first of equal candidates wins, the lower one on a distance tie (argmin)
    double best_dist = 1e300;
    int best = 0;
    if (j > 0) { best = c.first_of[j - 1]; best_dist = fabs(t - c.x[j - 1]); }
    if (j < K) {
        const double dh = fabs(t - c.x[j]);
        if (dh < best_dist) { best_dist = dh; best = j; }
    }
    Lookup r;
    if (best_dist < 1e-10) {            // exact grid point: that knot's DF, gradient to that knot only
        r.ka = r.kb = c.compact_of[best]; r.ba = 1.0; r.bb = 0.0;
        return r;
    }
    const double tau = t + 1e-12;
    const bool lzr = c.method == 4;
    if (tau < c.x[0] || tau > c.x[K - 1]) {   // jnp.interp is constant outside the knot range
        r.ka = r.kb = c.compact_of[tau < c.x[0] ? 0 : K - 1];
        r.bb = 0.0;
        r.ba = lzr ? t * c.inv_x[r.ka] : 1.0;
        return r;
    }
    // no knot lies in (t, t + 1e-12] (it would have snapped), so searchsorted(tau, 'right') == j
    const int i = min(max(j, 1), K - 1);
    const double xa = c.x[i - 1], xb = c.x[i];
    const double dx = xb - xa;
    double w;
    if constexpr (INV_DX) w = (tau - xa) * c.inv_dx[i];
    else w = (fabs(dx) <= 0x1p-104) ? 0.0 : (tau - xa) / dx;           // jnp.interp: fp[i-1] when dx ~ 0
    r.ka = c.compact_of[i - 1];
    r.kb = c.compact_of[i];
    if (lzr) {
        r.ba = t * (1.0 - w) * c.inv_x[r.ka];
        r.bb = t * w * c.inv_x[r.kb];
    } else {
        r.ba = 1.0 - w;
        r.bb = w;
    }
    return r;
}

template <bool INV_DX = false, class C>
__device__ __forceinline__ Lookup curve_lookup(const C& c, double t) {
    return curve_lookup_at<INV_DX>(c, t, curve_first_later(c, t));
}

}  // namespace adr
