// CDNA4 (gfx950) GENERAL kernel: PV, pillar delta ladder and pillar x pillar gamma of OIS trades with no
// assumption on the curve's structure or the trade's cash flows.  It serves the trades the fast kernel
// (kernels_fast.hip) does not take: coupons with a payment lag (ratio terms) and curves without the sparse
// pillar-support structure.  Dense 32-wide tables: LJ in LDS, LC tiles streamed from L2.
//
// What is computed (reference: cavour/market/position/engine.py:2414-2448 fixed leg, :2639-2728 float
// leg, :2541-2576 / :2899-2934 Greeks assembly; curve lookups: cavour/market/curves/
// interpolator_ad.py:186-249):
//
//   PV      = s_f * sum_j pay_j D(tp_j) [tp_j > 0]
//           + s_l * sum_j N ((D(ts_j)/D(te_j) - 1)/a_j + spread) a_j D(tp_j) [tp_j >= 0]
//   delta_p = 1e-4 dPV/dr_p,   gamma_pq = 1e-8 d2PV/dr_p dr_q
//
// Every discount factor is D(t) = exp(ba*L[ka] + bb*L[kb]) with L = ln(knot DF) and (ka, kb) the knots
// bracketing t (or a single snapped knot), so every PV term is w = c*exp(sum_i b_i L[k_i]) and
//   dPV/dr   = sum_terms w * v,              v = sum_i b_i LJ[k_i]
//   d2PV/dr2 = sum_terms w * (v v^T + sum_i b_i LC[k_i])
// (hand-rolled reverse sweep for v / delta, forward-over-reverse for gamma).
//
// Terms that share a time share D and v, so before any exponential is taken the cash flows of a trade
// are folded into "nodes" (time, coefficient): with no payment lag (te == tp) a float coupon is
// N*(D(ts) - (1 - spread*a) D(tp)), its start node coincides with the previous coupon's payment node,
// and the fixed coupon on the same date adds to the same node.  A standard OIS with M coupons becomes
// M + 1 nodes instead of 4M discount factors.  Coupons with te != tp stay ratio terms (3 times, 6 knots).
//
// Mapping: one 64-lane wavefront per trade.  Lanes = cash flows while nodes are built (coalesced loads
// of the trade's arrays, binary search of the knot times in LDS, exp); then lanes = pillars for delta
// and lane = one 4x4 block of the 32x32 gamma; node data is broadcast with v_readlane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "kernels.hpp"

namespace adr {

namespace {

constexpr int kThreadsL2 = kGeneralThreads;      // convexity tiles streamed from L2: 2 blocks of 4 waves per CU
constexpr int kThreadsLds = kGeneralLdsThreads;  // convexity rows resident in LDS: 1 block of 8 waves per CU
constexpr int kConvSlices = 3;                   // 64-entry slices of a packed convexity row (at most 192 core pairs)
constexpr int kConvStage = 64 * kConvSlices;     // doubles per wave for handing the flat convexity sums to the blocks
// Wide variants with GAMMA stage a trade's packed ladder in LDS (7 / 10 / 17 KB per wave) on its way to the row-major
// matrix: 8 waves per block next to the tables for up to 50 pillars (two waves per SIMD), 4 waves beyond (one per SIMD,
// which may then use the whole register file).
__host__ __device__ constexpr int general_block_threads(bool ldslc, int wide) {
    return ldslc ? kThreadsLds : (wide > 0 && wide <= 10 ? 512 : kThreadsL2);
}
__host__ __device__ constexpr int wide_bands(int nch) { return nch <= 7 ? 14 : (nch <= 10 ? 20 : 32); }   // 41^2, 50^2, 64^2 elements

__device__ __forceinline__ int readlane_i(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }

__device__ __forceinline__ double readlane_d(double x, int lane) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

typedef double nt_pair __attribute__((ext_vector_type(2)));      // operand type of the non-temporal 16-byte stores
__device__ __forceinline__ void nt_store2(double2* dst, double x, double y) {
    nt_pair v; v.x = x; v.y = y;
    __builtin_nontemporal_store(v, reinterpret_cast<nt_pair*>(dst));
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// Curve tables staged in LDS, shared by the block's waves.
struct CurveLds {
    const double* x;        // [K]
    const double* log_df;   // [Kc]
    const double* inv_x;    // [Kc]
    const double* lj;       // [Kc][32] of the row tile, then (off-diagonal tile pairs) [Kc][32] of the column tile
    int lj_off;             // this lane's table: lanes 0-31 build v on the row tile, lanes 32-63 on the column tile
    int col0;               // where the column tile's v sits in the wave's hand-off buffer (0 on diagonal tile pairs)
    bool diag;              // the launch's tile pair is on the diagonal (always, for P <= 32): blocks below it are mirrors
    const int16_t* lut;         // [n_lut][2] knot-search table (curve_tables.hpp)
    int n_lut;
    const int16_t* first_of;    // [K]
    const int16_t* compact_of;  // [K]
    int K;
    int method;
    unsigned went[kWideMaxChunks];   // WIDE: this lane's pair of packed gamma entries per chunk: row a (even) | column b << 8 |
                                     //       (entry 0 inside the triangle) << 16 | (entry 1: row a + 1) << 17, in position space
    int wrow;                        // WIDE: doubles per row of the packed convexity table
    int wpos;                        // WIDE: position of this lane's pillar in the packing's pillar order
};

// One discount-factor lookup.  FLAT_FWD / LINEAR_ZERO: D(t) = exp(ba*L[ka] + bb*L[kb]); ka/kb are rows of the
// compact tables.  LINEAR_FWD_RATES is linear in the knot DFs themselves, D = (1-w) d_a + w d_b: its log-gradient
// is the convex combination rho_a LJ[ka] + rho_b LJ[kb] with rho_a = (1-w) d_a / D, so the same node machinery
// applies with (ba, bb) = (rho_a, rho_b), plus a rank-one correction rho_a rho_b (LJ[ka] - LJ[kb])(...)^T in the
// Hessian of ln D (`kappa` = rho_a rho_b; 0 for the log-linear schemes).
struct Lookup {
    int ka, kb;
    double ba, bb;
    double ln_d;     // ln D(t)
    double kappa;
};

// InterpolatorAd.simple_interpolate for one time (interpolator_ad.py:210-243) in weight form.
__device__ __forceinline__ Lookup curve_lookup(const CurveLds& c, double t) {
    const int K = c.K;
    // j = first knot with x > t, searched inside the index range the time's bucket allows
    const double tb = t * kLutPerYear;
    const int bucket = tb > 0.0 ? (tb < static_cast<double>(c.n_lut) ? static_cast<int>(tb) : c.n_lut - 1) : 0;
    int lo = c.lut[2 * bucket], hi = c.lut[2 * bucket + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c.x[mid] > t) hi = mid; else lo = mid + 1;
    }
    const int j = lo;
    // nearest knot; the first of equal candidates wins, the lower one on a distance tie (argmin)
    double best_dist = 1e300;
    int best = 0;
    if (j > 0) { best = c.first_of[j - 1]; best_dist = fabs(t - c.x[j - 1]); }
    if (j < K) {
        const double dh = fabs(t - c.x[j]);
        if (dh < best_dist) { best_dist = dh; best = j; }
    }
    Lookup r;
    r.kappa = 0.0;
    if (best_dist < 1e-10) {            // exact grid point: that knot's DF, gradient to that knot only
        r.ka = c.compact_of[best]; r.kb = 0; r.ba = 1.0; r.bb = 0.0;
        r.ln_d = c.log_df[r.ka];
        return r;
    }
    const double tau = t + 1e-12;
    const bool lzr = c.method == 4;
    if (tau < c.x[0] || tau > c.x[K - 1]) {   // jnp.interp is constant outside the knot range
        const int k = tau < c.x[0] ? 0 : K - 1;
        r.ka = c.compact_of[k]; r.kb = 0; r.bb = 0.0;
        r.ba = lzr ? t * c.inv_x[r.ka] : 1.0;
        r.ln_d = r.ba * c.log_df[r.ka];
        return r;
    }
    // no knot lies in (t, t + 1e-12] (it would have snapped), so searchsorted(tau, 'right') == j
    const int i = min(max(j, 1), K - 1);
    const double xa = c.x[i - 1], xb = c.x[i];
    const double dx = xb - xa;
    // jnp.interp returns fp[i-1] when |dx| <= spacing(eps) = 2^-104
    const double w = (fabs(dx) <= 0x1p-104) ? 0.0 : (tau - xa) / dx;
    r.ka = c.compact_of[i - 1];
    r.kb = c.compact_of[i];
    if (c.method == 2) {                 // LINEAR_FWD_RATES (interpolator_ad.py:234-235)
        const double da = exp(c.log_df[r.ka]), db = exp(c.log_df[r.kb]);
        const double d = da + w * (db - da);
        r.bb = w * db / d;
        r.ba = 1.0 - r.bb;
        r.kappa = r.ba * r.bb;
        r.ln_d = log(d);
        return r;
    }
    if (lzr) {
        r.ba = t * (1.0 - w) * c.inv_x[r.ka];
        r.bb = t * w * c.inv_x[r.kb];
    } else {
        r.ba = 1.0 - w;
        r.bb = w;
    }
    r.ln_d = fma(r.ba, c.log_df[r.ka], r.bb * c.log_df[r.kb]);
    return r;
}

// Per-wave accumulators of one trade (and, separately, of the wave's running portfolio sums).
template <bool GAMMA, int WIDE = 0>
struct Ladders {
    static constexpr int kGammaRegs = GAMMA ? (WIDE > 0 ? 2 * WIDE : kGammaPerLane) : 1;
    double pv;                 // lane-partial, reduced at the end of the trade
    double delta;              // lane p (and p + 32, duplicated) holds pillar p; WIDE: lane p holds pillar p of 64
    double gamma[kGammaRegs];  // WIDE: packed entries 128 c + 2 lane, 128 c + 2 lane + 1 of chunk c at [2 c], [2 c + 1]
    double conv[GAMMA ? kConvSlices : 1];   // LDS path: sum_k coef_k * LC_k on the packed core pairs, entry lane + 64 s
    __device__ void clear() {
        pv = 0.0; delta = 0.0;
#pragma unroll
        for (int e = 0; e < kGammaRegs; ++e) gamma[e] = 0.0;
#pragma unroll
        for (int e = 0; e < (GAMMA ? kConvSlices : 1); ++e) conv[e] = 0.0;
    }
};

// Where the curve-convexity term sum_i coef_i LC[k_i] comes from.  L2 path: dense 8 KB tiles per knot (any curve).
// LDS path (curves with the packed tables of curve_tables.cpp): the packed core-pair rows the fast kernel uses,
// resident in LDS; the term is linear in LC and independent of the rank-one part, so it is accumulated in the rows'
// own flat layout - lane l adds row[l], row[l + 64], row[l + 128], three conflict-free reads per knot instead of a
// 16-entry gather per lane - and scattered to the 4x4 blocks once per trade.  Short-end knots (at most two pillars)
// carry their one to three numbers in 64-byte records; they go to flat entries behind the core pairs (the fringe
// pairs of the packed layout), which the same scatter serves.
struct ConvLds {
    const double* lcc;          // [Kcore + 1][ec_stride]
    const MiniKnot* mini;       // [n_mini]
    int ec_stride;              // Ec + 1
    const int16_t* flat_of;     // [32*32] flat index of pair (r, c) (CurveDev::lcc_pos; read for short-end knots only)
};

// Add the nodes held by the lanes in `mask` to the trade's ladders.  A node is NK (knot, weight) pairs
// and a coefficient*exp() value `omega`; NK is 2 for plain nodes and 6 for ratio nodes.
// CF: the convexity coefficients of the entries come from `cf` (final values: the caller has summed the weights
// of the nodes that share a knot) instead of omega * b; without CF the argument is ignored.
// LDSLC: the knot words carry the knot's class (curve_tables.hpp, knot_class) in their high half, `conv` is used.
// NCORR (LINEAR_FWD_RATES, LDS-resident convexity rows): the rank-one corrections of the node's NCORR = NK / 2 lookups
// (`add_df_correction`), weight kw[q] on the vector LJ[k[2q]] - LJ[k[2q+1]], ride in the node's own walk - their vectors are
// differences of the Jacobian entries the walk has just read, and they reach the lanes through the idle halves of the
// hand-off buffers (one tile: lanes 32-63 of `vbuf` hold a copy; `stage2` is the wave's trade-end staging area): one walk
// per node instead of one per node and lookup (a payment-lag coupon: two walks instead of six).
template <int NK, bool DELTA, bool GAMMA, bool CF = false, bool LDSLC = false, int NCORR = 0>
__device__ __forceinline__ void add_nodes(unsigned long long mask, const int (&k)[NK], const double (&b)[NK],
                                          double omega, const CurveLds& c, const double* __restrict__ lc_lanes,
                                          const unsigned long long* lc_block_mask,
                                          double* vbuf, int lane, Ladders<GAMMA>& acc, const double (&cf)[NK],
                                          const ConvLds& conv = ConvLds{},
                                          const double (&kw)[NCORR > 0 ? NCORR : 1] = {0.0}, double* stage2 = nullptr) {
    static_assert(NCORR == 0 || (NCORR == NK / 2 && LDSLC && GAMMA), "fused corrections: one per lookup, LDS rows, gamma");
    const int p = lane & 31;
    const int bi = lane >> 3, bj = lane & 7;
    while (mask) {
        const int n = __builtin_ctzll(mask);
        mask &= mask - 1;
        const double om = readlane_d(omega, n);
        int kk[NK], cls[NK];
        double bb[NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int word = readlane_i(k[i], n);
            kk[i] = LDSLC ? (word & 0xffff) : word;
            cls[i] = LDSLC ? (word >> 16) : 0;
            bb[i] = readlane_d(b[i], n);
        }
        double cc[NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) cc[i] = (CF && GAMMA) ? readlane_d(cf[i], n) : om * bb[i];
        double v = 0.0, ljv[NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) { ljv[i] = c.lj[c.lj_off + kk[i] * kPillarPad + p]; v = fma(bb[i], ljv[i], v); }
        if (DELTA) acc.delta = fma(om, v, acc.delta);
        double wq[NCORR > 0 ? NCORR : 1];
        if constexpr (NCORR > 0) {
#pragma unroll
            for (int q = 0; q < NCORR; ++q) wq[q] = readlane_d(kw[q], n);
        }
        if (GAMMA) {
            // hand v[0..31] to every lane through the wave's LDS slot (same-wave LDS ops are ordered)
            // (the DS instructions of one wavefront execute in issue order: a compiler barrier is all the
            // hand-off needs, no wait for the write to retire)
            __builtin_amdgcn_wave_barrier();
            if constexpr (NCORR > 0) {
                const double d0 = ljv[0] - ljv[1];
                vbuf[lane] = lane < 32 ? v : d0;
                if constexpr (NCORR > 1) {
                    const double d1 = ljv[2] - ljv[3];
                    double d2 = 0.0;
                    if constexpr (NCORR > 2) d2 = ljv[4] - ljv[5];
                    stage2[lane] = lane < 32 ? d1 : d2;
                }
            } else
            vbuf[lane] = v;                       // lanes 32-63: the column tile's v (a copy on diagonal tile pairs)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            if constexpr (LDSLC) {
                // curve-convexity part, flat: every lane, three consecutive-address reads per knot
#pragma unroll
                for (int i = 0; i < NK; ++i) {
                    const double coef = cc[i];
                    if (coef == 0.0 || cls[i] == -2) continue;                 // wave-uniform
                    if (cls[i] >= 0) {
                        const double* row = conv.lcc + cls[i] * conv.ec_stride + lane;
#pragma unroll
                        for (int sl = 0; sl < kConvSlices; ++sl)
                            if (lane + 64 * sl < conv.ec_stride) acc.conv[sl] = fma(coef, row[64 * sl], acc.conv[sl]);
                    } else {                                // short-end knot: one to three numbers, on the fringe pairs
                        const MiniKnot& m = conv.mini[-3 - cls[i]];   // that follow the core pairs in the flat layout
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const int pr = m.p[j == 2 ? 1 : 0], pc = m.p[j == 0 ? 0 : 1];     // (p0,p0), (p0,p1), (p1,p1)
                            if (pr < 0 || pc < 0) continue;                     // single-pillar knot: one entry
                            const int at = conv.flat_of[pr * kPillarPad + pc];  // a fringe pair, or a core pair (p1 may be core)
                            if (at < 0) continue;
                            const double add = (lane == (at & 63)) ? coef * m.lc[j] : 0.0;
#pragma unroll
                            for (int sl = 0; sl < kConvSlices; ++sl) acc.conv[sl] += (sl == (at >> 6)) ? add : 0.0;
                        }
                    }
                }
            }
            // gamma is symmetric: only the lanes holding a block on or above the diagonal accumulate (and read
            // LC tiles); the mirror blocks are written from their registers at output time
            if (c.diag && bi > bj) continue;
            double vr[4], vc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { vr[i] = om * vbuf[4 * bi + i]; vc[i] = vbuf[c.col0 + 4 * bj + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) acc.gamma[i * 4 + jx] = fma(vr[i], vc[jx], acc.gamma[i * 4 + jx]);
            if constexpr (NCORR > 0) {
#pragma unroll
                for (int q = 0; q < NCORR; ++q) {
                    const double* src = (q == 0 ? vbuf + 32 : (q == 1 ? stage2 : stage2 + 32));
                    double dr[4], dc_[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { dr[i] = wq[q] * src[4 * bi + i]; dc_[i] = src[4 * bj + i]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jx = 0; jx < 4; ++jx) acc.gamma[i * 4 + jx] = fma(dr[i], dc_[jx], acc.gamma[i * 4 + jx]);
                }
            }
            // curve-convexity part: sum_i om*b_i * LC[k_i]
            if constexpr (!LDSLC)
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const double coef = cc[i];
                // most 4x4 blocks of LC_k are structurally zero (a knot depends on its bootstrap chain's
                // pillars only): the lanes holding such a block skip the tile read
                if (coef != 0.0 && ((lc_block_mask[kk[i]] >> lane) & 1ull)) {
                    const double2* tile = reinterpret_cast<const double2*>(
                        lc_lanes + (static_cast<size_t>(kk[i]) * 64 + lane) * kGammaPerLane);
#pragma unroll
                    for (int e = 0; e < kGammaPerLane / 2; ++e) {
                        const double2 tv = tile[e];
                        acc.gamma[2 * e] = fma(coef, tv.x, acc.gamma[2 * e]);
                        acc.gamma[2 * e + 1] = fma(coef, tv.y, acc.gamma[2 * e + 1]);
                    }
                }
            }
        }
    }
}

// WIDE variants (curves of 33-64 pillars; curve_tables.hpp, wide layout): lane p builds v for pillar p of 64 from the 64-wide
// Jacobian table; the gamma matrix is held as its packed upper triangle, two entries per lane and 128-entry chunk.  The
// rank-one term reads omega v of the lane's two rows (one 16-byte read) and v of its column from the wave's hand-off buffers,
// which hold v in the packing's pillar order; the convexity term adds
// coef * (row of LC_k in the same packed order) straight into the same registers - 16 bytes per lane and chunk from L2,
// and only the chunks the knot's bit mask names (the pillar order of the packing puts a knot's pairs at the front of the
// row: typically one to three chunks of 1 KB per knot instead of a 128-byte tile for each of the wave's 64 lanes).
// One launch covers the whole ladder instead of one launch per pair of 32-pillar tiles.
template <int NK, bool DELTA, bool GAMMA, bool CF, int NCH>
__device__ __forceinline__ void add_nodes_wide(unsigned long long mask, const int (&k)[NK], const double (&b)[NK],
                                               double omega, const CurveLds& c, const double* __restrict__ lcflat,
                                               const unsigned* knot_chunks, double* vbuf, int lane,
                                               Ladders<GAMMA, NCH>& acc, const double (&cf)[NK]) {
    static_assert(NK % 2 == 0, "knots are walked in pairs");
    while (mask) {
        const int n = __builtin_ctzll(mask);
        mask &= mask - 1;
        const double om = readlane_d(omega, n);
        int kk[NK];
        double bb[NK], cc[NK];
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            kk[i] = readlane_i(k[i], n);
            bb[i] = readlane_d(b[i], n);
        }
#pragma unroll
        for (int i = 0; i < NK; ++i) cc[i] = (CF && GAMMA) ? readlane_d(cf[i], n) : om * bb[i];
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < NK; ++i) v = fma(bb[i], c.lj[kk[i] * kWidePad + lane], v);
        if (DELTA) acc.delta = fma(om, v, acc.delta);
        if constexpr (GAMMA) {
            __builtin_amdgcn_wave_barrier();
            vbuf[c.wpos] = v;                       // in position space: the rows of a column are consecutive
            vbuf[kWidePad + c.wpos] = om * v;
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            // Convexity rows of the node's knots, a pair of knots at a time: every chunk either knot has entries in
            // (wave-uniform bit masks) is requested for both, all requests of the pair ahead of the rank-one update (first
            // pair) and of their FMAs.  No register is initialised for a chunk that is not read.
#pragma unroll
            for (int i0 = 0; i0 < NK; i0 += 2) {
                unsigned m = (cc[i0] != 0.0 ? knot_chunks[kk[i0]] : 0u) | (cc[i0 + 1] != 0.0 ? knot_chunks[kk[i0 + 1]] : 0u);
                m = __builtin_amdgcn_readfirstlane(m);
                const double2* ra = reinterpret_cast<const double2*>(lcflat + static_cast<size_t>(kk[i0]) * c.wrow) + lane;
                const double2* rb = reinterpret_cast<const double2*>(lcflat + static_cast<size_t>(kk[i0 + 1]) * c.wrow) + lane;
#ifndef ADR_WIDE_BATCH
#define ADR_WIDE_BATCH 4
#endif
                constexpr int BATCH = ADR_WIDE_BATCH;   // chunks in flight: 8 registers of row data each
#pragma unroll
                for (int c0 = 0; c0 < NCH; c0 += BATCH) {
                    double2 la[BATCH], lb[BATCH];
                    const bool any = ((m >> c0) & ((1u << BATCH) - 1u)) != 0;      // wave-uniform
                    if (any) {
#pragma unroll
                        for (int q = 0; q < BATCH; ++q)
                            if (c0 + q < NCH && ((m >> (c0 + q)) & 1u)) { la[q] = ra[(c0 + q) * 64]; lb[q] = rb[(c0 + q) * 64]; }
                    }
                    if (i0 == 0 && c0 == 0) {       // the rank-one term, under the first requests
#pragma unroll
                        for (int ch = 0; ch < NCH; ++ch) {
                            const unsigned w = c.went[ch];
                            const double2 wa = *reinterpret_cast<const double2*>(vbuf + kWidePad + (w & 0xff));   // omega v at rows a, a + 1
                            const double vb = vbuf[(w >> 8) & 0xff];
                            acc.gamma[2 * ch] = fma(wa.x, vb, acc.gamma[2 * ch]);
                            acc.gamma[2 * ch + 1] = fma(wa.y, vb, acc.gamma[2 * ch + 1]);
                        }
                    }
                    if (any) {
#pragma unroll
                        for (int q = 0; q < BATCH; ++q)
                            if (c0 + q < NCH && ((m >> (c0 + q)) & 1u)) {
                                acc.gamma[2 * (c0 + q)] = fma(cc[i0], la[q].x, fma(cc[i0 + 1], lb[q].x, acc.gamma[2 * (c0 + q)]));
                                acc.gamma[2 * (c0 + q) + 1] = fma(cc[i0], la[q].y, fma(cc[i0 + 1], lb[q].y, acc.gamma[2 * (c0 + q) + 1]));
                            }
                    }
                }
            }
        }
    }
}

template <int NK, bool DELTA, bool GAMMA, bool CF, bool LDSLC, int WIDE>
__device__ __forceinline__ void add_nodes_any(unsigned long long mask, const int (&k)[NK], const double (&b)[NK],
                                              double omega, const CurveLds& c, const double* __restrict__ lc_lanes,
                                              const unsigned long long* lc_block_mask, double* vbuf, int lane,
                                              Ladders<GAMMA, WIDE>& acc, const double (&cf)[NK], const ConvLds& conv) {
    if constexpr (WIDE > 0) add_nodes_wide<NK, DELTA, GAMMA, CF, WIDE>(mask, k, b, omega, c, lc_lanes, reinterpret_cast<const unsigned*>(lc_block_mask), vbuf, lane, acc, cf);
    else add_nodes<NK, DELTA, GAMMA, CF, LDSLC>(mask, k, b, omega, c, lc_lanes, lc_block_mask, vbuf, lane, acc, cf, conv);
}

// LINEAR_FWD_RATES: the rank-one term a discount factor that is linear in its two knot DFs adds to the Hessian,
// weight * (LJ[ka] - LJ[kb]) (LJ[ka] - LJ[kb])^T with weight = (term value) * (+1 numerator / -1 denominator) * kappa
// (see `Lookup`).  It has no first-order and no convexity part, so it runs as a node without DELTA and with zero
// convexity coefficients.
template <bool GAMMA, bool LDSLC = false, int WIDE = 0>
__device__ __forceinline__ void add_df_correction(bool on, int ka, int kb, double weight, const CurveLds& c,
                                                  const double* __restrict__ lc_lanes,
                                                  const unsigned long long* lc_block_mask, double* vbuf, int lane,
                                                  Ladders<GAMMA, WIDE>& acc) {
    if constexpr (GAMMA) {
        const int k2[2] = {ka, kb};     // untagged is fine: the convexity coefficients are zero
        const double b2[2] = {1.0, -1.0}, none[2] = {0.0, 0.0};
        const double om = on ? weight : 0.0;
        add_nodes_any<2, false, true, true, LDSLC, WIDE>(__ballot(om != 0.0), k2, b2, om, c, lc_lanes, lc_block_mask, vbuf, lane, acc, none, ConvLds{});
    }
}

// LINDF: the curve interpolates with LINEAR_FWD_RATES (a compile-time switch: the state of the corrections would
// otherwise cost the log-linear instantiations registers - measured +9 % on the gamma paths)
// LDSLC: the convexity rows are LDS-resident (`ConvLds`): one 512-thread block per CU instead of two of 256.
// RATIO (wide variants): false for batches without payment lag / weighted coupons - the ratio-node path, whose state sets the
// kernel's register ceiling, is compiled out
template <bool DELTA, bool GAMMA, bool LINDF, bool LDSLC, int WIDE = 0, bool RATIO = true>
// Two waves per SIMD either way (two blocks of 4 waves, the curve tables being about 75 KB, or one block of 8 waves next
// to 112 KB of convexity rows): the register budget is pinned to that (without the bound the gamma instantiation
// drifts to 256 VGPRs + AGPRs and one wave per SIMD).
__global__ __launch_bounds__(general_block_threads(LDSLC, WIDE), (WIDE > 10) ? 1 : 2) void price_general_kernel(CurveDev cv, TradesDev tr, OutputsDev out) {
    static_assert(!(WIDE > 0 && LDSLC), "the wide variants stream their convexity rows from L2");
    constexpr int kBlockThreads = general_block_threads(LDSLC, WIDE);
    constexpr int kWideBands = wide_bands(WIDE);                  // 128-element bands of the P x P output matrix
    constexpr int kWavesPerBlock = kBlockThreads / 64;
    constexpr int kLjPad = WIDE > 0 ? kWidePad : kPillarPad;      // row width of the Jacobian table in LDS
    constexpr int kVbuf = WIDE > 0 ? 2 * kWidePad : 64;           // doubles of hand-off buffer per wave (WIDE: v and omega v)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // LDS carve-up: doubles first (16-byte aligned base), then the int16 index tables
    // (WIDE with GAMMA: the waves' staging areas for the packed ladder come first - 16-byte aligned)
    double* s_wstage = reinterpret_cast<double*>(smem_raw);
    double* s_wvbuf = s_wstage + ((WIDE > 0 && GAMMA) ? kWavesPerBlock * WIDE * kWideChunk : 0);      // WIDE: the v hand-off buffers next
    double* s_x = s_wvbuf + (WIDE > 0 ? kWavesPerBlock * kVbuf : 0);
    double* s_log = s_x + cv.K;
    double* s_invx = s_log + cv.Kc;
    const bool diag = WIDE > 0 || cv.tile_i == cv.tile_j;
    const int n_lj = cv.Kc * kLjPad;                     // one pillar tile of LJ (WIDE: all 64 columns)
    double* s_lj = s_invx + cv.Kc;
    double* s_vbuf = WIDE > 0 ? s_wvbuf : s_lj + static_cast<size_t>(n_lj) * (diag ? 1 : 2);
    double* s_after_lj = s_lj + static_cast<size_t>(n_lj) * ((diag || WIDE > 0) ? 1 : 2);
    // per-knot masks of the structurally non-zero LC blocks: read before every tile, so LDS-resident (a global
    // read here would put a second L2 round trip in front of each tile)
    unsigned long long* s_lcmask = reinterpret_cast<unsigned long long*>(WIDE > 0 ? s_after_lj : s_vbuf + kWavesPerBlock * kVbuf);   // WIDE: 32-bit chunk masks
    // LDS path: packed convexity rows, the short-end records and the per-wave staging of the flat sums
    const int ec_stride = cv.Ec + 1;
    const int n_lcc = (LDSLC && GAMMA) ? (cv.Kcore + 1) * ec_stride : 0;
    double* s_lcc = reinterpret_cast<double*>(s_lcmask + ((GAMMA && !LDSLC) ? (WIDE > 0 ? (cv.Kc + 1) / 2 : cv.Kc) : 0));
    double* s_stage = s_lcc + n_lcc;
    MiniKnot* s_mini = reinterpret_cast<MiniKnot*>(s_stage + ((LDSLC && GAMMA) ? kWavesPerBlock * kConvStage : 0));
    int16_t* s_first = reinterpret_cast<int16_t*>(s_mini + ((LDSLC && GAMMA) ? cv.n_mini : 0));
    int16_t* s_comp = s_first + cv.K;
    int16_t* s_class = s_comp + cv.K;
    int16_t* s_lut = s_class + (LDSLC ? cv.Kc : 0);      // [kLutMax][2] reserved

    for (int i = threadIdx.x; i < cv.K; i += kBlockThreads) {
        s_x[i] = cv.x[i];
        s_first[i] = cv.first_of[i];
        s_comp[i] = cv.compact_of[i];
    }
    for (int i = threadIdx.x; i < cv.Kc; i += kBlockThreads) {
        s_log[i] = cv.log_df[i];
        s_invx[i] = cv.inv_x[i];
        if (GAMMA && !LDSLC && WIDE == 0) s_lcmask[i] = cv.lc_block_mask[i];
        if (LDSLC) s_class[i] = cv.knot_class[i];
    }
    if constexpr (GAMMA && WIDE > 0)
        for (int i = threadIdx.x; i < cv.Kc; i += kBlockThreads) reinterpret_cast<unsigned*>(s_lcmask)[i] = cv.wide_knot_chunks[i];
    for (int i = threadIdx.x; i < n_lcc; i += kBlockThreads) s_lcc[i] = cv.lcc[i];
    if (LDSLC && GAMMA) {
        const double* src = reinterpret_cast<const double*>(cv.mini);
        double* dst = reinterpret_cast<double*>(s_mini);
        for (int i = threadIdx.x; i < cv.n_mini * 8; i += kBlockThreads) dst[i] = src[i];
    }
    for (int i = threadIdx.x; i < 2 * cv.n_lut; i += kBlockThreads) s_lut[i] = cv.lut[i];
    if constexpr (WIDE > 0) {
        for (int i = threadIdx.x; i < n_lj; i += kBlockThreads) s_lj[i] = cv.lj64[i];
    } else {
        const double* row_tile = cv.lj + static_cast<size_t>(cv.tile_i) * n_lj;
        const double* col_tile = cv.lj + static_cast<size_t>(cv.tile_j) * n_lj;
        for (int i = threadIdx.x; i < n_lj; i += kBlockThreads) s_lj[i] = row_tile[i];
        if (!diag)
            for (int i = threadIdx.x; i < n_lj; i += kBlockThreads) s_lj[n_lj + i] = col_tile[i];
    }
    __syncthreads();

    CurveLds c;
    c.x = s_x; c.log_df = s_log; c.inv_x = s_invx; c.lj = s_lj;
    c.first_of = s_first; c.compact_of = s_comp; c.K = cv.K; c.method = cv.method; c.lut = s_lut; c.n_lut = cv.n_lut;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> scalar loads of the header
    const int knot0 = c.compact_of[0];                                    // the value-time knot (t = 0)
    constexpr bool linear_df = LINDF;                                     // LINEAR_FWD_RATES: see `Lookup`
    // ... with LDS-resident convexity rows the corrections of a node's lookups ride in the node's walk (`add_nodes`, NCORR)
    constexpr bool FUSE = LINDF && LDSLC && GAMMA && WIDE == 0;
    double* vbuf = s_vbuf + wave * kVbuf;
    const int P = cv.P;
    c.lj_off = (lane >= 32 && !diag) ? n_lj : 0;
    c.col0 = diag ? 0 : 32;
    c.diag = diag;
    const int row0 = kPillarPad * cv.tile_i, col0g = kPillarPad * cv.tile_j;    // first pillar of the row / column tile
    const bool first_tile = WIDE > 0 || (cv.tile_i == 0 && cv.tile_j == 0);
    const bool one_tile = cv.T == 1;
    const double* __restrict__ lc_lanes = WIDE > 0 ? cv.lcflat : cv.lc_lanes;
#pragma unroll
    for (int s_ = 0; s_ < kWideMaxChunks; ++s_) c.went[s_] = (WIDE > 0 && GAMMA && s_ < WIDE) ? cv.wide_ent[s_ * 64 + lane] : 0u;
    c.wrow = WIDE * kWideChunk;
    c.wpos = WIDE > 0 ? cv.wide_pos[lane] : lane;
    // WIDE: packed entries feeding elements 128 band + 2 lane, + 1 of the row-major P x P matrix (0xffff beyond P * P)
    unsigned wmap[(WIDE > 0 && GAMMA) ? kWideBands : 1];
    if constexpr (WIDE > 0 && GAMMA) {
#pragma unroll
        for (int band = 0; band < kWideBands; ++band) wmap[band] = band * 128 < P * P ? cv.wide_store_map[band * 64 + lane] : 0xffffffffu;
    }
    double* wstage = s_wstage + wave * (WIDE * kWideChunk);
    double* pending_gamma = nullptr;     // WIDE: matrix of the trade whose packed ladder waits in the staging area (wave-uniform)
    auto flush_gamma = [&]() {
        if constexpr (WIDE > 0 && GAMMA) {
            double* g = pending_gamma;
            if (!g) return;
            pending_gamma = nullptr;
            const bool vec = (P & 1) == 0;      // P even: every pair of elements is a 16-byte aligned pair of the matrix
#pragma unroll
            for (int band = 0; band < kWideBands; ++band) {
                if (band * 128 >= P * P) continue;                     // wave-uniform
                const unsigned w = wmap[band];
                const unsigned e0 = w & 0xffffu, e1 = w >> 16;
                const double x0 = e0 != 0xffffu ? wstage[e0] : 0.0;
                const double x1 = e1 != 0xffffu ? wstage[e1] : 0.0;
                double* dst = g + band * 128 + 2 * lane;
                // write-once output: non-temporal, it should not displace the convexity rows in L2
                if (vec) {
                    if (e0 != 0xffffu) nt_store2(reinterpret_cast<double2*>(dst), x0, x1);
                } else {
                    if (e0 != 0xffffu) __builtin_nontemporal_store(x0, dst);
                    if (e1 != 0xffffu) __builtin_nontemporal_store(x1, dst + 1);
                }
            }
        }
    };
    const unsigned long long* lc_block_mask = s_lcmask;
    ConvLds conv;
    conv.lcc = s_lcc; conv.mini = s_mini; conv.ec_stride = ec_stride; conv.flat_of = cv.lcc_pos;
    double* stage = s_stage + wave * kConvStage;
    // LDS path: the knot words handed to add_nodes carry the knot's class in their high half
    auto tag = [&](int k) { return LDSLC ? (k | (static_cast<int>(s_class[k]) << 16)) : k; };
    // ... and the flat convexity sums are scattered to this lane's 4x4 block once per trade: position of each of
    // the block's 16 entries in a packed row (the row's trailing zero for pairs outside the core), two per register
    int conv_pos[(LDSLC && GAMMA) ? kGammaPerLane / 2 : 1];
    if constexpr (LDSLC && GAMMA) {
        const int bi0 = lane >> 3, bj0 = lane & 7;
#pragma unroll
        for (int e = 0; e < kGammaPerLane; e += 2) {
            const int r = 4 * bi0 + (e >> 2), q = 4 * bj0 + (e & 3);
            const int p0 = cv.lcc_pos[r * kPillarPad + q], p1 = cv.lcc_pos[r * kPillarPad + q + 1];
            conv_pos[e / 2] = (p0 < 0 ? cv.Ec : p0) | ((p1 < 0 ? cv.Ec : p1) << 16);
        }
    }

    Ladders<GAMMA, WIDE> total;   // this wave's share of the portfolio aggregate
    total.clear();

    const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
    for (int64_t it = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave; it < tr.n_list; it += wave_stride) {
        const int64_t t = tr.list ? static_cast<int64_t>(tr.list[it]) : it;
        const TradeHeader h = tr.header[t];
        const double N = h.notional, spread = h.spread;
        const double sl = static_cast<double>(h.flt_sign), sf = static_cast<double>(h.fix_sign);
        const int n_flt = h.n_flt, n_fix = h.n_fix;
        const double* f_tp = tr.flt_tp + h.flt_begin;
        const double* f_ts = tr.flt_ts + h.flt_begin;
        const double* f_te = tr.flt_te + h.flt_begin;
        const double* f_al = tr.flt_alpha + h.flt_begin;
        const double* f_w = tr.flt_weight ? tr.flt_weight + h.flt_begin : nullptr;   // wave-uniform
        const double* x_tp = tr.fix_tp + h.fix_begin;
        const double* x_pay = tr.fix_pay + h.fix_begin;

        Ladders<GAMMA, WIDE> acc;
        acc.clear();

        // ---------------------------------------------------------------- float coupons (+ merged fixed)
        for (int base = 0; base < n_flt; base += 64) {
            const int j = base + lane;
            const bool in = j < n_flt;
            double tp = 0.0, ts = 0.0, te = 0.0, al = 0.0;
            if (in) { tp = f_tp[j]; ts = f_ts[j]; te = f_te[j]; al = f_al[j]; }
            const double Nw = (f_w && in) ? N * f_w[j] : N;     // the coupon's own notional
            const bool valid = in && tp >= 0.0;
            const bool accrues = al > 0.0;
            const bool linear = accrues && te == tp;      // D(ts)/D(te)*D(tp) collapses to D(ts)
            const bool ratio = accrues && te != tp;

            // payment node P_j: -N(1 - spread*a) D(tp) (or N*spread*a*D(tp) when nothing accrues)
            double a_pay = valid ? sl * Nw * (spread * al - (accrues ? 1.0 : 0.0)) : 0.0;
            // next coupon's start node lands here when its accrual starts on this payment time
            if (in && j + 1 < n_flt) {
                const double ntp = f_tp[j + 1], nts = f_ts[j + 1], nte = f_te[j + 1], nal = f_al[j + 1];
                if (nal > 0.0 && nte == ntp && ntp >= 0.0 && nts == tp) a_pay += sl * (f_w ? N * f_w[j + 1] : N);
            }
            // the fixed coupon paid at the same time joins the node
            if (in && j < n_fix) {
                const double xtp = x_tp[j];
                if (xtp == tp && xtp > 0.0) a_pay = fma(sf, x_pay[j], a_pay);
            }
            // own start node S_j unless it coincides with the previous payment node
            bool own_start = valid && linear;
            if (own_start && j > 0 && f_tp[j - 1] == ts) own_start = false;
            if (base == 0) flush_gamma();   // WIDE: the previous trade's gamma stores, behind this trade's input loads

            // payment node; its ladder work comes after the ratio nodes, which take over its convexity entries
            int kp[2]; double bp[2], cfp[2]; double omega_p = 0.0;
            const bool pay_on = in && a_pay != 0.0;
            kp[0] = kp[1] = 0; bp[0] = bp[1] = 0.0;
            Lookup qpay;
            qpay.ka = qpay.kb = 0; qpay.ba = qpay.bb = 0.0; qpay.ln_d = 0.0; qpay.kappa = 0.0;
            if (pay_on || (valid && ratio)) qpay = curve_lookup(c, tp);
            if (pay_on) {
                kp[0] = qpay.ka; kp[1] = qpay.kb; bp[0] = qpay.ba; bp[1] = qpay.bb;
                omega_p = a_pay * exp(linear_df ? qpay.ln_d : fma(qpay.ba, c.log_df[qpay.ka], qpay.bb * c.log_df[qpay.kb]));
                acc.pv += omega_p;
            }
            cfp[0] = omega_p * bp[0]; cfp[1] = omega_p * bp[1];
            // a node on the value-time knot alone (t = 0: D = 1, no sensitivity - build_curve_tables checks
            // that) adds to the PV only; the cross-currency assembly pays every weighted coupon there
            const bool pay_node = pay_on && !(kp[0] == knot0 && bp[1] == 0.0);
            {   // unmerged start nodes
                int k[2]; double b[2]; double omega = 0.0;
                k[0] = k[1] = 0; b[0] = b[1] = 0.0;
                double kap = 0.0;
                if (own_start) {
                    const Lookup q = curve_lookup(c, ts);
                    k[0] = q.ka; k[1] = q.kb; b[0] = q.ba; b[1] = q.bb; kap = q.kappa;
                    omega = sl * Nw * exp(linear_df ? q.ln_d : fma(q.ba, c.log_df[q.ka], q.bb * c.log_df[q.kb]));
                    acc.pv += omega;
                }
                { int kt_[2];
#pragma unroll
                  for (int i_ = 0; i_ < 2; ++i_) kt_[i_] = tag(k[i_]);
                  if constexpr (FUSE) {
                      const double kw1[1] = {omega * kap};
                      add_nodes<2, DELTA, GAMMA, false, LDSLC, 1>(__ballot(own_start), kt_, b, omega, c, lc_lanes, lc_block_mask, vbuf, lane, acc, b, conv, kw1, stage);
                  } else
                  add_nodes_any<2, DELTA, GAMMA, false, LDSLC, WIDE>(__ballot(own_start), kt_, b, omega, c, lc_lanes, lc_block_mask, vbuf, lane, acc, b, conv); }
                if constexpr (linear_df && !FUSE) add_df_correction<GAMMA, LDSLC, WIDE>(own_start, k[0], k[1], omega * kap, c, lc_lanes, lc_block_mask, vbuf, lane, acc);
            }
            const bool own_ratio = RATIO && valid && ratio;
            if (RATIO && __ballot(own_ratio)) {   // payment lag: N D(ts) D(tp) / D(te) keeps all three lookups
                int k[6]; double b[6]; double omega = 0.0;
                int kc[6]; double kap[3] = {0.0, 0.0, 0.0};    // LINEAR_FWD: the lookups' own knots and correction weights
#pragma unroll
                for (int i = 0; i < 6; ++i) { k[i] = 0; b[i] = 0.0; kc[i] = 0; }
                if (own_ratio) {
                    const Lookup qs = curve_lookup(c, ts), qe = curve_lookup(c, te), qp = qpay;
                    k[0] = qs.ka; k[1] = qs.kb; b[0] = qs.ba; b[1] = qs.bb;
                    k[2] = qe.ka; k[3] = qe.kb; b[2] = -qe.ba; b[3] = -qe.bb;
                    k[4] = qp.ka; k[5] = qp.kb; b[4] = qp.ba; b[5] = qp.bb;
                    double l = 0.0;
                    if constexpr (linear_df) {
                        l = qs.ln_d - qe.ln_d + qp.ln_d;
                        kap[0] = qs.kappa; kap[1] = -qe.kappa; kap[2] = qp.kappa;
#pragma unroll
                        for (int i = 0; i < 6; ++i) kc[i] = k[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < 6; ++i) l = fma(b[i], c.log_df[k[i]], l);
                    }
                    omega = sl * Nw * exp(l);
                    acc.pv += omega;
                    // te and tp are a few days apart and usually bracketed by the same knots: fold the te weights
                    // into the tp entries (zero-weight entries are skipped downstream)
#pragma unroll
                    for (int i = 2; i < 4; ++i)
#pragma unroll
                        for (int jx = 4; jx < 6; ++jx)
                            if (b[i] != 0.0 && k[i] == k[jx]) { b[jx] += b[i]; b[i] = 0.0; }
                }
                // Convexity per DATE instead of per coupon: coupon j's accrual start is usually coupon j-1's accrual
                // end, bracketed by the same two knots - its start weights (+omega_j b) are handed to the previous
                // lane's end / payment entries on those knots (-omega_{j-1} b'), so that each date's LC tiles are
                // read once.
                double cf[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) cf[i] = own_ratio ? omega * b[i] : 0.0;
                if constexpr (GAMMA) {
                    const bool prev_ratio = lane > 0 && __shfl_up(own_ratio ? 1 : 0, 1, 64) != 0;
                    int pk[4];
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) pk[jx] = __shfl_up(k[2 + jx], 1, 64);
                    double give[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        bool given = false;
                        // payment entries first: with a payment lag inside one knot interval the end weights
                        // have been folded into them, and adding to a live entry costs no extra tile read
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            const int jx = (o + 2) & 3;
                            if (!given && own_ratio && prev_ratio && cf[i] != 0.0 && pk[jx] == k[i]) {
                                give[jx] += cf[i]; cf[i] = 0.0; given = true;
                            }
                        }
                    }
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        const double got = __shfl_down(give[jx], 1, 64);
                        if (lane < 63) cf[2 + jx] += got;
                    }
                }
                // the coupon's payment node sits on the same two knots as the ratio node's payment entries: one
                // set of LC tiles serves both
                if (own_ratio && pay_node) {
                    cf[4] += cfp[0]; cf[5] += cfp[1];
                    cfp[0] = cfp[1] = 0.0;
                }
                // paid on the value-time knot alone (the cross-currency assembly): the payment entries carry no
                // sensitivity, the node is its four accrual entries
                const bool pay_flat = k[4] == knot0 && b[5] == 0.0;
                { int kt_[6];
#pragma unroll
                  for (int i_ = 0; i_ < 6; ++i_) kt_[i_] = tag(k[i_]);
                  if constexpr (FUSE) {
                      const double kw3[3] = {omega * kap[0], omega * kap[1], omega * kap[2]};
                      add_nodes<6, DELTA, GAMMA, true, LDSLC, 3>(__ballot(own_ratio && !pay_flat), kt_, b, omega, c, lc_lanes, lc_block_mask, vbuf, lane, acc, cf, conv, kw3, stage);
                  } else
                  add_nodes_any<6, DELTA, GAMMA, true, LDSLC, WIDE>(__ballot(own_ratio && !pay_flat), kt_, b, omega, c, lc_lanes, lc_block_mask, vbuf, lane, acc, cf, conv); }
                const unsigned long long flat_mask = __ballot(own_ratio && pay_flat);
                if (flat_mask) {
                    int k4[4]; double b4[4], cf4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { k4[i] = k[i]; b4[i] = b[i]; cf4[i] = cf[i]; }
                    { int kt_[4];
#pragma unroll
                      for (int i_ = 0; i_ < 4; ++i_) kt_[i_] = tag(k4[i_]);
                      if constexpr (FUSE) {
                          const double kw2[2] = {omega * kap[0], omega * kap[1]};      // (the payment lookup sits on the value-time knot: no correction)
                          add_nodes<4, DELTA, GAMMA, true, LDSLC, 2>(flat_mask, kt_, b4, omega, c, lc_lanes, lc_block_mask, vbuf, lane, acc, cf4, conv, kw2, stage);
                      } else
                      add_nodes_any<4, DELTA, GAMMA, true, LDSLC, WIDE>(flat_mask, kt_, b4, omega, c, lc_lanes, lc_block_mask, vbuf, lane, acc, cf4, conv); }
                }
                if constexpr (linear_df && !FUSE) {
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        add_df_correction<GAMMA, LDSLC, WIDE>(own_ratio, kc[2 * q], kc[2 * q + 1], omega * kap[q], c, lc_lanes, lc_block_mask, vbuf, lane, acc);
                }
            }
            { int kt_[2];
#pragma unroll
              for (int i_ = 0; i_ < 2; ++i_) kt_[i_] = tag(kp[i_]);
              if constexpr (FUSE) {
                  const double kw1[1] = {omega_p * qpay.kappa};
                  add_nodes<2, DELTA, GAMMA, true, LDSLC, 1>(__ballot(pay_node), kt_, bp, omega_p, c, lc_lanes, lc_block_mask, vbuf, lane, acc, cfp, conv, kw1, stage);
              } else
              add_nodes_any<2, DELTA, GAMMA, true, LDSLC, WIDE>(__ballot(pay_node), kt_, bp, omega_p, c, lc_lanes, lc_block_mask, vbuf, lane, acc, cfp, conv); }
            if constexpr (linear_df && !FUSE) add_df_correction<GAMMA, LDSLC, WIDE>(pay_node, kp[0], kp[1], omega_p * qpay.kappa, c, lc_lanes, lc_block_mask, vbuf, lane, acc);
        }
        // ---------------------------------------------------------------- fixed coupons not merged above
        for (int base = 0; base < n_fix; base += 64) {
            const int j = base + lane;
            bool on = false;
            double tp = 0.0, a = 0.0;
            if (j < n_fix) {
                tp = x_tp[j];
                const bool merged = j < n_flt && f_tp[j] == tp;
                on = !merged && tp > 0.0;
                if (on) a = sf * x_pay[j];
                on = on && a != 0.0;
            }
            int k[2]; double b[2]; double omega = 0.0;
            k[0] = k[1] = 0; b[0] = b[1] = 0.0;
            double kap = 0.0;
            if (on) {
                const Lookup q = curve_lookup(c, tp);
                k[0] = q.ka; k[1] = q.kb; b[0] = q.ba; b[1] = q.bb; kap = q.kappa;
                omega = a * exp(linear_df ? q.ln_d : fma(q.ba, c.log_df[q.ka], q.bb * c.log_df[q.kb]));
                acc.pv += omega;
            }
            { int kt_[2];
#pragma unroll
              for (int i_ = 0; i_ < 2; ++i_) kt_[i_] = tag(k[i_]);
              if constexpr (FUSE) {
                  const double kw1[1] = {omega * kap};
                  add_nodes<2, DELTA, GAMMA, false, LDSLC, 1>(__ballot(on), kt_, b, omega, c, lc_lanes, lc_block_mask, vbuf, lane, acc, b, conv, kw1, stage);
              } else
              add_nodes_any<2, DELTA, GAMMA, false, LDSLC, WIDE>(__ballot(on), kt_, b, omega, c, lc_lanes, lc_block_mask, vbuf, lane, acc, b, conv); }
            if constexpr (linear_df && !FUSE) add_df_correction<GAMMA, LDSLC, WIDE>(on, k[0], k[1], omega * kap, c, lc_lanes, lc_block_mask, vbuf, lane, acc);
        }

        // ---------------------------------------------------------------- results of this trade
        // (more than 32 pillars: this launch holds the tile pair (tile_i, tile_j) of the ladders - the PV comes from the
        // launch of tile (0, 0), the delta tiles from the diagonal launches, off-diagonal gamma tiles are written twice)
        const double pv = wave_sum(acc.pv);
        if constexpr (WIDE > 0) {
            // wide variants: the whole ladder of the trade - pv, lane p = pillar p of the delta ladder, the lanes' 4x4 blocks
            // of the upper triangle and their mirrors
            if (lane == 0) {
                if (out.pv) out.pv[t] = pv;
                total.pv += pv;
            }
            if (DELTA) {
                const double d = acc.delta * 1e-4;
                if (lane < P && out.delta) out.delta[t * P + lane] = d;
                total.delta += d;
            }
            if constexpr (GAMMA) {
                // the packed ladder goes through the wave's staging area to the row-major matrix: 1 KB-contiguous stores
                // (8-byte scattered stores of the entries to their two mirrored positions cost 0.6 ms per 100 k trades at
                // 40 pillars and 3 ms at 64: one write request per element).  The stores themselves are issued at the start
                // of the wave's NEXT trade, behind that trade's input loads (flush_gamma): vector memory operations retire in
                // order, so loads queued behind 13-32 KB of stores would wait for them to drain.
                flush_gamma();                     // (a trade without float coupons has not flushed its predecessor yet)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ch = 0; ch < WIDE; ++ch) {
                    const double g0 = acc.gamma[2 * ch] * 1e-8, g1 = acc.gamma[2 * ch + 1] * 1e-8;
                    total.gamma[2 * ch] += (c.went[ch] & 0x10000u) ? g0 : 0.0;
                    total.gamma[2 * ch + 1] += (c.went[ch] & 0x20000u) ? g1 : 0.0;
                    if (out.gamma) *reinterpret_cast<double2*>(wstage + ch * kWideChunk + 2 * lane) = make_double2(g0, g1);
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                asm volatile("" ::: "memory");
                pending_gamma = out.gamma ? out.gamma + t * static_cast<int64_t>(P) * P : nullptr;
            }
        } else {
        if (lane == 0 && first_tile) {
            if (out.pv) out.pv[t] = pv;
            total.pv += pv;
        }
        if (DELTA && diag) {
            const double d = acc.delta * 1e-4;
            if (lane < kPillarPad && row0 + lane < P && out.delta) out.delta[t * P + row0 + lane] = d;
            total.delta += d;
        }
        if constexpr (GAMMA && LDSLC) {
            // scatter the flat convexity sums to the 4x4 blocks through the wave's staging slot
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int sl = 0; sl < kConvSlices; ++sl) stage[lane + 64 * sl] = acc.conv[sl];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int e = 0; e < kGammaPerLane; e += 2) {
                acc.gamma[e] += stage[conv_pos[e / 2] & 0xffff];
                acc.gamma[e + 1] += stage[conv_pos[e / 2] >> 16];
            }
        }
        if (GAMMA) {
            const int bi = lane >> 3, bj = lane & 7;
            const bool mine_block = !diag || bi <= bj;                       // diagonal tiles: blocks below the diagonal are mirrors
            double* g = out.gamma ? out.gamma + t * static_cast<int64_t>(P) * P : nullptr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = row0 + 4 * bi + i;
                double gv[4];
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    gv[jx] = acc.gamma[i * 4 + jx] * 1e-8;
                    total.gamma[i * 4 + jx] += gv[jx];
                }
                if (g && r < P && mine_block) {
                    if (one_tile && P == kPillarPad) {
                        double2* dst = reinterpret_cast<double2*>(g + r * kPillarPad + 4 * bj);
                        nt_store2(dst, gv[0], gv[1]);        // write-once output: it should not displace the tables in L2
                        nt_store2(dst + 1, gv[2], gv[3]);
                    } else {
#pragma unroll
                        for (int jx = 0; jx < 4; ++jx)
                            if (col0g + 4 * bj + jx < P) g[r * P + col0g + 4 * bj + jx] = gv[jx];
                    }
                }
            }
            if (g && (!diag || bi < bj)) {   // the mirror block: rows (col0g + 4 bj + jx) hold column jx of this block
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const int r = col0g + 4 * bj + jx;
                    if (r >= P) continue;
                    double gt[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) gt[i] = acc.gamma[i * 4 + jx] * 1e-8;
                    if (one_tile && P == kPillarPad) {
                        double2* dst = reinterpret_cast<double2*>(g + r * kPillarPad + 4 * bi);
                        nt_store2(dst, gt[0], gt[1]);
                        nt_store2(dst + 1, gt[2], gt[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (row0 + 4 * bi + i < P) g[r * P + row0 + 4 * bi + i] = gt[i];
                    }
                }
            }
        }
    }   // WIDE == 0
    }

    flush_gamma();
    // ------------------------------------------------------------------------ block partial of the aggregate
    if constexpr (WIDE > 0) {
        if (out.block_partials) {
            // one record [pv, delta[64], packed gamma entries] per block, in the kernel's own layouts: every wave leaves its
            // totals in its hand-off buffer (pv, delta) and its staging area (gamma), thread e adds entry e over the waves
            // in a fixed order and the sums go out as contiguous stores; the packed entries are mapped to the matrix by
            // the final reduction (reduce_wide_kernel)
            flush_gamma();
            __syncthreads();
            vbuf[lane] = DELTA ? total.delta : 0.0;
            if (lane == 0) vbuf[kWidePad] = total.pv;
            if constexpr (GAMMA) {
#pragma unroll
                for (int ch = 0; ch < WIDE; ++ch)
                    *reinterpret_cast<double2*>(wstage + ch * kWideChunk + 2 * lane) = make_double2(total.gamma[2 * ch], total.gamma[2 * ch + 1]);
            }
            __syncthreads();
            constexpr int kRecord = 1 + kWidePad + (GAMMA ? WIDE * kWideChunk : 0);
            double* dst = out.block_partials + static_cast<size_t>(blockIdx.x) * (1 + kWidePad + cv.wide_nch * kWideChunk);
            for (int i = threadIdx.x; i < kRecord; i += kBlockThreads) {
                double sum = 0.0;
#pragma unroll
                for (int w = 0; w < kWavesPerBlock; ++w)
                    sum += i == 0 ? s_vbuf[w * kVbuf + kWidePad] : (i <= kWidePad ? s_vbuf[w * kVbuf + i - 1]
                                                                                 : s_wstage[w * (WIDE * kWideChunk) + i - 1 - kWidePad]);
                dst[i] = sum;
            }
        }
    } else
    if (out.block_partials) {
        __syncthreads();   // every wave is done with the curve tables; reuse the LDS for the reduction
        double* red = reinterpret_cast<double*>(smem_raw);   // [waves][kAggStride]
        double* mine = red + wave * kAggStride;
        if (lane == 0) mine[0] = total.pv;
        if (lane < kPillarPad) mine[1 + lane] = DELTA ? total.delta : 0.0;
        {
            const int bi = lane >> 3, bj = lane & 7;
#pragma unroll
            for (int e = 0; e < kGammaPerLane; ++e) {
                const int r = 4 * bi + (e >> 2), q = 4 * bj + (e & 3);
                const double x = GAMMA ? total.gamma[GAMMA ? e : 0] : 0.0;
                if (!diag || bi <= bj) mine[1 + kPillarPad + r * kPillarPad + q] = x;
                if (diag && bi < bj) mine[1 + kPillarPad + q * kPillarPad + r] = x;     // lower blocks of a diagonal tile: mirrors
            }
        }
        __syncthreads();
        double* dst = out.block_partials + static_cast<size_t>(blockIdx.x) * kAggStride;
        for (int i = threadIdx.x; i < kAggStride; i += kBlockThreads) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) s += red[w * kAggStride + i];
            dst[i] = s;
        }
    }
}

}  // namespace

size_t general_kernel_lds_bytes(int K, int Kc, bool two_tiles) {
    constexpr int kWavesPerBlock = kThreadsL2 / 64;
    size_t tables = sizeof(double) * (static_cast<size_t>(K) + 3 * Kc + static_cast<size_t>(Kc) * kPillarPad * (two_tiles ? 2 : 1) +
                                      kWavesPerBlock * 64) + sizeof(int16_t) * (2 * static_cast<size_t>(K) + 2 * kLutMax);
    size_t reduce = sizeof(double) * kWavesPerBlock * kAggStride;
    size_t need = tables > reduce ? tables : reduce;
    return (need + 15) & ~static_cast<size_t>(15);
}

// LDS of the variant with resident convexity rows for a curve of these sizes
size_t general_lds_kernel_lds_bytes_for(int K, int Kc, int Kcore, int Ec, int n_mini, int n_lut, bool gamma) {
    constexpr int kWavesPerBlock = kThreadsLds / 64;
    size_t bytes = sizeof(double) * (static_cast<size_t>(K) + 2 * Kc + static_cast<size_t>(Kc) * kPillarPad + kWavesPerBlock * 64);
    if (gamma)
        bytes += sizeof(double) * (static_cast<size_t>(Kcore + 1) * (Ec + 1) + kWavesPerBlock * kConvStage) +
                 sizeof(MiniKnot) * n_mini;
    // (the bucket table is the last array of the carve-up: its actual size counts, not the kLutMax reserve - with the
    // reserve the README-sized curves are 214 bytes over the 160 KB)
    bytes += sizeof(int16_t) * (2 * static_cast<size_t>(K) + Kc + 2 * static_cast<size_t>(n_lut));
    const size_t reduce = sizeof(double) * kWavesPerBlock * kAggStride;
    if (bytes < reduce) bytes = reduce;
    return (bytes + 15) & ~static_cast<size_t>(15);
}

// ... of an uploaded curve, or 0 when it has no packed tables
size_t general_lds_kernel_lds_bytes(const CurveDev& cv, bool gamma) {
    if (!cv.lcc_pos || !cv.knot_class || (gamma && !cv.lcc) || cv.T > 1) return 0;
    return general_lds_kernel_lds_bytes_for(cv.K, cv.Kc, cv.Kcore, cv.Ec, cv.n_mini, cv.n_lut, gamma);
}

// worth it only for GAMMA (the rows are the convexity term); needs the packed tables and at most 192 flat entries
bool general_lds_rows_fit(size_t lds_bytes, int Ec, int n_fringe) {
    return lds_bytes > 0 && lds_bytes <= 160 * 1024 && Ec + 1 + n_fringe <= kConvStage;
}

bool general_kernel_uses_lds_rows(const CurveDev& cv, bool gamma) {
    return gamma && general_lds_rows_fit(general_lds_kernel_lds_bytes(cv, gamma), cv.Ec, cv.n_fringe);
}

int general_kernel_threads(const CurveDev& cv, bool gamma) {
    return general_kernel_uses_lds_rows(cv, gamma) ? kThreadsLds : kThreadsL2;
}

hipError_t launch_price_general(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, bool want_delta,
                                bool want_gamma, int n_blocks, hipStream_t stream) {
    const bool lin = cv.method == 2;
    if (general_kernel_uses_lds_rows(cv, want_gamma)) {
        const size_t lds = general_lds_kernel_lds_bytes(cv, true);
        dim3 grid(n_blocks), block(kThreadsLds);
        if (lin) hipLaunchKernelGGL((price_general_kernel<true, true, true, true>), grid, block, lds, stream, cv, tr, out);
        else hipLaunchKernelGGL((price_general_kernel<true, true, false, true>), grid, block, lds, stream, cv, tr, out);
        return hipGetLastError();
    }
    const size_t lds = general_kernel_lds_bytes(cv.K, cv.Kc, cv.tile_i != cv.tile_j);
    dim3 grid(n_blocks), block(kThreadsL2);
    if (want_gamma) {
        if (lin) hipLaunchKernelGGL((price_general_kernel<true, true, true, false>), grid, block, lds, stream, cv, tr, out);
        else hipLaunchKernelGGL((price_general_kernel<true, true, false, false>), grid, block, lds, stream, cv, tr, out);
    } else if (want_delta) {
        if (lin) hipLaunchKernelGGL((price_general_kernel<true, false, true, false>), grid, block, lds, stream, cv, tr, out);
        else hipLaunchKernelGGL((price_general_kernel<true, false, false, false>), grid, block, lds, stream, cv, tr, out);
    } else {
        if (lin) hipLaunchKernelGGL((price_general_kernel<false, false, true, false>), grid, block, lds, stream, cv, tr, out);
        else hipLaunchKernelGGL((price_general_kernel<false, false, false, false>), grid, block, lds, stream, cv, tr, out);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ wide variants
namespace {

// Fixed-order sum of the wide kernel's block partials ([n_blocks][1 + 64 + nch * 128]: pv, delta by pillar, gamma on the
// packed triangle) -> agg[1 + P + P*P]: one wavefront per output, lanes stride over the blocks, then a fixed butterfly;
// element (r, c) of the matrix reads its packed entry through the curve's store map.
__global__ __launch_bounds__(256) void reduce_wide_kernel(const double* partials, int n_blocks, int P, int stride, int has_delta,
                                                           int has_gamma, const uint32_t* store_map, double* agg) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);       // index into agg
    if (o >= 1 + P + P * P) return;
    int i = 0;                                                               // index into a block's record
    bool live = true;
    if (o >= 1 + P) {
        const int f = o - 1 - P;                                             // flat index r * P + c
        const uint32_t w = store_map[(f >> 7) * 64 + ((f & 127) >> 1)];
        i = 1 + kWidePad + static_cast<int>((f & 1) ? (w >> 16) : (w & 0xffffu));
        live = has_gamma != 0;
    } else if (o >= 1) { i = o; live = has_delta != 0; }
    double s = 0.0;
    if (live)
        for (int b = lane; b < n_blocks; b += 64) s += partials[static_cast<size_t>(b) * stride + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) agg[o] = s;
}

template <bool LINDF>
void collect_wide(std::vector<const void*>& fns) {
    if (!LINDF) {
        fns.push_back(reinterpret_cast<const void*>(&price_general_kernel<true, true, false, false, 7, false>));
        fns.push_back(reinterpret_cast<const void*>(&price_general_kernel<true, true, false, false, 10, false>));
        fns.push_back(reinterpret_cast<const void*>(&price_general_kernel<true, true, false, false, 17, false>));
    }
    fns.push_back(reinterpret_cast<const void*>(&price_general_kernel<true, true, LINDF, false, 7>));
    fns.push_back(reinterpret_cast<const void*>(&price_general_kernel<true, true, LINDF, false, 10>));
    fns.push_back(reinterpret_cast<const void*>(&price_general_kernel<true, true, LINDF, false, 17>));
    fns.push_back(reinterpret_cast<const void*>(&price_general_kernel<true, false, LINDF, false, 7>));
    fns.push_back(reinterpret_cast<const void*>(&price_general_kernel<false, false, LINDF, false, 7>));
}

}  // namespace

size_t wide_kernel_lds_bytes(int K, int Kc, int nch, bool gamma) {
    const int kWavesPerBlock = general_block_threads(false, gamma ? nch : 7) / 64;
    size_t bytes = sizeof(double) * (static_cast<size_t>(K) + 2 * Kc + static_cast<size_t>(Kc) * kWidePad + kWavesPerBlock * 2 * kWidePad +
                                     (gamma ? static_cast<size_t>(kWavesPerBlock) * nch * kWideChunk : 0)) +
                   (gamma ? sizeof(unsigned long long) * static_cast<size_t>((Kc + 1) / 2) : 0) +
                   sizeof(int16_t) * (2 * static_cast<size_t>(K) + 2 * kLutMax);
    return (bytes + 15) & ~static_cast<size_t>(15);
}

int wide_partial_doubles(int nch) { return 1 + kWidePad + nch * kWideChunk; }

int wide_kernel_threads(int nch, bool gamma) { return general_block_threads(false, gamma ? nch : 7); }

// blocks a CU holds: the register budget allows two waves per SIMD (launch bounds), the LDS image the rest
int wide_kernel_blocks_per_cu(size_t lds_bytes, int threads) {
    const size_t by_lds = lds_bytes ? (160 * 1024) / lds_bytes : 1, by_regs = static_cast<size_t>(512 / threads);
    const size_t n = by_lds < by_regs ? by_lds : by_regs;
    return static_cast<int>(n < 1 ? 1 : n);
}

hipError_t launch_price_wide(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, bool want_delta,
                             bool want_gamma, int n_blocks, hipStream_t stream) {
    const bool lin = cv.method == 2;
    const size_t lds = wide_kernel_lds_bytes(cv.K, cv.Kc, cv.wide_nch, want_gamma);
    dim3 grid(n_blocks), block(wide_kernel_threads(cv.wide_nch, want_gamma));
    using Fn = void (*)(CurveDev, TradesDev, OutputsDev);
    Fn fn = nullptr;
    if (want_gamma && !lin && !tr.any_ratio) {
        switch (cv.wide_nch) {
            case 7: fn = &price_general_kernel<true, true, false, false, 7, false>; break;
            case 10: fn = &price_general_kernel<true, true, false, false, 10, false>; break;
            case 17: fn = &price_general_kernel<true, true, false, false, 17, false>; break;
            default: return hipErrorInvalidValue;
        }
    } else if (want_gamma) {
        switch (cv.wide_nch) {
            case 7: fn = lin ? &price_general_kernel<true, true, true, false, 7> : &price_general_kernel<true, true, false, false, 7>; break;
            case 10: fn = lin ? &price_general_kernel<true, true, true, false, 10> : &price_general_kernel<true, true, false, false, 10>; break;
            case 17: fn = lin ? &price_general_kernel<true, true, true, false, 17> : &price_general_kernel<true, true, false, false, 17>; break;
            default: return hipErrorInvalidValue;
        }
    } else if (want_delta) {
        fn = lin ? &price_general_kernel<true, false, true, false, 7> : &price_general_kernel<true, false, false, false, 7>;
    } else {
        fn = lin ? &price_general_kernel<false, false, true, false, 7> : &price_general_kernel<false, false, false, false, 7>;
    }
    hipLaunchKernelGGL(fn, grid, block, lds, stream, cv, tr, out);
    return hipGetLastError();
}

hipError_t launch_reduce_wide(const CurveDev& cv, const double* partials, int n_blocks, bool has_delta, bool has_gamma,
                              double* agg, hipStream_t stream) {
    const int P = cv.P, n_out = 1 + P + P * P;
    hipLaunchKernelGGL(reduce_wide_kernel, dim3((n_out + 3) / 4), dim3(256), 0, stream, partials, n_blocks, P,
                       wide_partial_doubles(cv.wide_nch), has_delta ? 1 : 0, has_gamma ? 1 : 0, cv.wide_store_map, agg);
    return hipGetLastError();
}

hipError_t set_general_kernel_lds_limit(size_t bytes) {
    std::vector<const void*> wide;
    collect_wide<false>(wide);
    collect_wide<true>(wide);
    for (const void* f : wide) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        if (e != hipSuccess) return e;
    }
    const void* fns[] = {reinterpret_cast<const void*>(&price_general_kernel<true, true, false, false>),
                         reinterpret_cast<const void*>(&price_general_kernel<true, false, false, false>),
                         reinterpret_cast<const void*>(&price_general_kernel<false, false, false, false>),
                         reinterpret_cast<const void*>(&price_general_kernel<true, true, true, false>),
                         reinterpret_cast<const void*>(&price_general_kernel<true, false, true, false>),
                         reinterpret_cast<const void*>(&price_general_kernel<false, false, true, false>),
                         reinterpret_cast<const void*>(&price_general_kernel<true, true, false, true>),
                         reinterpret_cast<const void*>(&price_general_kernel<true, true, true, true>)};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace adr
