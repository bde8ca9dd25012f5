// CDNA4 (gfx950) FAST kernel: PV, pillar delta ladder and pillar x pillar gamma of OIS trades.
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
// are folded into "nodes" (time, coefficient): with te == tp a float coupon is
// N*(D(ts) - (1 - spread*a) D(tp)), its start node coincides with the previous coupon's payment node, and
// the fixed coupon on the same date joins the same node: a standard OIS with M coupons is M + 1 nodes.
// Trades with te != tp (ratio terms) are not handled here; the host routes them to kernels_general.hip.
//
// Layout.  Everything a node touches lives in LDS: the knot search arrays, LJ restricted to the curve's
// core pillars, LC on the packed upper triangle of core x core, and 64-byte records for the short-end
// knots that depend on at most two par rates (curve_tables.cpp, build_packed_layout).  One 64-lane
// wavefront prices one trade at a time: lanes = cash flows while nodes are built (coalesced loads, binary
// search, exp), then lanes = packed gamma entries (EPL per lane) and lanes = pillars for delta while the
// nodes are consumed, their few scalars broadcast with v_readlane.  The packed ladder is expanded to the
// full 32x32 matrix through a per-wave LDS slot when the trade is written (full 512-byte row stores).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kernels.hpp"

namespace adr {

namespace {

constexpr int kBlockThreads = kFastThreads;
constexpr int kWavesPerBlock = kBlockThreads / 64;

__device__ __forceinline__ int readlane_i(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }

__device__ __forceinline__ double readlane_d(double x, int lane) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// Same-wave LDS hand-off: LDS operations of one wave execute in order; the fences only stop the
// compiler from moving accesses across the hand-off point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct CurveLds {
    const double* x;            // [K]
    const double* log_df;       // [Kc]
    const double* inv_x;        // [Kc]
    const double* ljc;          // [Kcore][pc_pad]
    const double* lcc;          // [Kcore][Ec + 1]
    const MiniKnot* mini;       // [n_mini]
    const int16_t* first_of;    // [K]
    const int16_t* compact_of;  // [K]
    const int16_t* knot_class;  // [Kc]
    int K, method, pc_pad, ec_stride;
    int core_slots;             // 64-entry slots that hold core pairs: ceil(Ec / 64)
};

struct Lookup {
    int ka, kb;        // compact knots
    double ba, bb;     // D = exp(ba*L[ka] + bb*L[kb]); bb == 0: single knot
};

// InterpolatorAd.simple_interpolate for one time (interpolator_ad.py:210-243) in weight form.
__device__ __forceinline__ Lookup curve_lookup(const CurveLds& c, double t) {
    const int K = c.K;
    int lo = 0, hi = K;                 // j = first knot with x > t
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
    const double w = (fabs(dx) <= 0x1p-104) ? 0.0 : (tau - xa) / dx;   // jnp.interp: fp[i-1] when dx ~ 0
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

template <bool GAMMA, int EPL>
struct Ladders {
    double pv;
    double delta;                      // lane p (and p + 32, duplicated) holds pillar p
    double gamma[GAMMA ? EPL : 1];     // packed entry lane + 64 s
    __device__ __forceinline__ void clear() {
        pv = 0.0; delta = 0.0;
#pragma unroll
        for (int s = 0; s < (GAMMA ? EPL : 1); ++s) gamma[s] = 0.0;
    }
};

// Per-lane constants of the packed layout.
template <int EPL>
struct PackedLane {
    int pil;                 // pillar this lane builds v for (lanes 32..63 duplicate 0..31)
    int col;                 // its column in ljc (the zero column outside the core)
    int up[EPL];             // p and q of packed entry lane + 64 s (entry 0's when there is none)
    int vq[EPL];
};

// v_p = d lnD / d r_p share of one knot for this lane's pillar.
__device__ __forceinline__ double knot_v(int cls, double b, const CurveLds& c, int col, int pil, double v) {
    if (cls >= 0) return fma(b, c.ljc[cls * c.pc_pad + col], v);
    if (cls <= -3) {
        const MiniKnot& m = c.mini[-3 - cls];
        return fma(b, pil == m.p[0] ? m.lj[0] : (pil == m.p[1] ? m.lj[1] : 0.0), v);
    }
    return v;
}

// Second-derivative share of one knot: coef * LC[knot] added to the packed entries.
// A core knot's row is contiguous: entry lane + 64 s sits at row[lane + 64 s] for the slots that hold core
// pairs.  The last of those slots is only partly used by core pairs; its other lanes pick up whatever
// follows the row (finite numbers: the next row or the zero slack) into entries nothing ever reads.
template <int EPL>
__device__ __forceinline__ void knot_lc(int cls, double coef, const CurveLds& c, int lane, double (&gamma)[EPL]) {
    if (cls >= 0) {
        const double* row = c.lcc + cls * c.ec_stride + lane;
#pragma unroll
        for (int s = 0; s < EPL; ++s)
            if (s < c.core_slots) gamma[s] = fma(coef, row[64 * s], gamma[s]);
    } else if (cls <= -3) {
        const MiniKnot& m = c.mini[-3 - cls];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = __builtin_amdgcn_readfirstlane(m.e[i]);      // wave-uniform entry, -1 when unused
            const double val = coef * m.lc[i];
#pragma unroll
            for (int s = 0; s < EPL; ++s)
                if ((e >> 6) == s && lane == (e & 63)) gamma[s] += val;
        }
    }
}

template <bool DELTA, bool GAMMA, int EPL>
__global__ __launch_bounds__(kBlockThreads) void price_fast_kernel(CurveDev cv, TradesDev tr, OutputsDev out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // LDS carve-up: 64-byte records, doubles, then the int16 tables
    const int ec_stride = cv.Ec + 1;
    const int n_ljc = (cv.Kcore + 1) * cv.pc_pad;          // + the all-zero row
    const int n_lcc = GAMMA ? (cv.Kcore + 1) * ec_stride : 0;
    const int n_slack = GAMMA ? 64 * EPL : 0;            // zeros behind the last row (rows are read 64 wide)
    const int stage_stride = GAMMA ? max((cv.Eu + 1) & ~1, 2 * kPillarPad) : 0;
    MiniKnot* s_mini = reinterpret_cast<MiniKnot*>(smem_raw);
    double* s_x = reinterpret_cast<double*>(s_mini + cv.n_mini);
    double* s_log = s_x + cv.K;
    double* s_invx = s_log + cv.Kc;
    double* s_ljc = s_invx + cv.Kc;
    double* s_lcc = s_ljc + n_ljc;
    double* s_stage = s_lcc + n_lcc + n_slack;                      // per wave: packed ladder at output time; its first
                                                          // 64 doubles double as the node hand-off buffers u, v
    int16_t* s_first = reinterpret_cast<int16_t*>(s_stage + kWavesPerBlock * stage_stride);
    int16_t* s_comp = s_first + cv.K;
    int16_t* s_class = s_comp + cv.K;
    int16_t* s_omap = s_class + cv.Kc;                    // [32*32] packed entry of gamma[r][c], -1 if none

    {
        const double* src = reinterpret_cast<const double*>(cv.mini);
        double* dst = reinterpret_cast<double*>(s_mini);
        for (int i = threadIdx.x; i < cv.n_mini * 8; i += kBlockThreads) dst[i] = src[i];
    }
    for (int i = threadIdx.x; i < cv.K; i += kBlockThreads) {
        s_x[i] = cv.x[i];
        s_first[i] = cv.first_of[i];
        s_comp[i] = cv.compact_of[i];
    }
    for (int i = threadIdx.x; i < cv.Kc; i += kBlockThreads) {
        s_log[i] = cv.log_df[i];
        s_invx[i] = cv.inv_x[i];
        s_class[i] = cv.knot_class[i];
    }
    if (GAMMA)
        for (int i = threadIdx.x; i < kPillarPad * kPillarPad; i += kBlockThreads) s_omap[i] = cv.out_map[i];
    for (int i = threadIdx.x; i < n_ljc; i += kBlockThreads) s_ljc[i] = cv.ljc[i];
    for (int i = threadIdx.x; i < n_lcc; i += kBlockThreads) s_lcc[i] = cv.lcc[i];
    for (int i = threadIdx.x; i < n_slack; i += kBlockThreads) s_lcc[n_lcc + i] = 0.0;
    __syncthreads();

    CurveLds c;
    c.x = s_x; c.log_df = s_log; c.inv_x = s_invx; c.ljc = s_ljc; c.lcc = s_lcc; c.mini = s_mini;
    c.first_of = s_first; c.compact_of = s_comp; c.knot_class = s_class;
    c.K = cv.K; c.method = cv.method; c.pc_pad = cv.pc_pad; c.ec_stride = ec_stride;
    c.core_slots = (cv.Ec + 63) >> 6;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> scalar header loads
    double* stage = s_stage + wave * stage_stride;
    double* ubuf = stage;
    double* vbuf = ubuf + kPillarPad;
    const int P = cv.P;
    const int bi = lane >> 3, bj = lane & 7;
    const int zero_row = cv.Kcore;

    PackedLane<EPL> pl;
    pl.pil = lane & 31;
    pl.col = cv.pillar_to_core[pl.pil];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int e = lane + 64 * s;
        const bool on = GAMMA && e < cv.Eu;
        pl.up[s] = on ? cv.ent_pq[2 * e] : 0;
        pl.vq[s] = on ? cv.ent_pq[2 * e + 1] : 0;
    }
    // this lane's 4x4 block of the output matrix starts at row 4*bi, column 4*bj
    const int16_t* omap = s_omap + (4 * bi) * kPillarPad + 4 * bj;

    Ladders<GAMMA, EPL> total;   // this wave's share of the portfolio aggregate
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
        const double* f_al = tr.flt_alpha + h.flt_begin;
        const double* x_tp = tr.fix_tp + h.fix_begin;
        const double* x_pay = tr.fix_pay + h.fix_begin;

        Ladders<GAMMA, EPL> acc;
        acc.clear();

        // Two kinds of lookup passes: float chunks (payment nodes + start nodes in the spare lanes) and
        // fixed chunks (only the fixed coupons that did not merge into a float payment node).
        const int n_flt_chunks = (n_flt + 63) >> 6, n_fix_chunks = (n_fix + 63) >> 6;
        for (int chunk = 0; chunk < n_flt_chunks + n_fix_chunks; ++chunk) {
            double qt = 0.0, qa = 0.0;      // this lane's query: time and coefficient
            bool qon = false;
            unsigned long long leftover = 0;
            double ts = 0.0;
            if (chunk < n_flt_chunks) {
                const int base = chunk << 6;
                const int j = base + lane;
                const bool in = j < n_flt;
                double tp = 0.0, al = 0.0;
                if (in) { tp = f_tp[j]; ts = f_ts[j]; al = f_al[j]; }
                const bool valid = in && tp >= 0.0;
                const bool accrues = al > 0.0;        // te == tp for every coupon of a fast-path trade
                // payment node P_j: -N(1 - spread*a) D(tp)   (N*spread*a*D(tp) when nothing accrues)
                double a_pay = valid ? sl * N * (spread * al - (accrues ? 1.0 : 0.0)) : 0.0;
                // the next coupon's start node lands here when its accrual starts on this payment time
                if (in && j + 1 < n_flt) {
                    const double ntp = f_tp[j + 1], nts = f_ts[j + 1], nal = f_al[j + 1];
                    if (nal > 0.0 && ntp >= 0.0 && nts == tp) a_pay += sl * N;
                }
                // the fixed coupon paid at the same time joins the node
                if (in && j < n_fix) {
                    const double xtp = x_tp[j];
                    if (xtp == tp && xtp > 0.0) a_pay = fma(sf, x_pay[j], a_pay);
                }
                // own start node S_j unless it coincides with the previous payment node
                bool own_start = valid && accrues;
                if (own_start && j > 0 && f_tp[j - 1] == ts) own_start = false;

                qt = tp; qa = a_pay; qon = in && a_pay != 0.0;
                leftover = __ballot(own_start);
                int dst = min(n_flt - base, 64);      // first spare lane
                while (leftover && dst < 64) {
                    const int src = __builtin_ctzll(leftover);
                    leftover &= leftover - 1;
                    const double st = readlane_d(ts, src);
                    if (lane == dst) { qt = st; qa = sl * N; qon = true; }
                    ++dst;
                }
            } else {
                const int j = ((chunk - n_flt_chunks) << 6) + lane;
                if (j < n_fix) {
                    qt = x_tp[j];
                    const bool merged = j < n_flt && f_tp[j] == qt;
                    qa = sf * x_pay[j];
                    qon = !merged && qt > 0.0 && qa != 0.0;
                }
            }

            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 1) {        // start nodes that found no spare lane (a full 64-coupon chunk)
                    if (!leftover) break;
                    qt = ts; qa = sl * N; qon = ((leftover >> lane) & 1ull) != 0;
                }
                const unsigned long long mask0 = __ballot(qon);
                if (!mask0) continue;
                // ---- build: lookup + exp in the lanes that own a query
                int cls_a = -2, cls_b = -2;
                double ba = 0.0, bb = 0.0, omega = 0.0;
                if (qon) {
                    const Lookup q = curve_lookup(c, qt);
                    ba = q.ba; bb = q.bb;
                    cls_a = c.knot_class[q.ka];
                    cls_b = bb != 0.0 ? c.knot_class[q.kb] : -2;
                    omega = qa * exp(fma(ba, c.log_df[q.ka], bb * c.log_df[q.kb]));
                    acc.pv += omega;
                }
                if (!DELTA) continue;
                // Nodes whose knots are core rows (an all-zero row standing in for a knot nothing depends
                // on) take the branch-free loop; nodes touching a short-end knot the general one.
                const bool greeks = qon && !(cls_a == -2 && cls_b == -2);
                const bool has_mini = cls_a <= -3 || cls_b <= -3;
                const int row_a = cls_a >= 0 ? cls_a : zero_row, row_b = cls_b >= 0 ? cls_b : zero_row;
                unsigned long long mask = __ballot(greeks && !has_mini);
                while (mask) {
                    const int n = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int ra = readlane_i(row_a, n), rb = readlane_i(row_b, n);
                    const double om = readlane_d(omega, n);
                    const double wa = readlane_d(ba, n), wb = readlane_d(bb, n);
                    const double v = fma(wb, c.ljc[rb * c.pc_pad + pl.col], wa * c.ljc[ra * c.pc_pad + pl.col]);
                    acc.delta = fma(om, v, acc.delta);
                    if (GAMMA) {
                        __builtin_amdgcn_wave_barrier();
                        if (lane < 32) { ubuf[lane] = om * v; vbuf[lane] = v; }
                        wave_lds_sync();
                        const double* rowa = c.lcc + ra * c.ec_stride + lane;
                        const double* rowb = c.lcc + rb * c.ec_stride + lane;
                        const double coa = om * wa, cob = om * wb;
#pragma unroll
                        for (int s = 0; s < EPL; ++s) {
                            double g = fma(ubuf[pl.up[s]], vbuf[pl.vq[s]], acc.gamma[s]);
                            if (s < c.core_slots) g = fma(cob, rowb[64 * s], fma(coa, rowa[64 * s], g));
                            acc.gamma[s] = g;
                        }
                    }
                }
                mask = __ballot(greeks && has_mini);
                while (mask) {
                    const int n = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const int ca = readlane_i(cls_a, n), cb = readlane_i(cls_b, n);
                    const double om = readlane_d(omega, n);
                    const double wa = readlane_d(ba, n), wb = readlane_d(bb, n);
                    const double v = knot_v(cb, wb, c, pl.col, pl.pil, knot_v(ca, wa, c, pl.col, pl.pil, 0.0));
                    acc.delta = fma(om, v, acc.delta);
                    if (GAMMA) {
                        __builtin_amdgcn_wave_barrier();
                        if (lane < 32) { ubuf[lane] = om * v; vbuf[lane] = v; }
                        wave_lds_sync();
#pragma unroll
                        for (int s = 0; s < EPL; ++s) acc.gamma[s] = fma(ubuf[pl.up[s]], vbuf[pl.vq[s]], acc.gamma[s]);
                        knot_lc<EPL>(ca, om * wa, c, lane, acc.gamma);
                        knot_lc<EPL>(cb, om * wb, c, lane, acc.gamma);
                    }
                }
            }
        }

        // ---------------------------------------------------------------- results of this trade
        const double pv = wave_sum(acc.pv);
        if (lane == 0) {
            if (out.pv) out.pv[t] = pv;
            total.pv += pv;
        }
        if (DELTA) {
            if (lane < P && out.delta) out.delta[t * P + lane] = acc.delta * 1e-4;
            total.delta += acc.delta;
        }
        if (GAMMA) {
#pragma unroll
            for (int s = 0; s < EPL; ++s) total.gamma[s] += acc.gamma[s];
            if (out.gamma) {
                // expand the packed entries to the symmetric 32x32 matrix through the wave's LDS slot
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int s = 0; s < EPL; ++s)
                    if (lane + 64 * s < cv.Eu) stage[lane + 64 * s] = acc.gamma[s];
                wave_lds_sync();
                double* g = out.gamma + t * static_cast<int64_t>(P) * P;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * bi + i;
                    double gv[4];
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        const int m = omap[i * kPillarPad + jx];
                        gv[jx] = m >= 0 ? stage[m] * 1e-8 : 0.0;
                    }
                    if (r < P) {
                        if (P == kPillarPad) {
                            double2* dst = reinterpret_cast<double2*>(g + r * kPillarPad + 4 * bj);
                            dst[0] = make_double2(gv[0], gv[1]);
                            dst[1] = make_double2(gv[2], gv[3]);
                        } else {
#pragma unroll
                            for (int jx = 0; jx < 4; ++jx)
                                if (4 * bj + jx < P) g[r * P + 4 * bj + jx] = gv[jx];
                        }
                    }
                }
            }
        }
    }

    // ------------------------------------------------------------------------ block partial of the aggregate
    if (out.block_partials) {
        double tot_gamma[GAMMA ? kGammaPerLane : 1];
        if (GAMMA) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < EPL; ++s)
                if (lane + 64 * s < cv.Eu) stage[lane + 64 * s] = total.gamma[s];
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < kGammaPerLane; ++e) {
                const int m = omap[(e >> 2) * kPillarPad + (e & 3)];
                tot_gamma[e] = m >= 0 ? stage[m] * 1e-8 : 0.0;
            }
        }
        __syncthreads();   // every wave is done with the curve tables; reuse the LDS for the reduction
        double* red = reinterpret_cast<double*>(smem_raw);   // [waves][kAggStride]
        double* mine = red + wave * kAggStride;
        if (lane == 0) mine[0] = total.pv;
        if (lane < kPillarPad) mine[1 + lane] = DELTA ? total.delta * 1e-4 : 0.0;
#pragma unroll
        for (int e = 0; e < kGammaPerLane; ++e) {
            const int r = 4 * bi + (e >> 2), q = 4 * bj + (e & 3);
            mine[1 + kPillarPad + r * kPillarPad + q] = GAMMA ? tot_gamma[GAMMA ? e : 0] : 0.0;
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

// Fixed-order sum of the block partials -> agg[1 + P + P*P]: one wavefront per output, lanes stride over
// the blocks, then a fixed butterfly - the aggregate does not depend on scheduling.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* partials, int n_blocks, int P,
                                                               double* agg) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int n_out = 1 + P + P * P;
    if (i >= n_out) return;
    int src;
    if (i == 0) src = 0;
    else if (i < 1 + P) src = i;
    else { const int r = (i - 1 - P) / P, q = (i - 1 - P) % P; src = 1 + kPillarPad + r * kPillarPad + q; }
    double s = 0.0;
    for (int b = lane; b < n_blocks; b += 64) s += partials[static_cast<size_t>(b) * kAggStride + src];
    s = wave_sum(s);
    if (lane == 0) agg[i] = s;
}

template <bool DELTA, bool GAMMA>
void launch_epl(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, int n_blocks, size_t lds,
                hipStream_t stream) {
    dim3 grid(n_blocks), block(kBlockThreads);
    if (!GAMMA) {
        hipLaunchKernelGGL((price_fast_kernel<DELTA, false, 1>), grid, block, lds, stream, cv, tr, out);
        return;
    }
    switch (cv.epl) {
        case 3: hipLaunchKernelGGL((price_fast_kernel<true, true, 3>), grid, block, lds, stream, cv, tr, out); break;
        case 4: hipLaunchKernelGGL((price_fast_kernel<true, true, 4>), grid, block, lds, stream, cv, tr, out); break;
        case 6: hipLaunchKernelGGL((price_fast_kernel<true, true, 6>), grid, block, lds, stream, cv, tr, out); break;
        default: hipLaunchKernelGGL((price_fast_kernel<true, true, 9>), grid, block, lds, stream, cv, tr, out); break;
    }
}

}  // namespace

size_t fast_kernel_lds_bytes(const CurveDev& cv, bool gamma) {
    const size_t stage_stride = gamma ? std::max<size_t>((cv.Eu + 1) & ~1, 2 * kPillarPad) : 0;
    size_t doubles = static_cast<size_t>(cv.K) + 2 * cv.Kc + static_cast<size_t>(cv.Kcore + 1) * cv.pc_pad +
                     (gamma ? static_cast<size_t>(cv.Kcore + 1) * (cv.Ec + 1) : 0) +
                     kWavesPerBlock * stage_stride + 64 * 9;   // slack: convexity rows are read 64*EPL wide
    size_t tables = sizeof(MiniKnot) * cv.n_mini + sizeof(double) * doubles +
                    sizeof(int16_t) * (2 * static_cast<size_t>(cv.K) + cv.Kc + (gamma ? kPillarPad * kPillarPad : 0));
    size_t reduce = sizeof(double) * kWavesPerBlock * kAggStride;
    size_t need = tables > reduce ? tables : reduce;
    return (need + 15) & ~static_cast<size_t>(15);
}

hipError_t launch_price_fast(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, bool want_delta,
                             bool want_gamma, int n_blocks, hipStream_t stream) {
    const size_t lds = fast_kernel_lds_bytes(cv, want_gamma);
    if (want_gamma) launch_epl<true, true>(cv, tr, out, n_blocks, lds, stream);
    else if (want_delta) launch_epl<true, false>(cv, tr, out, n_blocks, lds, stream);
    else launch_epl<false, false>(cv, tr, out, n_blocks, lds, stream);
    return hipGetLastError();
}

hipError_t launch_reduce_partials(const double* partials, int n_blocks, int P, double* agg, hipStream_t stream) {
    const int n_out = 1 + P + P * P;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((n_out + 3) / 4), dim3(256), 0, stream, partials, n_blocks, P,
                       agg);
    return hipGetLastError();
}

hipError_t set_general_kernel_lds_limit(size_t bytes);

hipError_t set_kernel_lds_limits(size_t general_bytes, size_t fast_bytes) {
    hipError_t e = set_general_kernel_lds_limit(general_bytes);
    if (e != hipSuccess) return e;
    const void* fns[] = {
        reinterpret_cast<const void*>(&price_fast_kernel<true, true, 3>),
        reinterpret_cast<const void*>(&price_fast_kernel<true, true, 4>),
        reinterpret_cast<const void*>(&price_fast_kernel<true, true, 6>),
        reinterpret_cast<const void*>(&price_fast_kernel<true, true, 9>),
        reinterpret_cast<const void*>(&price_fast_kernel<true, false, 1>),
        reinterpret_cast<const void*>(&price_fast_kernel<false, false, 1>),
    };
    for (const void* f : fns) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(fast_bytes));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace adr
