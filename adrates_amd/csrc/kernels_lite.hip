// CDNA4 (gfx950) LITE kernel: PV and the pillar delta ladder of OIS trades when no gamma is requested
// (BASELINE.json configs[1]: PV + 32-pillar delta).
//
// Same mathematics as kernels_fast.hip (reference: cavour/market/position/engine.py:2414-2448 fixed leg, :2639-2728
// float leg, :2541-2561 / :2899-2919 value and delta assembly; lookups: cavour/market/curves/interpolator_ad.py:186-249):
// cash flows are folded into nodes (time, coefficient), every node is w = c * exp(ba L[ka] + bb L[kb]), and
//
//   PV = sum_nodes w,      dPV/dr = sum_nodes w (ba LJ[ka] + bb LJ[kb])  =  sum_knots c_k LJ[k]
//
// i.e. the reverse sweep stops at the knots: each node leaves two (coefficient, knot row) entries and the ladder is a
// sparse combination of rows of LJ = d ln(knot DF) / d(par rates).
//
// Why a kernel of its own.  Without the gamma state the pass is bound by vector-ALU issue, not by LDS or HBM
// (2 trades per wavefront at ~400 VALU instructions per trade), so the mapping is chosen to
// spend as few wave-instructions per trade as possible:
//   * a wavefront prices FOUR trades at a time, 16 lanes each, and a trade's coupons arrive as rows of 16 slots
//     (15 coupons + a spare lane for the leg's start node; longer trades are 2 to 9 consecutive rows), so
//     half-empty 32-slot rows are neither loaded nor computed on: ~0.95 KB of row data per trade on the benchmark
//     portfolio instead of 1.3 KB, fetched as 16-byte-per-lane loads of pair-interleaved arrays;
//   * lanes = coupons for folding, lookup and exp; then every lane leaves its node as two 16-byte entries
//     {w * b, byte offset of the knot's LJ row} in the wave's LDS slot and lane l accumulates pillars 2l and 2l + 1:
//     per entry one broadcast b128 read, one b128 read of the row pair, two FMAs - no per-node decoding, no short-end
//     special cases (the dense 32-wide LJ of all reachable knots fits LDS once the gamma tables are not needed);
//   * neighbour moves and the 16-lane PV sum are DPP row operations (VALU, no LDS traffic).
// Any curve of the three schemes qualifies (no packed layout, any pillar count up to 32).
// W64 instantiations (round 3): curves of 33-64 pillars - the 64-wide Jacobian table of the wide layout (curve_tables.hpp),
// four pillars per lane (two b128 reads of the row and four FMAs per entry), block partials in the wide route's record.
//
// XC instantiation (round 4): the FOREIGN LEG of a book of cross-currency swaps in one launch (`Engine._compute_xccy`'s second
// `_float_leg_jax` call, cavour/market/position/engine.py:1640-1733: forwards off the foreign OIS curve, discounting on the
// XCCY curve).  The payment-lag rows carry the leg as it is - payment times in the XCCY curve's day count, accrual times in
// the leg's own - and the kernel holds TWO curves' tables: a coupon is N ((D_f(ts) / D_f(te) - 1) + s a) D_x(tp), looked up
// once (D_x(tp); D_f(te); D_f(ts) from the previous lane), and leaves entries for the foreign-rate ladder (N D_x R on the
// knots of ts / te) AND for the basis ladder (the amount times D_x on the knots of tp); the notional exchanges are fixed
// flows on the XCCY curve.  No weights from a host pre-pass, no second batch: two ladders and the PV from one read.
//
// KNOT instantiations (round 4): the AGGREGATE-ONLY mode - Portfolio.compute's single ladder (cavour/market/portfolio/
// portfolio.py:39-66), no per-trade output.  The reference's chain rule, jac.T @ hess_dfs @ jac + sum_k g_k hess[k]
// (engine.py:2551-2567), is linear in the knot-space gradient g and Hessian hess_dfs of a trade, so summed over a book it
// needs only their SUMS: in log space, per node w = c exp(ba L[ka] + bb L[kb]),
//     w_k  += w b_k                      (first order, per knot)
//     D_k  += w b_k^2,   O_ka += w ba bb (second order: a node's two knots are neighbours in the compact knot order)
// - five numbers per node into a per-wave table over the reachable knots (LDS atomic adds; a wave's own instructions
// execute in order, so the sums do not depend on scheduling), summed over waves and blocks in a fixed order, and ONE
// projection per launch (kernels_knot.hip): delta = 1e-4 LJ^T w, gamma = 1e-8 (LJ^T W LJ + sum_k w_k LC_k).  No Jacobian
// table in LDS, no per-trade sweep, any pillar count.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "curve_lookup.hpp"
#include "kernels.hpp"

namespace adr {

namespace {

constexpr int kBlockThreads = kLiteThreads;
constexpr int kWavesPerBlock = kBlockThreads / 64;
#ifndef ADR_LITE_XC_THREADS
#define ADR_LITE_XC_THREADS 768                   // the two-curve instantiation: 167 VGPRs, three waves per SIMD in ONE block per CU
#endif                                            // (at the 128 registers of four waves it spills 160 bytes per lane: +20 % time)
constexpr int kXcThreads = ADR_LITE_XC_THREADS;
#ifndef ADR_LITE_LAG_THREADS
#define ADR_LITE_LAG_THREADS 768                  // the payment-lag rows' PV + delta instantiation likewise (145 VGPRs; 68 bytes of spills at 128: +8 %)
#endif
constexpr int kLagThreads = ADR_LITE_LAG_THREADS;
#ifndef ADR_LITE_W64_THREADS
#define ADR_LITE_W64_THREADS 768                  // the 64-wide ladders likewise (157 VGPRs; 104-176 bytes of spills at 128: 0.105 -> 0.064 ms per 100 k trades)
#endif
constexpr int kW64Threads = ADR_LITE_W64_THREADS;
// threads per block of an instantiation: 512 (four waves per SIMD in two blocks per CU) unless its registers ask for more
constexpr int lite_block_threads(bool delta, bool lag, bool w64, int knot, bool xc) {
    return xc ? kXcThreads : ((delta && knot == 0 && w64) ? kW64Threads : ((delta && knot == 0 && lag) ? kLagThreads : kLiteThreads));
}
constexpr int L = kLiteSlots;                 // lanes per trade
constexpr int G = 64 / L;                     // trades per wavefront
constexpr int kRecBytesPerWave = G * (2 * L + 2) * 16;  // per group: two 16-byte entries per lane + a pad entry
constexpr int kBatch = 4;                               // entries whose LDS operands are fetched together

typedef double nt_pair __attribute__((ext_vector_type(2)));

// DPP moves inside rows of 16 lanes (= one trade): lane i receives lane i + 1 / i - 1 of its row; the row's last /
// first lane keeps its own value.
__device__ __forceinline__ double row_next(double x) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), 0x101, 0xf, 0xf, false);   // row_shl:1
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), 0x101, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_prev(double x) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), 0x111, 0xf, 0xf, false);   // row_shr:1
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), 0x111, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double row_perm(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// Sum over the 16 lanes of a row, result in every lane: quad butterflies, then the two mirrors.
__device__ __forceinline__ double row_sum(double x) {
    x += row_perm<0xB1>(x);      // quad_perm [1,0,3,2]
    x += row_perm<0x4E>(x);      // quad_perm [2,3,0,1]
    x += row_perm<0x141>(x);     // row_half_mirror
    x += row_perm<0x140>(x);     // row_mirror
    return x;
}

#ifdef ADR_STAMPS
#define ADR_STAMP(slot_) do { const unsigned long long now_ = clock64(); stamp_sum[slot_] += now_ - stamp_t; stamp_t = now_; } while (0)
#else
#define ADR_STAMP(slot_) do {} while (0)
#endif

__device__ __forceinline__ void wave_lds_sync() {     // same-wave LDS hand-off, see kernels_fast.hip
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

struct CurveLds {
    const double* x;            // [K]
    const double* log_df;       // [Kc]
    const double* inv_x;        // [Kc]
    const int16_t* lut;         // [n_lut][2]
    int n_lut;
    const int16_t* first_of;    // [K]
    const int16_t* compact_of;  // [K]
    const double* inv_dx;       // [K]  1 / (x[i] - x[i-1]); 0 for i = 0 and where the two knot times coincide
    int K, method;
};


// LINDF: LINEAR_FWD_RATES - D = ba d_a + bb d_b, a node is two single-knot exponentials and its two entries carry the
// two amounts (kernels_fast.hip, `lindf`)
// LAG: the rows hold trades whose coupons accrue to a date other than their payment date and / or carry a per-coupon
// notional multiplier (`te_w`): a coupon is the ratio node N w D(ts) D(tp) / D(te) - one exponential, three lookups -
// plus the payment node -N w (1 - spread a) D(tp).  The first-order sum is linear in a node's knot weights, so the
// lane simply leaves three pairs of entries (ts: +, te: -, tp: + with both amounts) in three sweeps; no telescoping.
// NSEG: segments of the row table the kernel looks at (3 covers tables of at most three distinct row counts - every
// table of trades without payment lag; kLiteSegments otherwise)
// KNOT: 0 = per-trade ladders; 1 = aggregate-only, first order (w_k); 2 = aggregate-only with the second-order sums
// XC: foreign-leg rows of cross-currency swaps on two curves (cv: the foreign OIS curve, cx: the XCCY curve)
template <bool DELTA, bool LINDF, bool LAG, int NSEG, bool W64 = false, int KNOT = 0, bool XC = false>
__global__ __launch_bounds__(lite_block_threads(DELTA, LAG, W64, KNOT, XC), lite_block_threads(DELTA, LAG, W64, KNOT, XC) == kBlockThreads
                                                                                  ? kLiteWavesPerSimd : lite_block_threads(DELTA, LAG, W64, KNOT, XC) / 256)
void price_lite_kernel(CurveDev cv, LiteRowsDev tr,
                                                                                       OutputsDev out, CurveDev cx) {
    static_assert(!XC || (LAG && DELTA && !W64 && KNOT == 0), "the two-curve mode works on payment-lag rows, per trade");
    static_assert(KNOT == 0 || (DELTA && !W64), "aggregate-only mode: any pillar count (no 64-wide Jacobian table is needed)");
    constexpr int kBlockThreads = lite_block_threads(DELTA, LAG, W64, KNOT, XC);      // (these two hide the namespace's)
    constexpr int kWavesPerBlock = kBlockThreads / 64;
    constexpr int PW = W64 ? kWidePad : kPillarPad;       // pillars per row of the Jacobian table
    constexpr int PPL = PW / L;                           // pillars per lane: 2, or 4 on the 64-wide table
    // KNOT: tables per wave - w; D and the pair bands P[d - 1][k] = sum over pairs of knots (k, k + d): one band (the old O)
    // for rows without ratio nodes, kKnotBand for payment-lag rows, whose nodes couple knots of up to three intervals
    constexpr int BAND = LAG ? kKnotBand : 1;
    constexpr int NT = KNOT == 2 ? 2 + BAND : 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // LDS carve-up: per-wave entry slots (16-byte aligned) or knot tables, doubles, int16 tables
    unsigned char* s_rec = smem_raw;
    double* s_lj = reinterpret_cast<double*>(s_rec + ((DELTA && !KNOT) ? kWavesPerBlock * kRecBytesPerWave : 0));
    double* s_knot = s_lj;                                // KNOT: [waves][NT][Kc]
    double* s_x = s_lj + (KNOT ? kWavesPerBlock * NT * cv.Kc : (DELTA ? cv.Kc * PW : 0));
    double* s_log = s_x + cv.K;
    double* s_invx = s_log + cv.Kc;
    double* s_invdx = s_invx + cv.Kc;
    // XC: the second curve's Jacobian table and search arrays behind the first one's doubles
    double* s_lj2 = s_invdx + cv.K;
    double* s_x2 = s_lj2 + (XC ? cx.Kc * kPillarPad : 0);
    double* s_log2 = s_x2 + (XC ? cx.K : 0);
    double* s_invx2 = s_log2 + (XC ? cx.Kc : 0);
    double* s_invdx2 = s_invx2 + (XC ? cx.Kc : 0);
    int16_t* s_first = reinterpret_cast<int16_t*>(s_invdx2 + (XC ? cx.K : 0));
    int16_t* s_comp = s_first + cv.K;
    int16_t* s_lut = s_comp + cv.K;
    int16_t* s_first2 = s_lut + 2 * cv.n_lut;
    int16_t* s_comp2 = s_first2 + (XC ? cx.K : 0);
    int16_t* s_lut2 = s_comp2 + (XC ? cx.K : 0);

    if (KNOT) {
        for (int i = threadIdx.x; i < kWavesPerBlock * NT * cv.Kc; i += kBlockThreads) s_knot[i] = 0.0;
    } else if (DELTA) {
        const double* lj_src = W64 ? cv.lj64 : cv.lj;
        for (int i = threadIdx.x; i < cv.Kc * PW; i += kBlockThreads) s_lj[i] = lj_src[i];
    }
    for (int i = threadIdx.x; i < cv.K; i += kBlockThreads) {
        s_x[i] = cv.x[i];
        {   // jnp.interp returns fp[i-1] when |dx| <= 2^-104: weight 0
            const double dx = i > 0 ? cv.x[i] - cv.x[i - 1] : 0.0;
            s_invdx[i] = fabs(dx) <= 0x1p-104 ? 0.0 : 1.0 / dx;
        }
        s_first[i] = cv.first_of[i];
        s_comp[i] = cv.compact_of[i];
    }
    for (int i = threadIdx.x; i < cv.Kc; i += kBlockThreads) {
        s_log[i] = cv.log_df[i];
        s_invx[i] = cv.inv_x[i];
    }
    for (int i = threadIdx.x; i < 2 * cv.n_lut; i += kBlockThreads) s_lut[i] = cv.lut[i];
    if (XC) {
        for (int i = threadIdx.x; i < cx.Kc * kPillarPad; i += kBlockThreads) s_lj2[i] = cx.lj[i];
        for (int i = threadIdx.x; i < cx.K; i += kBlockThreads) {
            s_x2[i] = cx.x[i];
            const double dx = i > 0 ? cx.x[i] - cx.x[i - 1] : 0.0;
            s_invdx2[i] = fabs(dx) <= 0x1p-104 ? 0.0 : 1.0 / dx;
            s_first2[i] = cx.first_of[i];
            s_comp2[i] = cx.compact_of[i];
        }
        for (int i = threadIdx.x; i < cx.Kc; i += kBlockThreads) { s_log2[i] = cx.log_df[i]; s_invx2[i] = cx.inv_x[i]; }
        for (int i = threadIdx.x; i < 2 * cx.n_lut; i += kBlockThreads) s_lut2[i] = cx.lut[i];
    }
    __syncthreads();

    CurveLds c;
    c.x = s_x; c.log_df = s_log; c.inv_x = s_invx; c.lut = s_lut; c.n_lut = cv.n_lut;
    c.first_of = s_first; c.compact_of = s_comp; c.inv_dx = s_invdx; c.K = cv.K; c.method = cv.method;
    CurveLds c2 = c;             // XC: the XCCY curve
    if (XC) {
        c2.x = s_x2; c2.log_df = s_log2; c2.inv_x = s_invx2; c2.lut = s_lut2; c2.n_lut = cx.n_lut;
        c2.first_of = s_first2; c2.compact_of = s_comp2; c2.inv_dx = s_invdx2; c2.K = cx.K; c2.method = cx.method;
    }

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane / L, l = lane % L;
    const int gbase = g * L;
    const int P = cv.P;
    // per wave: [4 groups][2 entries per lane + 1 pad entry] x 16 bytes; the pad shifts each group by 4 banks, so the
    // four groups' broadcast reads of one entry index do not collide
    unsigned char* rec_wave = s_rec + wave * kRecBytesPerWave;
    const unsigned char* rec_group = rec_wave + g * ((2 * L + 1) * 16);
    unsigned char* rec_mine = rec_wave + (g * (2 * L + 1) + 2 * l) * 16;
    const unsigned char* lj_lane = reinterpret_cast<const unsigned char*>(s_lj) + l * (8 * PPL);    // pillars PPL l .. PPL l + PPL - 1
    const unsigned char* lj2_lane = reinterpret_cast<const unsigned char*>(s_lj2) + l * 16;         // XC: pillars 2l, 2l + 1 of the second ladder
    double* knot_w = s_knot + wave * (NT * cv.Kc);        // KNOT: this wave's tables
    double* knot_d = knot_w + (NT > 1 ? cv.Kc : 0);
    double* knot_o = knot_w + (NT > 1 ? 2 * cv.Kc : 0);      // band d at knot_o + (d - 1) Kc
    // second-order sum of a pair of knot weights of one node: w ci cj on (ki, kj) and on (kj, ki).  Equal knots: the diagonal
    // takes both; neighbours up to BAND apart: the pair bands; farther (a long accrual period on a dense short end): the
    // launch's dense overflow matrix in global memory (rare; the only sum whose order depends on scheduling).  The value-time
    // knot (0) carries no sensitivity and is left out - it is also the knot farthest from everything.
    auto pair_add = [&](int ki, double ci, int kj, double cj, double om) {
        const double v = om * ci * cj;
        const bool on = ki != 0 && kj != 0 && v != 0.0;
        const int lo = ki < kj ? ki : kj, d = ki < kj ? kj - ki : ki - kj;
        // one predicated LDS add for the diagonal and the bands (no branch per case: the cases differ in address and factor)
        double* at = d == 0 ? knot_d + lo : knot_o + (d - 1) * cv.Kc + lo;
        if (on && d <= BAND) __hip_atomic_fetch_add(at, d == 0 ? 2.0 * v : v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (__ballot(on && d > BAND)) {
            if (on && d > BAND) {
                unsafeAtomicAdd(out.knot_overflow + static_cast<size_t>(lo) * cv.Kc + (lo + d), v);
                out.knot_overflow[static_cast<size_t>(cv.Kc) * cv.Kc] = 1.0;       // "the matrix is in use": the projection scans it only then
            }
        }
    };
    double tot_pv = 0.0, tot_d0 = 0.0, tot_d1 = 0.0, tot_d2 = 0.0, tot_d3 = 0.0;

    // ---------------------------------------------------------------------------------------------------
    // Main loop over (unit, row) steps.  A unit is 4 trade slots (one per group of 16 lanes); its trades have R rows
    // each (R is uniform inside a segment).  The inputs of the NEXT step - the row slice and, at a unit's first row,
    // the per-trade scalars - are requested before the current step is worked on, so the HBM round trip overlaps a
    // whole step of lookups, exponentials and ladder work instead of heading every step's dependency chain.
    // All index arithmetic is 32-bit and wave-uniform (scalar unit) except one multiply-add per lane.
    const uint32_t n_units = static_cast<uint32_t>(tr.n_units);
    const uint32_t wave_stride = gridDim.x * kWavesPerBlock;
    uint32_t seg_u0[NSEG], seg_at0[NSEG];
#pragma unroll
    for (int k = 0; k < NSEG; ++k) {
        seg_u0[k] = static_cast<uint32_t>(tr.seg_unit0[k]);
        seg_at0[k] = static_cast<uint32_t>(tr.seg_row0[k]) * L;
    }
    auto segment = [&](uint32_t u, int& R, uint32_t& at0) {       // rows per trade; array index of the unit's first slot
        R = tr.seg_rows[0];
        uint32_t unit0 = seg_u0[0];
        at0 = seg_at0[0];
#pragma unroll
        for (int k = 1; k < NSEG; ++k)
            if (u >= seg_u0[k]) { R = tr.seg_rows[k]; unit0 = seg_u0[k]; at0 = seg_at0[k]; }
        at0 += (u - unit0) * (G * L) * R;
    };
    double nx_tp = 0.0, nx_ts = 0.0, nx_al = 0.0, nx_xtp = 0.0, nx_xpay = 0.0, nx_N = 0.0, nx_spread = 0.0;
    double nx_te = 0.0, nx_w = 1.0;
    int nx_meta = 0, nx_trade = -1;
    auto request_row = [&](uint32_t at0, int R, int r) {            // group g's row r: R rows per trade, 16 slots per row
        const uint32_t at = at0 + r * L + __umul24(g, R * L) + l;
        const nt_pair a = __builtin_nontemporal_load(reinterpret_cast<const nt_pair*>(tr.tp_ts) + at);
        const nt_pair b = __builtin_nontemporal_load(reinterpret_cast<const nt_pair*>(tr.al_xtp) + at);
        nx_tp = a.x; nx_ts = a.y; nx_al = b.x; nx_xtp = b.y;
        nx_xpay = __builtin_nontemporal_load(tr.xpay + at);
        if (LAG) {
            const nt_pair cw = __builtin_nontemporal_load(reinterpret_cast<const nt_pair*>(tr.te_w) + at);
            nx_te = cw.x; nx_w = cw.y;
        }
    };
    auto request_trade = [&](uint32_t u) {
        // (scalar loads of the unit's four slots + a pick by group were tried: the wave then waits for the scalar
        // data in this phase - slower than four broadcast vector loads)
        const LiteTrade* rec = tr.slot + (u * G + g);
        const nt_pair a = *reinterpret_cast<const nt_pair*>(&rec->notional);
        const int2 b = *reinterpret_cast<const int2*>(&rec->meta);
        nx_N = a.x; nx_spread = a.y; nx_meta = b.x; nx_trade = b.y;
    };

    uint32_t unit = blockIdx.x * kWavesPerBlock + wave;
    int R = 1, r = 0;
    uint32_t at0 = 0;
    if (unit < n_units) {
        segment(unit, R, at0);
        request_trade(unit);
        request_row(at0, R, 0);
    }
    double N = 0.0, spread = 0.0, sl = 1.0, sf = 1.0;
    int n_flt = 0, n_fix = 0, t = -1;
    bool live = false;
    double pv = 0.0, d0 = 0.0, d1 = 0.0, e0 = 0.0, e1 = 0.0;     // two accumulator pairs: shorter FMA chains
    double d2 = 0.0, d3 = 0.0, e2 = 0.0, e3 = 0.0;               // W64: pillars 4l + 2, 4l + 3
#ifdef ADR_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = clock64();
#endif
    while (unit < n_units) {
        // ---- this step's inputs
        const double tp = nx_tp, ts = nx_ts, al = nx_al, xtp = nx_xtp, xpay = nx_xpay;
        const double te = nx_te, cw = nx_w;
#ifdef ADR_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        ADR_STAMP(0);   // waiting for the step's inputs
        if (r == 0) {
            N = nx_N; spread = nx_spread; t = nx_trade; live = t >= 0;
            n_flt = nx_meta & 0x1ff; n_fix = (nx_meta >> 9) & 0x1ff;            // (up to 390 coupons per leg: 26 rows)
            sl = (nx_meta & 0x40000) ? -1.0 : 1.0; sf = (nx_meta & 0x80000) ? -1.0 : 1.0;
            pv = d0 = d1 = e0 = e1 = 0.0;
            d2 = d3 = e2 = e3 = 0.0;
        }
        // ---- request the next step's inputs
        const bool last_row = r + 1 == R;
        const uint32_t next_unit = last_row ? unit + wave_stride : unit;
        int next_R = R;
        uint32_t next_at0 = at0;
        if (last_row && next_unit < n_units) {
            segment(next_unit, next_R, next_at0);
            request_trade(next_unit);
            request_row(next_at0, next_R, 0);
        } else if (!last_row) {
            request_row(at0, R, r + 1);
        }
        ADR_STAMP(5);   // requesting the next step's inputs
        {
            const int m_flt = min(max(n_flt - r * kLiteCoupons, 0), kLiteCoupons);   // coupons of this row
            const int m_fix = min(max(n_fix - r * kLiteCoupons, 0), kLiteCoupons);

            // ---- fold the coupons into nodes (lane l = coupon l of the row); same rules as kernels_fast.hip
            const bool in = live && l < m_flt;
            const double ntp = row_next(tp), nts = row_next(ts), nal = row_next(al);
            const double ptp = row_prev(tp);
            const bool valid = in && tp >= 0.0;
            const bool accrues = al > 0.0;
            double a_pay = valid ? sl * N * (spread * al - (accrues ? 1.0 : 0.0)) : 0.0;
            if (!LAG && in && l + 1 < m_flt && nal > 0.0 && ntp >= 0.0 && nts == tp) a_pay += sl * N;
            const bool fix_in = live && l < m_fix;
            const bool fix_merged = fix_in && in && xtp == tp;
            if (fix_merged && xtp > 0.0) a_pay = fma(sf, xpay, a_pay);
            bool own_start = !LAG && valid && accrues && !(l > 0 && ptp == ts);
            const bool own_fixed = fix_in && !fix_merged && xtp > 0.0 && sf * xpay != 0.0;

            double qt = tp, qa = a_pay;
            bool qon = in && a_pay != 0.0;
            {   // the row's first start node moves to the row's spare lane (m_flt <= 15 < 16: there always is one)
                const unsigned mine = static_cast<unsigned>(__ballot(own_start) >> gbase) & 0xffffu;
                const int src = gbase + (mine ? __builtin_ctz(mine) : 0);
                const double st = __shfl(ts, src, 64);
                if (mine != 0 && l == m_flt) { qt = st; qa = sl * N; qon = true; }
                if (mine != 0 && lane == src) own_start = false;
            }
            const bool more_starts = __ballot(own_start) != 0, more_fixed = __ballot(own_fixed) != 0;
            ADR_STAMP(1);   // folding

            // ---- reverse sweep to the knots: every lane leaves its node as two entries {coefficient, byte offset of the
            // knot's LJ row}.  Knot 0 is the value-time knot, whose row is all zero, so idle lanes and single-knot
            // nodes need no flags.  (On curves whose pillar dates are runs of duplicate knots a node's right-hand knot -
            // the first of a run - is never the next node's left-hand knot - the last of that run -, so merging
            // neighbours' entries buys nothing: tried, slower.)
            auto sweep = [&](bool on, double ca, double cb, int ka, int kb, double ba, double bb, bool plain_node = true,
                             bool second = false) {
                if constexpr (KNOT != 0) {
                    // aggregate-only: the node's five numbers into this wave's knot tables.  (ca, cb) = w (ba, bb) - under
                    // LINEAR_FWD_RATES the two single-knot amounts, each with weight 1 on its own knot and no cross term.
                    if (on) {
                        __hip_atomic_fetch_add(knot_w + ka, ca, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        if (cb != 0.0) __hip_atomic_fetch_add(knot_w + kb, cb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        if (KNOT == 2 && plain_node) {          // (the parts of a ratio node leave their second-order sums together)
                            __hip_atomic_fetch_add(knot_d + ka, LINDF ? ca : ca * ba, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            if (cb != 0.0) {
                                __hip_atomic_fetch_add(knot_d + kb, LINDF ? cb : cb * bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                if (!LINDF) __hip_atomic_fetch_add(knot_o + ka, ca * bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            }
                        }
                    }
                    return;
                }
                const int off_a = ka * (PW * 8), off_b = kb * (PW * 8);
#if ADR_LITE_SWEEP_PRIO
                __builtin_amdgcn_s_setprio(ADR_LITE_SWEEP_PRIO);
#endif
                __builtin_amdgcn_wave_barrier();
                {
                    double2* wp = reinterpret_cast<double2*>(rec_mine);
                    wp[0] = make_double2(ca, __hiloint2double(0, off_a));
                    wp[1] = make_double2(cb, __hiloint2double(0, off_b));
                }
                wave_lds_sync();
                unsigned long long any = __ballot(on);
                any |= any >> 32; any |= any >> 16;
                const unsigned rows_any = static_cast<unsigned>(any) & 0xffffu;
                if (!rows_any) {
#if ADR_LITE_SWEEP_PRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
                    return;
                }
                const int n_e = 2 * (32 - __builtin_clz(rows_any));      // entries up to the highest live lane of any row
                for (int e = 0; e < n_e; e += kBatch) {                   // lanes past n_e wrote zero entries
                    double2 rc[kBatch], rw[kBatch], rv[W64 ? kBatch : 1];
#pragma unroll
                    for (int i = 0; i < kBatch; ++i) rc[i] = *reinterpret_cast<const double2*>(rec_group + (e + i) * 16);
#pragma unroll
                    for (int i = 0; i < kBatch; ++i) {
                        rw[i] = *reinterpret_cast<const double2*>((XC && second ? lj2_lane : lj_lane) + __double2loint(rc[i].y));
                        if (W64) rv[i] = *reinterpret_cast<const double2*>(lj_lane + __double2loint(rc[i].y) + 16);
                    }
#pragma unroll
                    for (int i = 0; i < kBatch; ++i) {
                        if (XC && second) {                       // (wave-uniform) the second curve's ladder
                            if (i & 1) { e2 = fma(rc[i].x, rw[i].x, e2); e3 = fma(rc[i].x, rw[i].y, e3); }
                            else { d2 = fma(rc[i].x, rw[i].x, d2); d3 = fma(rc[i].x, rw[i].y, d3); }
                            continue;
                        }
                        if (i & 1) { e0 = fma(rc[i].x, rw[i].x, e0); e1 = fma(rc[i].x, rw[i].y, e1); }
                        else { d0 = fma(rc[i].x, rw[i].x, d0); d1 = fma(rc[i].x, rw[i].y, d1); }
                        if (W64) {
                            if (i & 1) { e2 = fma(rc[i].x, rv[i].x, e2); e3 = fma(rc[i].x, rv[i].y, e3); }
                            else { d2 = fma(rc[i].x, rv[i].x, d2); d3 = fma(rc[i].x, rv[i].y, d3); }
                        }
                    }
                }
#if ADR_LITE_SWEEP_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            };
            if constexpr (XC) {
                // ---- foreign leg of a cross-currency swap (lane = coupon): forwards off the first curve (the foreign OIS
                // curve), discounting on the second (the XCCY curve):  N ((R - 1) + s a) D_x(tp),  R = D_f(ts) / D_f(te)
                const bool ratio = valid && accrues;
                Lookup qs{0, 0, 0.0, 0.0}, qe{0, 0, 0.0, 0.0}, qp{0, 0, 0.0, 0.0};
                // (a factor's value and weights under LINEAR_FWD_RATES: as in the payment-lag rows below)
                constexpr double kNone = LINDF ? 1.0 : 0.0;
                auto factor = [&](const CurveLds& cc, Lookup& q) {
                    if constexpr (LINDF) {
                        const double da = q.ba * exp(cc.log_df[q.ka]);
                        const double db = q.bb != 0.0 ? q.bb * exp(cc.log_df[q.kb]) : 0.0;
                        const double f = da + db, inv = 1.0 / f;
                        q.ba = da * inv; q.bb = db * inv;
                        return f;
                    } else {
                        return fma(q.ba, cc.log_df[q.ka], q.bb * cc.log_df[q.kb]);
                    }
                };
                double ls = kNone, le = kNone, lp = kNone;
                const bool paid_later = valid && tp != 0.0;          // (paid AT the value time: D_x = 1, no basis sensitivity)
                if (paid_later) { qp = curve_lookup<true>(c2, tp); lp = factor(c2, qp); }
                const double prev_ratio = row_prev(ratio ? 1.0 : 0.0), prev_te = row_prev(te);
                const bool chained = ratio && l > 0 && prev_ratio != 0.0 && prev_te == ts;      // (as in the payment-lag rows below)
                bool own_ts = ratio && !chained;
                const unsigned starts = static_cast<unsigned>(__ballot(own_ts) >> gbase) & 0xffffu;
                const int src = gbase + (starts ? __builtin_ctz(starts) : 0);
                const double st = __shfl(ts, src, 64);
                const bool spare = starts != 0 && l == m_flt;                                   // (the spare-lane pass of the rows below)
                if (starts != 0 && lane == src) own_ts = false;
                if (ratio || spare) { qe = curve_lookup<true>(c, spare ? st : te); le = factor(c, qe); }
                const double le_prev = row_prev(le), ls_spare = __shfl(le, gbase + m_flt, 64);
                if (starts != 0 && lane == src) ls = ls_spare;
                if (own_ts) { qs = curve_lookup<true>(c, ts); ls = factor(c, qs); }
                if (chained) ls = le_prev;
                const double w_not = sl * N * cw;
                const double dx = LINDF ? lp : exp(lp);               // D_x(tp) / D_x(0)
                const double R = ratio ? (LINDF ? ls / le : exp(ls - le)) : 1.0;
                const double om_r = ratio ? w_not * dx * R : 0.0;     // what the foreign rates move: N D_x(tp) D_f(ts) / D_f(te)
                double amount = valid ? w_not * ((R - 1.0) + spread * al) : 0.0;
                if (fix_merged && xtp > 0.0) amount = fma(sf, xpay, amount);          // an exchange paid on the coupon's date
                const double om_b = amount * dx;                      // the coupon's value; what the basis spreads move
                pv += om_b;
                ADR_STAMP(2);   // lookups + exp
                {
                    const double next_flag = row_next(chained ? 1.0 : 0.0), next_om = row_next(om_r);
                    const bool next_chained = l + 1 < L && next_flag != 0.0;
                    double om_e = (next_chained ? next_om : 0.0) - om_r;
                    const double om_src = __shfl(om_r, src, 64);
                    if (spare) om_e = om_src;                         // the spare lane's entries: the start of coupon `src`
                    if (__ballot(own_ts)) sweep(own_ts, om_r * qs.ba, om_r * qs.bb, qs.ka, qs.kb, qs.ba, qs.bb, false);
                    sweep(ratio || spare, om_e * qe.ba, om_e * qe.bb, qe.ka, qe.kb, qe.ba, qe.bb, false);
                    sweep(paid_later, om_b * qp.ba, om_b * qp.bb, qp.ka, qp.kb, qp.ba, qp.bb, false, true);
                }
                ADR_STAMP(3);   // entries + ladders
            } else if (LAG) {
                // ---- the row's coupons: ratio node + payment node (lane = coupon); three lookups, two exponentials
                const bool ratio = valid && accrues;
                Lookup qs{0, 0, 0.0, 0.0}, qe{0, 0, 0.0, 0.0}, qp{0, 0, 0.0, 0.0};
                // A factor's value: its LOG discount factor (log-linear schemes: the node is one exponential of their signed sum), or
                // under LINEAR_FWD_RATES the discount factor F = ba D_a + bb D_b itself, with the lookup's weights replaced by the
                // factor's effective log weights g = (ba D_a, bb D_b) / F = d ln F / d ln D - what every first-order sum below uses.
                constexpr double kNone = LINDF ? 1.0 : 0.0;           // a factor that is not there: D = 1
                auto factor = [&](Lookup& q) {
                    if constexpr (LINDF) {
                        const double da = q.ba * exp(c.log_df[q.ka]);
                        const double db = q.bb != 0.0 ? q.bb * exp(c.log_df[q.kb]) : 0.0;
                        const double f = da + db, inv = 1.0 / f;
                        q.ba = da * inv; q.bb = db * inv;
                        return f;
                    } else {
                        return fma(q.ba, c.log_df[q.ka], q.bb * c.log_df[q.kb]);
                    }
                };
                double ls = kNone, le = kNone, lp = kNone;
                // (a coupon "paid" at the value time - the weighted coupons of a leg projected on another curve, DESIGN.md
                // section 9 - needs no lookup: D(0) = 1 and the value-time knot carries no sensitivity)
                const bool paid_later = valid && tp != 0.0;
                if (paid_later) { qp = curve_lookup<true>(c, tp); lp = factor(qp); }
                // Accrual periods tile a leg: coupon j starts where coupon j - 1 ends.  Such a start time IS the previous
                // lane's end time - the same lookup - so the lane takes the previous lane's log discount factor through DPP
                // instead of searching again, and its (+) entries join the previous lane's (-) entries on the same two knots:
                // one lookup, one exponential and one pair of entries per coupon instead of two; only a row's first coupon
                // (and a coupon after an accrual gap) looks its start time up and leaves start entries of its own.
                // The row's first start of its own (the leg's first coupon, as a rule) is looked up by the row's SPARE lane
                // (rows hold 15 coupons) in the same pass as the accrual ends, and leaves its entries from there in the same
                // sweep: a pass of its own cost a whole lookup and a whole hand-off for one lane in sixteen.
                // (every cross-lane move sits outside conditionals: inside one, its source lanes may be masked off)
                const double prev_ratio = row_prev(ratio ? 1.0 : 0.0), prev_te = row_prev(te);
                const bool chained = ratio && l > 0 && prev_ratio != 0.0 && prev_te == ts;
                bool own_ts = ratio && !chained;
                const unsigned starts = static_cast<unsigned>(__ballot(own_ts) >> gbase) & 0xffffu;
                const int src = gbase + (starts ? __builtin_ctz(starts) : 0);
                const double st = __shfl(ts, src, 64);
                const bool spare = starts != 0 && l == m_flt;
                if (starts != 0 && lane == src) own_ts = false;
                if (ratio || spare) { qe = curve_lookup<true>(c, spare ? st : te); le = factor(qe); }
                const double le_prev = row_prev(le), ls_spare = __shfl(le, gbase + m_flt, 64);
                if (starts != 0 && lane == src) ls = ls_spare;
                if constexpr (KNOT == 2) {          // the second-order sums below want the start's knots and weights in the coupon's lane
                    const int ska = __shfl(qe.ka, gbase + m_flt, 64), skb = __shfl(qe.kb, gbase + m_flt, 64);
                    const double sba = __shfl(qe.ba, gbase + m_flt, 64), sbb = __shfl(qe.bb, gbase + m_flt, 64);
                    if (starts != 0 && lane == src) { qs.ka = ska; qs.kb = skb; qs.ba = sba; qs.bb = sbb; }
                }
                if (own_ts) { qs = curve_lookup<true>(c, ts); ls = factor(qs); }    // (after an accrual gap: rare)
                if (chained) ls = le_prev;
                const double w_not = sl * N * cw;
                const double om_r = ratio ? (LINDF ? w_not * ls * lp / le : w_not * exp(ls - le + lp)) : 0.0;
                double a_q = valid ? w_not * (spread * al - (accrues ? 1.0 : 0.0)) : 0.0;
                if (fix_merged && xtp > 0.0) a_q = fma(sf, xpay, a_q);
                const double om_p = valid ? a_q * (LINDF ? lp : exp(lp)) : 0.0;
                pv += om_r + om_p;
                ADR_STAMP(2);   // lookups + exp
                if constexpr (KNOT == 2) {
                    // second-order sums of the ratio node omega_r exp(L(ts) - L(te) + L(tp)): u = b(ts) - b(te) + b(tp) over up to
                    // six knots (a chained start sits on the previous lane's end knots), W += omega_r u u^T; and of the payment
                    // node omega_p on b(tp).  Six squares and fifteen pairs at most; zero weights and the value-time knot drop out.
                    const int pka = __shfl_up(qe.ka, 1, 64), pkb = __shfl_up(qe.kb, 1, 64);          // (outside conditionals)
                    const double pba = row_prev(qe.ba), pbb = row_prev(qe.bb);
                    int kk[6] = {chained ? pka : qs.ka, chained ? pkb : qs.kb, qe.ka, qe.kb, qp.ka, qp.kb};
                    double cc[6] = {chained ? pba : qs.ba, chained ? pbb : qs.bb, -qe.ba, -qe.bb, qp.ba, qp.bb};
                    // a payment time a few days behind the accrual end sits on the same two knots: its weights join the end's
                    // (four terms instead of six: ten sums instead of twenty-one)
                    // LINEAR_FWD_RATES: a factor F = ba D_a + bb D_b is not an exponential of a linear form; its logarithm has the
                    // Hessian g_a g_b (e_a - e_b)(e_a - e_b)^T (g: the effective weights), so every factor of a node of value omega
                    // adds +- omega g_a g_b of that matrix (-: the accrual end, which divides)
                    auto factor_curvature = [&](int ka, int kb, double ga, double gb, double om) {
                        const double w = om * ga * gb;
                        if (ka != 0 && w != 0.0) __hip_atomic_fetch_add(knot_d + ka, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        if (kb != 0 && w != 0.0) __hip_atomic_fetch_add(knot_d + kb, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        pair_add(ka, 1.0, kb, -1.0, w);
                    };
                    if constexpr (LINDF) {
                        factor_curvature(kk[0], kk[1], cc[0], cc[1], om_r);
                        factor_curvature(qe.ka, qe.kb, qe.ba, qe.bb, -om_r);
                        factor_curvature(qp.ka, qp.kb, qp.ba, qp.bb, om_r + om_p);        // the ratio node's and the payment node's
                    }
                    if (qp.ka == qe.ka && qp.kb == qe.kb) { cc[2] += cc[4]; cc[3] += cc[5]; cc[4] = 0.0; cc[5] = 0.0; }
                    if (ratio) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) {
                            if (kk[i] != 0 && cc[i] != 0.0)
                                __hip_atomic_fetch_add(knot_d + kk[i], om_r * cc[i] * cc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#pragma unroll
                            for (int j = i + 1; j < 6; ++j) pair_add(kk[i], cc[i], kk[j], cc[j], om_r);
                        }
                    }
                    if (paid_later) {
                        if (qp.ka != 0) __hip_atomic_fetch_add(knot_d + qp.ka, om_p * qp.ba * qp.ba, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        if (qp.kb != 0 && qp.bb != 0.0) __hip_atomic_fetch_add(knot_d + qp.kb, om_p * qp.bb * qp.bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        pair_add(qp.ka, qp.ba, qp.kb, qp.bb, om_p);
                    }
                }
                if (DELTA) {
                    // the end entries carry -omega of their own coupon and +omega of the next one when that one is chained to it
                    const double next_flag = row_next(chained ? 1.0 : 0.0), next_om = row_next(om_r);
                    const bool next_chained = l + 1 < L && next_flag != 0.0;
                    double om_e = (next_chained ? next_om : 0.0) - om_r;
                    const double om_src = __shfl(om_r, src, 64);
                    if (spare) om_e = om_src;                         // the spare lane's entries: the start of coupon `src`
                    if (__ballot(own_ts)) sweep(own_ts, om_r * qs.ba, om_r * qs.bb, qs.ka, qs.kb, qs.ba, qs.bb, false);
                    sweep(ratio || spare, om_e * qe.ba, om_e * qe.bb, qe.ka, qe.kb, qe.ba, qe.bb, false);
                    sweep(paid_later, (om_r + om_p) * qp.ba, (om_r + om_p) * qp.bb, qp.ka, qp.kb, qp.ba, qp.bb, false);
                }
                ADR_STAMP(3);   // entries + ladder
            }
            for (int pass = LAG ? 1 : 0; pass < 3; ++pass) {
                if (pass == 1) {            // fixed coupons that did not merge into a float payment node
                    if (!more_fixed) continue;
                    qt = xtp; qa = sf * xpay; qon = own_fixed;
                } else if (pass == 2) {     // further start nodes (accrual gaps)
                    if (!more_starts) break;
                    qt = ts; qa = sl * N; qon = own_start;
                }
                double ca = 0.0, cb = 0.0, q_ba = 0.0, q_bb = 0.0;
                int q_ka = 0, q_kb = 0;
                if (qon) {
                    const Lookup q = XC ? curve_lookup<true>(c2, qt) : curve_lookup<true>(c, qt);      // XC: the exchanges, on the XCCY curve
                    const double* lg = XC ? c2.log_df : c.log_df;
                    if (LINDF) {
                        ca = qa * q.ba * exp(lg[q.ka]);
                        cb = q.bb != 0.0 ? qa * q.bb * exp(lg[q.kb]) : 0.0;
                        pv += ca + cb;
                    } else {
                        const double omega = qa * exp(fma(q.ba, lg[q.ka], q.bb * lg[q.kb]));
                        pv += omega;
                        ca = omega * q.ba; cb = omega * q.bb;
                    }
                    q_ka = q.ka; q_kb = q.kb; q_ba = q.ba; q_bb = q.bb;
                }
                ADR_STAMP(2);   // lookup + exp
                if (!DELTA) continue;
                sweep(qon, ca, cb, q_ka, q_kb, q_ba, q_bb, true, XC);
                ADR_STAMP(3);   // entries + ladder
            }
        }
        if (!last_row) { ++r; continue; }

        // ---- results of the unit's trades
#if ADR_LITE_OUT_PRIO
        __builtin_amdgcn_s_setprio(ADR_LITE_OUT_PRIO);
#endif
        if (KNOT) {
            tot_pv += pv;                                  // per lane; the lanes are summed once, at the end
        } else {
            pv = row_sum(pv);
            if (live && l == 0) {
                if (out.pv) out.pv[t] = pv;
                tot_pv += pv;
            }
        }
        if (!KNOT && DELTA && live) {
            d0 = (d0 + e0) * 1e-4; d1 = (d1 + e1) * 1e-4;
            tot_d0 += d0; tot_d1 += d1;
            if (W64 || XC) { d2 = (d2 + e2) * 1e-4; d3 = (d3 + e3) * 1e-4; tot_d2 += d2; tot_d3 += d3; }
            if (XC && out.delta2) {      // the second curve's ladder (its pillar count is even: the XCCY tables are padded)
                double* dst2 = out.delta2 + static_cast<int64_t>(t) * cx.P + 2 * l;
                if ((cx.P & 1) == 0) {
                    if (2 * l < cx.P) { nt_pair pr; pr.x = d2; pr.y = d3; __builtin_nontemporal_store(pr, reinterpret_cast<nt_pair*>(dst2)); }
                } else {
                    if (2 * l < cx.P) dst2[0] = d2;
                    if (2 * l + 1 < cx.P) dst2[1] = d3;
                }
            }
            if (out.delta) {
                double* dst = out.delta + static_cast<int64_t>(t) * P + PPL * l;
                if ((P & 1) == 0) {
                    if (PPL * l < P) { nt_pair pr; pr.x = d0; pr.y = d1; __builtin_nontemporal_store(pr, reinterpret_cast<nt_pair*>(dst)); }
                    if (W64 && PPL * l + 2 < P) { nt_pair pr; pr.x = d2; pr.y = d3; __builtin_nontemporal_store(pr, reinterpret_cast<nt_pair*>(dst + 2)); }
                } else {                 // odd pillar count: rows are not 16-byte aligned
                    if (PPL * l < P) dst[0] = d0;
                    if (PPL * l + 1 < P) dst[1] = d1;
                    if (W64 && PPL * l + 2 < P) dst[2] = d2;
                    if (W64 && PPL * l + 3 < P) dst[3] = d3;
                }
            }
        }
        unit = next_unit; R = next_R; at0 = next_at0; r = 0;
#if ADR_LITE_OUT_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        ADR_STAMP(4);   // outputs
    }
#ifdef ADR_STAMPS
    if (out.stamps && lane == 0) {
        unsigned long long* dst = out.stamps + (static_cast<size_t>(blockIdx.x) * kWavesPerBlock + wave) * 8;
        for (int i = 0; i < 8; ++i) dst[i] = stamp_sum[i];
    }
#endif

    // ------------------------------------------------------------------------ block partial of the aggregate
    if constexpr (KNOT != 0) {
        // knot tables: the waves' tables summed in wave order, the PV lanes by a fixed butterfly -> [pv, w[Kc], D[Kc], P[BAND][Kc]]
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) tot_pv += __shfl_xor(tot_pv, off, 64);
        __syncthreads();                                   // every wave's adds have been issued and (same-CU LDS) performed
        double* red_pv = reinterpret_cast<double*>(s_x);   // the search arrays are no longer needed
        if (lane == 0) red_pv[wave] = tot_pv;
        __syncthreads();
        double* dst = out.knot_partials + static_cast<size_t>(blockIdx.x) * (1 + (2 + BAND) * cv.Kc);
        for (int i = threadIdx.x; i < NT * cv.Kc; i += kBlockThreads) {
            double s_ = 0.0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) s_ += s_knot[w * (NT * cv.Kc) + i];
            dst[1 + i] = s_;
        }
        if (threadIdx.x == 0) {
            double s_ = 0.0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) s_ += red_pv[w];
            dst[0] = s_;
        }
        return;
    }
    if (out.block_partials) {
#pragma unroll
        for (int off = L; off < 64; off <<= 1) {
            tot_pv += __shfl_xor(tot_pv, off, 64);
            tot_d0 += __shfl_xor(tot_d0, off, 64);
            tot_d1 += __shfl_xor(tot_d1, off, 64);
            if (W64 || XC) { tot_d2 += __shfl_xor(tot_d2, off, 64); tot_d3 += __shfl_xor(tot_d3, off, 64); }
        }
        __syncthreads();   // every wave is done with the tables; reuse the LDS
        constexpr int kRed = 1 + PW + (XC ? kPillarPad : 0);        // XC: the second ladder behind the first
        double* red = reinterpret_cast<double*>(smem_raw);          // [waves][kRed]
        if (lane == 0) red[wave * kRed] = tot_pv;
        if (g == 0) {
            red[wave * kRed + 1 + PPL * l] = tot_d0; red[wave * kRed + 2 + PPL * l] = tot_d1;
            if (W64) { red[wave * kRed + 3 + PPL * l] = tot_d2; red[wave * kRed + 4 + PPL * l] = tot_d3; }
            if (XC) { red[wave * kRed + 1 + PW + 2 * l] = tot_d2; red[wave * kRed + 2 + PW + 2 * l] = tot_d3; }
        }
        __syncthreads();
        if (XC) {       // the second ladder's block record: [0, delta2[32], ...] in the second partials array
            double* dst2 = out.block_partials2 + static_cast<size_t>(blockIdx.x) * kAggStride;
            if (threadIdx.x <= kPillarPad) {
                double s2 = 0.0;
                if (threadIdx.x > 0) {
#pragma unroll
                    for (int w = 0; w < kWavesPerBlock; ++w) s2 += red[w * kRed + PW + threadIdx.x];
                }
                dst2[threadIdx.x] = s2;
            }
        }
        // (W64: the wide route's record - pv, delta[64], then the packed gamma entries, which no delta request reads)
        double* dst = out.block_partials + static_cast<size_t>(blockIdx.x) * (W64 ? 1 + kWidePad + cv.wide_nch * kWideChunk : kAggStride);
        if (threadIdx.x < 1 + PW) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) s += red[w * kRed + threadIdx.x];
            dst[threadIdx.x] = (DELTA || threadIdx.x == 0) ? s : 0.0;     // the gamma part is not read (no gamma requested)
        }
    }
}

}  // namespace

namespace {
using LiteFn = void (*)(CurveDev, LiteRowsDev, OutputsDev, CurveDev);

template <bool DELTA, bool LINDF, bool LAG, bool W64>
LiteFn lite_kernel_nseg(bool many) {
    return many ? &price_lite_kernel<DELTA, LINDF, LAG, kLiteSegments, W64> : &price_lite_kernel<DELTA, LINDF, LAG, 3, W64>;
}

template <bool W64>
LiteFn lite_kernel_w(bool delta, bool lindf, bool lag, bool many_segments) {
    if (lag && lindf) return delta ? lite_kernel_nseg<true, true, true, W64>(many_segments) : lite_kernel_nseg<false, true, true, W64>(many_segments);
    if (lag) return delta ? lite_kernel_nseg<true, false, true, W64>(many_segments) : lite_kernel_nseg<false, false, true, W64>(many_segments);
    if (lindf) return delta ? lite_kernel_nseg<true, true, false, W64>(many_segments) : lite_kernel_nseg<false, true, false, W64>(many_segments);
    return delta ? lite_kernel_nseg<true, false, false, W64>(many_segments) : lite_kernel_nseg<false, false, false, W64>(many_segments);
}

LiteFn lite_kernel(bool delta, bool lindf, bool lag, bool many_segments, bool w64 = false) {
    return w64 ? lite_kernel_w<true>(delta, lindf, lag, many_segments) : lite_kernel_w<false>(delta, lindf, lag, many_segments);
}

// aggregate-only instantiations (KNOT = 1: first order, 2: with the second-order sums); lag: the payment-lag rows
template <int KNOT>
LiteFn knot_kernel_k(bool lindf, bool many, bool lag) {
    if (lag && lindf) return many ? &price_lite_kernel<true, true, true, kLiteSegments, false, KNOT> : &price_lite_kernel<true, true, true, 3, false, KNOT>;
    if (lag) return many ? &price_lite_kernel<true, false, true, kLiteSegments, false, KNOT> : &price_lite_kernel<true, false, true, 3, false, KNOT>;
    if (lindf) return many ? &price_lite_kernel<true, true, false, kLiteSegments, false, KNOT> : &price_lite_kernel<true, true, false, 3, false, KNOT>;
    return many ? &price_lite_kernel<true, false, false, kLiteSegments, false, KNOT> : &price_lite_kernel<true, false, false, 3, false, KNOT>;
}
LiteFn knot_kernel(bool gamma, bool lindf, bool many, bool lag) {
    return gamma ? knot_kernel_k<2>(lindf, many, lag) : knot_kernel_k<1>(lindf, many, lag);
}
}  // namespace

int knot_record_doubles(const CurveDev& cv, bool lag) { return 1 + (2 + (lag ? kKnotBand : 1)) * cv.Kc; }

size_t knot_kernel_lds_bytes(const CurveDev& cv, bool gamma, bool lag) {
    size_t bytes = sizeof(double) * static_cast<size_t>(kWavesPerBlock) * (gamma ? 2 + (lag ? kKnotBand : 1) : 1) * cv.Kc;
    bytes += sizeof(double) * (2 * static_cast<size_t>(cv.K) + 2 * cv.Kc);
    bytes += sizeof(int16_t) * (2 * static_cast<size_t>(cv.K) + 2 * static_cast<size_t>(cv.n_lut));
    return (bytes + 15) & ~static_cast<size_t>(15);
}

int knot_kernel_threads() { return kBlockThreads; }

hipError_t launch_price_knot(const CurveDev& cv, const LiteRowsDev& tr, const OutputsDev& out, bool want_gamma, int n_blocks,
                             hipStream_t stream) {
    const bool lag = tr.te_w != nullptr;                                // payment-lag rows: ratio nodes, log-linear schemes
    if (!out.knot_partials || (lag && want_gamma && !out.knot_overflow)) return hipErrorInvalidValue;
    const size_t lds = knot_kernel_lds_bytes(cv, want_gamma, lag);
    hipLaunchKernelGGL(knot_kernel(want_gamma, cv.method == 2, tr.n_seg > 3, lag), dim3(n_blocks), dim3(kBlockThreads), lds, stream, cv, tr, out, CurveDev{});
    return hipGetLastError();
}

int lite_kernel_threads(const CurveDev& cv, bool delta, bool lag) { return lite_block_threads(delta, lag, cv.T > 1, 0, false); }

size_t lite_kernel_lds_bytes(const CurveDev& cv, bool delta, bool lag) {
    const int kWavesPerBlock = lite_kernel_threads(cv, delta, lag) / 64;
    const size_t pw = cv.T > 1 ? kWidePad : kPillarPad;           // more than 32 pillars: the 64-wide table (W64 instantiations)
    size_t bytes = delta ? static_cast<size_t>(kWavesPerBlock) * kRecBytesPerWave + sizeof(double) * cv.Kc * pw : 0;
    bytes += sizeof(double) * (2 * static_cast<size_t>(cv.K) + 2 * cv.Kc);
    bytes += sizeof(int16_t) * (2 * static_cast<size_t>(cv.K) + 2 * static_cast<size_t>(cv.n_lut));
    const size_t reduce = sizeof(double) * kWavesPerBlock * (1 + pw);
    if (bytes < reduce) bytes = reduce;
    return (bytes + 15) & ~static_cast<size_t>(15);
}

hipError_t launch_price_lite(const CurveDev& cv, const LiteRowsDev& tr, const OutputsDev& out, bool want_delta,
                             int n_blocks, hipStream_t stream) {
    const size_t lds = lite_kernel_lds_bytes(cv, want_delta, tr.te_w != nullptr);
    const dim3 grid(n_blocks), block(lite_kernel_threads(cv, want_delta, tr.te_w != nullptr));
    if (cv.T > 1 && !cv.lj64) return hipErrorInvalidValue;             // 33-64 pillars: the wide layout's table
    hipLaunchKernelGGL(lite_kernel(want_delta, cv.method == 2, tr.te_w != nullptr, tr.n_seg > 3, cv.T > 1), grid, block, lds, stream, cv, tr, out, CurveDev{});
    return hipGetLastError();
}

// the foreign leg of a cross-currency book on two curves (XC instantiations): cv = the foreign OIS curve, cx = the XCCY curve
namespace {
LiteFn lite_xc_kernel(bool many, bool lindf = false) {
    if (lindf) return many ? &price_lite_kernel<true, true, true, kLiteSegments, false, 0, true> : &price_lite_kernel<true, true, true, 3, false, 0, true>;
    return many ? &price_lite_kernel<true, false, true, kLiteSegments, false, 0, true> : &price_lite_kernel<true, false, true, 3, false, 0, true>;
}
}  // namespace

size_t lite_xc_kernel_lds_bytes(const CurveDev& cv, const CurveDev& cx) {
    constexpr int waves = kXcThreads / 64;
    size_t bytes = lite_kernel_lds_bytes(cv, true, false) + static_cast<size_t>(waves - kWavesPerBlock) * kRecBytesPerWave;
    bytes += sizeof(double) * (static_cast<size_t>(cx.Kc) * kPillarPad + 2 * static_cast<size_t>(cx.K) + 2 * cx.Kc);
    bytes += sizeof(int16_t) * (2 * static_cast<size_t>(cx.K) + 2 * static_cast<size_t>(cx.n_lut));
    const size_t reduce = sizeof(double) * waves * (1 + 2 * kPillarPad);
    if (bytes < reduce) bytes = reduce;
    return (bytes + 15) & ~static_cast<size_t>(15);
}

int lite_xc_kernel_threads() { return kXcThreads; }

hipError_t launch_price_lite_xc(const CurveDev& cv, const CurveDev& cx, const LiteRowsDev& tr, const OutputsDev& out, int n_blocks,
                                hipStream_t stream) {
    // (both curves on LINEAR_FWD_RATES, or both on a log-linear scheme: the kernel is built for one kind of factor)
    if (!tr.te_w || (cv.method == 2) != (cx.method == 2) || cv.T > 1 || cx.T > 1 || !cv.lj || !cx.lj) return hipErrorInvalidValue;
    const size_t lds = lite_xc_kernel_lds_bytes(cv, cx);
    hipLaunchKernelGGL(lite_xc_kernel(tr.n_seg > 3, cv.method == 2), dim3(n_blocks), dim3(kXcThreads), lds, stream, cv, tr, out, cx);
    return hipGetLastError();
}

hipError_t set_lite_kernel_lds_limit(size_t bytes) {
    std::vector<const void*> fns;
    for (int d = 0; d < 2; ++d)
        for (int lin = 0; lin < 2; ++lin)
            for (int lag = 0; lag < 2; ++lag)
                for (int many = 0; many < 2; ++many)
                {
                    fns.push_back(reinterpret_cast<const void*>(lite_kernel(d != 0, lin != 0, lag != 0, many != 0, false)));
                    fns.push_back(reinterpret_cast<const void*>(lite_kernel(d != 0, lin != 0, lag != 0, many != 0, true)));
                }
    for (int g = 0; g < 2; ++g)
        for (int lin = 0; lin < 2; ++lin)
            for (int many = 0; many < 2; ++many) {
                fns.push_back(reinterpret_cast<const void*>(knot_kernel(g != 0, lin != 0, many != 0, false)));
                fns.push_back(reinterpret_cast<const void*>(knot_kernel(g != 0, lin != 0, many != 0, true)));
            }
    for (int lin = 0; lin < 2; ++lin) {
        fns.push_back(reinterpret_cast<const void*>(lite_xc_kernel(false, lin != 0)));
        fns.push_back(reinterpret_cast<const void*>(lite_xc_kernel(true, lin != 0)));
    }
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace adr
