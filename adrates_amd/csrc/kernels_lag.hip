// CDNA4 (gfx950) kernel for OIS whose coupons accrue to a date other than their payment date (payment lag) and / or
// carry a per-coupon notional multiplier (`row_te`, `row_w`: the weighted coupons of a leg projected on one curve and
// discounted on another) - PV, delta ladder and gamma matrix, curves with the packed hub layout.
//
// Reference: cavour/market/position/engine.py:2639-2728 (float leg with payment times != accrual end times),
// :2414-2448 (fixed leg), :2541-2576 / :2899-2934 (Greeks assembly); lookups cavour/market/curves/
// interpolator_ad.py:186-249.  Mathematics as in kernels_fast.hip: every PV term is w = c exp(sum_i b_i L[k_i]) and
//   dPV/dr = sum w v,   d2PV/dr2 = sum w (v v^T + sum_i b_i LC[k_i]),   v = sum_i b_i LJ[k_i].
//
// A coupon is  N w (D(ts)/D(te) - 1 + spread a) D(tp):
//   the RATIO node   w_r = N w D(ts) D(tp) / D(te)      v_r = v(ts) - v(te) + v(tp)    (one exponential, up to six knots)
//   the PAYMENT node w_p = -N w (1 - spread a) D(tp)    v_p = v(tp)                    (+ the fixed coupon paid that day).
// Accrual periods tile a leg (ts of coupon j+1 == te of coupon j) and te, tp are a few days apart, so in the regular
// case te_j, tp_j and ts_{j+1} sit between the same two knots - the coupon's DATE.  One walk record per date then does
// all of it: with (ua, ub) the Jacobian rows of the date's knots,
//   v_D = (p - e) . (ua, ub)      the ratio node's part on this date      v_r = v_S + v_D   (v_S carried from the previous date)
//   v_P = p . (ua, ub)            the payment node
//   v_S(next) = v_P - v_D = e . (ua, ub)    the NEXT ratio node's accrual-start part (same time as this accrual end)
// two rank-one updates (w_r v_r v_r^T, w_p v_P v_P^T), one convexity-row read with the summed first-order weights.
//
// Mapping.  As in kernels_fast.hip a wavefront prices two rows (trades) at a time, 32 lanes each, 768-thread blocks,
// three waves per SIMD (the old variant, kernels_fast.hip LAG, kept 256 registers per lane and ran two).  The coupons
// are walked SIXTEEN at a time with TWO lanes per coupon: the even lane looks up tp (search) and te (the bracket of tp
// re-checked: no second search), the odd lane ts; the pair exchanges through DPP and leaves one 64-byte date record -
// each lane writes its own 32-byte half, so the per-wave LDS slot is the plain kernel's.  Coupons the regular pattern does
// not cover (an accrual end and a payment time that straddle a knot, a start that is not the previous end, the first
// coupon of a 16-coupon chunk) are "irregular": their ratio node is walked as three single-time parts in a pass of its
// own (rare), as are the fixed coupons that merge with no float payment date.
//
// Special nodes.  A ratio node that couples a short-end interval with another interval creates gamma entries the
// packed ladder has no slot for (pairs of a short-end pillar m with pillars of the other interval).  For these the
// lane of pillar q keeps a "side row" per short-end pillar m: side[m][q] = sum w_r v_r[m] v_r[q] over special nodes (four
// rows in registers, a per-wave global scratch row beyond that).  After the trade's matrix has been stored the elements
// (m, q) and (q, m) WITHOUT a packed entry are overwritten with the side row - later stores of the same wave to the
// same address land later.  Book aggregates of the side rows go through per-wave rows in the same scratch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <vector>

#include "curve_lookup.hpp"
#include "kernels.hpp"

namespace adr {

namespace {

typedef double nt_pair __attribute__((ext_vector_type(2)));

#ifndef ADR_LAG2_THREADS
#define ADR_LAG2_THREADS kFastThreads
#endif
#define ADR_LAG_BOUNDS ADR_LAG2_THREADS
constexpr int kThreads = ADR_LAG2_THREADS;        // 768: three waves per SIMD, one block per CU (LDS)
constexpr int kWaves = kThreads / 64;
constexpr int G = 2, L = 32;                      // trades per wavefront, lanes per trade
constexpr int kSideSlots = 4;                     // side rows kept in registers
constexpr int kOutParts = 2;
constexpr int kScratchPerWave = 2 * kPillarPad * 64 + 24 * 64;     // doubles of global scratch per wavefront (see the kernel)

__device__ __forceinline__ double shfl_d(double x, int src) { return __shfl(x, src, 64); }

template <int CTRL>
__device__ __forceinline__ int dpp_mov_i(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double x) {
    return __hiloint2double(dpp_mov_i<CTRL>(__double2hiint(x)), dpp_mov_i<CTRL>(__double2loint(x)));
}
// pair broadcasts (quad_perm): the even / the odd lane of every lane pair to both lanes
constexpr int kFromEven = 0xA0;   // [0, 0, 2, 2]
constexpr int kFromOdd = 0xF5;    // [1, 1, 3, 3]
constexpr int kShr1 = 0x138, kShl1 = 0x130;       // wave_shr:1 (lane i gets lane i - 1), wave_shl:1 (lane i gets lane i + 1)

__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

struct CurveLds {
    const double* x;
    const double* log_df;
    const double* inv_x;
    const double* ljc;
    const double* lcc;
    const MiniKnot* mini;
    const int16_t* lut;
    int n_lut;
    const int16_t* first_of;
    const int16_t* compact_of;
    const int16_t* knot_class;
    int K, method, pc_pad, ec_stride;
};

__host__ __device__ constexpr int lag_slot_doubles(int epg) {
    const int epl = (epg * L + 63) / 64;
    const int hand_off = 64 * 4 + G * kPillarPad;
    const int staging = 64 * epl + 2;
    return hand_off > staging ? hand_off : staging;
}

#ifdef ADR_STAMPS
#define LAG_STAMP(slot_) do { const unsigned long long now_ = clock64(); stamp_sum[slot_] += now_ - stamp_t; stamp_t = now_; } while (0)
#else
#define LAG_STAMP(slot_) do {} while (0)
#endif

constexpr int kNullPair = static_cast<int>((static_cast<unsigned>(-2) << 16) | (static_cast<unsigned>(-2) & 0xffffu));
// record flags (high word of a record's class double)
constexpr int kTileNext = 1, kSpecial = 2, kFire = 4;

// STORE: per-trade gamma matrices are written; LONG: rows are chains of 32-coupon pieces (meta bit 18, kernels_fast.hip);
// EPG / CPG: packed entries per group lane and how many of them are core slots (exact hub variants: CPG = EPG - 2).
template <bool STORE, bool LONG, int EPG, int CPG>
__global__ __launch_bounds__(ADR_LAG_BOUNDS) void price_lag_kernel(CurveDev cv, TradesDev tr, OutputsDev out) {
    static_assert(CPG == EPG - 2, "hub layout: the last two slots hold the fringe pairs");
    constexpr int EPL = (EPG * L + 63) / 64;
    constexpr unsigned long long kGroupMask = (1ull << L) - 1;

    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int ec_stride = cv.Ec + 1;
    const int n_ljc = (cv.Kcore + 1) * cv.pc_pad;
    const int n_lcc = (cv.Kcore + 1) * ec_stride;
    const int n_slack = L * CPG > cv.Ec + 1 ? L * CPG - (cv.Ec + 1) : 0;
    constexpr int kRecDoubles = 64 * 4;
    constexpr int kSlotDoubles = lag_slot_doubles(EPG);
    MiniKnot* s_mini = reinterpret_cast<MiniKnot*>(smem_raw);
    double* s_x = reinterpret_cast<double*>(s_mini + cv.n_mini);
    double* s_log = s_x + cv.K;
    double* s_invx = s_log + cv.Kc;
    double* s_ljc = s_invx + cv.Kc;
    double* s_lcc = s_ljc + n_ljc;
    const int n_front = cv.K + 2 * cv.Kc + n_ljc + n_lcc + n_slack;
    double* s_slot = s_x + n_front + (n_front & 1);
    int16_t* s_first = reinterpret_cast<int16_t*>(s_slot + kWaves * kSlotDoubles);
    int16_t* s_comp = s_first + cv.K;
    int16_t* s_class = s_comp + cv.K;
    int16_t* s_lut = s_class + cv.Kc;
    {
        const double* src = reinterpret_cast<const double*>(cv.mini);
        double* dst = reinterpret_cast<double*>(s_mini);
        for (int i = threadIdx.x; i < cv.n_mini * 8; i += kThreads) dst[i] = src[i];
    }
    for (int i = threadIdx.x; i < cv.K; i += kThreads) {
        s_x[i] = cv.x[i];
        s_first[i] = cv.first_of[i];
        s_comp[i] = cv.compact_of[i];
    }
    for (int i = threadIdx.x; i < cv.Kc; i += kThreads) {
        s_log[i] = cv.log_df[i];
        s_invx[i] = cv.inv_x[i];
        s_class[i] = cv.knot_class[i];
    }
    for (int i = threadIdx.x; i < 2 * cv.n_lut; i += kThreads) s_lut[i] = cv.lut[i];
    for (int i = threadIdx.x; i < n_ljc; i += kThreads) s_ljc[i] = cv.ljc[i];
    for (int i = threadIdx.x; i < n_lcc; i += kThreads) s_lcc[i] = cv.lcc[i];
    for (int i = threadIdx.x; i < n_slack; i += kThreads) s_lcc[n_lcc + i] = 0.0;
    __syncthreads();

    CurveLds c;
    c.x = s_x; c.log_df = s_log; c.inv_x = s_invx; c.ljc = s_ljc; c.lcc = s_lcc; c.mini = s_mini;
    c.first_of = s_first; c.compact_of = s_comp; c.knot_class = s_class; c.lut = s_lut; c.n_lut = cv.n_lut;
    c.K = cv.K; c.method = cv.method; c.pc_pad = cv.pc_pad; c.ec_stride = ec_stride;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane / L, l = lane % L;
    const int gbase = g * L;
    const bool odd = (l & 1) != 0;
    double* slot = s_slot + wave * kSlotDoubles;
    double* rec = slot;                                         // records, 4 doubles per lane
    double* vbuf = slot + kRecDoubles + g * kPillarPad;         // this group's v, 32 doubles
    const int P = cv.P;
    const int bi = lane >> 3, bj = lane & 7;
    const int zero_row = cv.Kcore;

    // per-lane constants (hub layout: see kernels_fast.hip)
    const int col0 = cv.pillar_to_core[l];
    const bool short_end_lane = col0 == cv.Pc;                  // pillar l is outside the core (only short-end knots touch it)
    int vq[EPG];
#pragma unroll
    for (int i = 0; i < EPG; ++i) {
        const int e = l + L * i;
        vq[i] = e < cv.Eu ? cv.ent_pq[2 * e + 1] : 0;
    }
    const int hub_p = l < cv.Eu ? cv.ent_pq[2 * l] : 0;
    constexpr int kZeroEntry = 64 * EPL;
    // Output map of this lane (elements 2 lane, 2 lane + 1 of each 128-element band, kernels_fast.hip): read from the
    // curve's table when a unit's results are written - eight registers the walks do not have to carry
    auto load_out_map = [&](int (&mm)[8], int& beyond) {
        beyond = 0;
#pragma unroll
        for (int band = 0; band < 8; ++band) {
            const int raw = *reinterpret_cast<const int*>(cv.store_map + 2 * lane + band * 128);
            const int m0 = static_cast<int16_t>(raw & 0xffff), m1 = raw >> 16;
            if (m0 == -2) beyond |= 1 << band;
            mm[band] = (m0 < 0 ? kZeroEntry : m0) | ((m1 < 0 ? kZeroEntry : m1) << 16);
        }
    };

    // per-wave global scratch: [0] book totals of the side rows, [1] side rows beyond the register slots; [32][64] doubles each
    double* const scratch_wave = out.lag_scratch + (static_cast<size_t>(blockIdx.x) * kWaves + wave) * kScratchPerWave;   // (wave-uniform)
#define scratch_tot (scratch_wave + lane)
#define scratch_ovf (scratch_wave + kPillarPad * 64 + lane)
#define scratch_run (scratch_wave + 2 * kPillarPad * 64 + lane)      // [0] pv, [1] delta, [2 ..] packed gamma slices
    const bool want_agg = out.block_partials != nullptr;
    if (want_agg) {
#pragma unroll 4
        for (int m = 0; m < kPillarPad; ++m) scratch_tot[m * 64] = 0.0;
#pragma unroll
        for (int i = 0; i < 2 + EPG; ++i) scratch_run[i * 64] = 0.0;
    }
    unsigned tot_rows = 0;                      // (wave-uniform) pillars whose total row has been added to

    // (the running book totals - pv, delta, the packed gamma slices - live in the wave's scratch, not in registers: five
    // read-modify-writes of 512 bytes per unit, nothing waits for them)

    const int64_t n_units = (tr.n_rows + G - 1) / G;
    const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * kWaves;
    int64_t unit = static_cast<int64_t>(blockIdx.x) * kWaves + wave;

    // inputs of a unit, prefetched: the first sixteen coupons of the group's row in the arrangement the chunks work in
    // (lane pair c holds coupon c); coupons 16-31 are fetched when their chunk starts, the odd fixed coupon of a date of
    // its own in the fixed pass - nothing of a row stays in registers across the walks
    double nx_tp = 0.0, nx_ts = 0.0, nx_te = 0.0, nx_al = 0.0, nx_w = 1.0, nx_xtp = 0.0, nx_xpay = 0.0, nx_N = 0.0, nx_spread = 0.0;
    int nx_meta = 0, nx_trade = -1;
    // A chunk's first lane pair can be a LEAD pair: no coupon of its own, only the accrual start of the chunk's first coupon
    // (which otherwise would find no previous date record to ride on).  Chunks after a row's first have one (they hold
    // fifteen coupons), and so does the first chunk of a row that continues a chain.
    auto chunk_coupon = [&](bool lead_row, int chunk, int pair) {      // coupon (slot of the row) of a lane pair; the lead pair: its chunk's first
        const int base = lead_row ? 15 * chunk : (chunk == 0 ? 0 : 16 + 15 * (chunk - 1));
        const bool lead = lead_row || chunk > 0;
        return base + (lead ? max(pair - 1, 0) : pair);
    };
    auto load_unit = [&](int64_t u, bool lead_row) {
        const int64_t row = u * G + g;
        nx_tp = nx_ts = nx_te = nx_al = nx_xtp = nx_xpay = nx_N = nx_spread = 0.0;
        nx_w = 1.0;
        nx_meta = 0; nx_trade = -1;
        if (u < n_units && row < tr.n_rows) {
            const int64_t at = row * kRowSlots + chunk_coupon(lead_row, 0, l >> 1);
            nx_tp = __builtin_nontemporal_load(tr.row_tp + at);
            nx_ts = __builtin_nontemporal_load(tr.row_ts + at);
            nx_te = __builtin_nontemporal_load(tr.row_te + at);
            nx_al = __builtin_nontemporal_load(tr.row_alpha + at);
            if (tr.row_w) nx_w = __builtin_nontemporal_load(tr.row_w + at);
            nx_xtp = __builtin_nontemporal_load(tr.row_xtp + at);
            nx_xpay = __builtin_nontemporal_load(tr.row_xpay + at);
            nx_N = tr.row_notional[row]; nx_spread = tr.row_spread[row];
            nx_meta = tr.row_meta[row]; nx_trade = tr.row_trade[row];
        }
    };
    auto pin_next = [&]() {
        asm volatile("" : "+v"(nx_tp), "+v"(nx_ts), "+v"(nx_te), "+v"(nx_al), "+v"(nx_w), "+v"(nx_xtp), "+v"(nx_xpay),
                          "+v"(nx_N), "+v"(nx_spread), "+v"(nx_meta), "+v"(nx_trade));
    };
    load_unit(unit, false);
    pin_next();

    double pv_chain = 0.0, dacc_chain = 0.0, acc_chain[EPG];
    double side[kSideSlots];
    int side_pillar[kSideSlots];               // (wave-uniform) pillar of side row s, -1: free
    unsigned ovf_rows = 0;                     // (wave-uniform) pillars whose side row lives in the scratch
#pragma unroll
    for (int s = 0; s < kSideSlots; ++s) { side[s] = 0.0; side_pillar[s] = -1; }
    bool fresh = true;
#ifdef ADR_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = clock64();
#endif
    for (; unit < n_units; unit += wave_stride) {
        double pv_unit = 0.0, dacc_unit = 0.0, acc_unit[EPG];
        double& pv = LONG ? pv_chain : pv_unit;
        double& dacc = LONG ? dacc_chain : dacc_unit;
        double (&acc)[EPG] = LONG ? acc_chain : acc_unit;
        double c_tp = nx_tp, c_ts = nx_ts, c_te = nx_te, c_al = nx_al, c_w = nx_w, c_xtp = nx_xtp, c_xpay = nx_xpay;   // chunk 0
        const double N = nx_N, spread = nx_spread;
        const int t = nx_trade;
        const bool live = t >= 0;
        const int n_flt = nx_meta & 0xff, n_fix = (nx_meta >> 8) & 0xff;
        const double sl = (nx_meta & 0x10000) ? -1.0 : 1.0, sf = (nx_meta & 0x20000) ? -1.0 : 1.0;
        const bool more = LONG && (__builtin_amdgcn_readfirstlane(nx_meta) & 0x40000) != 0;
        if (!LONG || fresh) {
            pv = 0.0; dacc = 0.0;
#pragma unroll
            for (int i = 0; i < EPG; ++i) acc[i] = 0.0;
        }
#ifdef ADR_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        LAG_STAMP(0);   // waiting for the unit's inputs (and everything older in the memory queue)
        const bool lead_row = LONG && !fresh;       // (wave-uniform: both groups' chains are padded to one length)
        int n_chunks;
        {
            int m = live ? max(n_flt, n_fix) : 0;
            m = max(m, __shfl_xor(m, 32, 64));
            m = __builtin_amdgcn_readfirstlane(m);
            n_chunks = lead_row ? (m + 14) / 15 : (m <= 16 ? (m > 0 ? 1 : 0) : 1 + (m - 16 + 14) / 15);
        }
        // the second chunk's coupons, requested now: their round trip hides behind the first chunk's work (a load issued
        // later would also queue behind this wave's earlier result stores - vector memory operations retire in order)
        double d_tp = 0.0, d_ts = 0.0, d_te = 0.0, d_al = 0.0, d_w = 1.0, d_xtp = 0.0, d_xpay = 0.0;
        auto load_chunk = [&](int chunk) {
            d_tp = d_ts = d_te = d_al = d_xtp = d_xpay = 0.0; d_w = 1.0;
            if (live) {
                const int64_t at = (unit * G + g) * kRowSlots + min(chunk_coupon(lead_row, chunk, l >> 1), kRowSlots - 1);
                d_tp = tr.row_tp[at]; d_ts = tr.row_ts[at]; d_te = tr.row_te[at]; d_al = tr.row_alpha[at];
                if (tr.row_w) d_w = tr.row_w[at];
                d_xtp = tr.row_xtp[at]; d_xpay = tr.row_xpay[at];
            }
        };
        if (n_chunks > 1) load_chunk(1);
        unsigned long long fixed_b[3] = {0, 0, 0};   // fixed coupons that share no float payment date: per chunk, bit = the coupon's even lane

        // ---------------------------------------------------------------- shared pieces of the two walks
        int carry_row = zero_row;
        double carry_w = 0.0;
        auto lc_row_pass = [&](int row, double w) {
            double lr[CPG];
            const double* src = c.lcc + __mul24(row, c.ec_stride) + l;      // (hub layout: identity row positions)
#pragma unroll
            for (int i = 0; i < CPG; ++i) lr[i] = src[L * i];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CPG; ++i) acc[i] = fma(w, lr[i], acc[i]);
        };
        // Jacobian entries of this lane's pillar at the two knots of a record (classes: >= 0 core row, -2 nothing, <= -3
        // short-end record)
        auto jacobian_rows = [&](int ca, int cb, double& ua, double& ub) {
            const int ra = ca >= 0 ? ca : zero_row, rb = cb >= 0 ? cb : zero_row;
            ua = c.ljc[__mul24(ra, c.pc_pad) + col0];
            ub = c.ljc[__mul24(rb, c.pc_pad) + col0];
            const bool mini_a = ca <= -3, mini_b = cb <= -3;
            if (__ballot(mini_a || mini_b)) {
                if (mini_a) {
                    const MiniKnot& m = c.mini[-3 - ca];
                    ua = l == m.p[0] ? m.lj[0] : (l == m.p[1] ? m.lj[1] : 0.0);
                }
                if (mini_b) {
                    const MiniKnot& m = c.mini[-3 - cb];
                    ub = l == m.p[0] ? m.lj[0] : (l == m.p[1] ? m.lj[1] : 0.0);
                }
                return true;
            }
            return false;
        };
        auto mini_convexity = [&](int ca, int cb, double coef_a, double coef_b) {
#pragma unroll
            for (int side_ = 0; side_ < 2; ++side_) {
                const int cls = side_ == 0 ? ca : cb;
                const bool mine = cls <= -3;
                if (!__ballot(mine)) continue;
                const MiniKnot& m = c.mini[mine ? (-3 - cls) : 0];
                const double coef = mine ? (side_ == 0 ? coef_a : coef_b) : 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int e = m.e[j];
                    if (j > 0 && !__ballot(mine && e >= 0)) continue;
                    const bool owner = mine && e >= 0 && (e % L) == l;
                    const double add = owner ? coef * m.lc[j] : 0.0;
                    const int at = e / L;
#pragma unroll
                    for (int i = 0; i < EPG; ++i) acc[i] += (i == at) ? add : 0.0;
                }
            }
        };
        // the right knot's convexity weight is carried to the next record, whose left knot it usually is (kernels_fast.hip)
        auto convexity_coef = [&](int ra, int rb, double coef_a, double coef_b) {
            if (__ballot(carry_row != zero_row && carry_row != ra)) {
                const bool flush = carry_row != ra;
                lc_row_pass(flush ? carry_row : zero_row, flush ? carry_w : 0.0);
                if (flush) { carry_row = zero_row; carry_w = 0.0; }
            }
            const double coa = coef_a + (carry_row == ra ? carry_w : 0.0);
            carry_row = rb; carry_w = coef_b;
            return coa;
        };
        // rank-one update om vv vv^T through the group's LDS buffer; WITH_ROW: the convexity row `row` x coa in the same batches
        auto rank_one = [&](auto with_row, double om, double vv_, double coa, int row) {
            constexpr bool WITH_ROW = decltype(with_row)::value;
            __builtin_amdgcn_wave_barrier();
            vbuf[l] = vv_;
            wave_lds_sync();
            __builtin_amdgcn_s_setprio(ADR_RANK_PRIO);
            const double* rowa = c.lcc + __mul24(row, c.ec_stride) + l;
            constexpr int kBatch = ADR_FAST_BATCH;
            double hub_v = 0.0;
#pragma unroll
            for (int i0 = 0; i0 < EPG; i0 += kBatch) {
                double vv[kBatch], la[kBatch];
                if (i0 == 0) hub_v = vbuf[hub_p];
#pragma unroll
                for (int i = 0; i < kBatch; ++i) {
                    if (i0 + i >= EPG) continue;
                    vv[i] = vbuf[vq[i0 + i]];
                    if (WITH_ROW && i0 + i < CPG) la[i] = rowa[L * (i0 + i)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < kBatch; ++i) {
                    if (i0 + i >= EPG) continue;
                    const double u_i = i0 + i < CPG ? hub_v : vv_;      // fringe slots: the lane's own pillar first
                    double gsum = fma(om * u_i, vv[i], acc[i0 + i]);
                    if (WITH_ROW && i0 + i < CPG) gsum = fma(coa, la[i], gsum);
                    acc[i0 + i] = gsum;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(ADR_WALK_PRIO);
        };
        // short-end pillars of the knots of a class pair, as a bit mask (uniform in a group)
        auto pillar_mask = [&](int ca, int cb) {
            unsigned m = 0;
            if (ca <= -3) { const MiniKnot& k = c.mini[-3 - ca]; m |= 1u << k.p[0]; if (k.p[1] >= 0) m |= 1u << k.p[1]; }
            if (cb <= -3) { const MiniKnot& k = c.mini[-3 - cb]; m |= 1u << k.p[0]; if (k.p[1] >= 0) m |= 1u << k.p[1]; }
            return m;
        };
        // side rows of a special node (weight om, vector v_own at this lane's pillar, the group's vector in vbuf):
        // side[m][q] += om v[m] v[q] for every short-end pillar m of the node
        auto side_update = [&](bool is_special, double om, double v_own, unsigned mask) {
            const unsigned mine = is_special ? mask : 0u;
            unsigned todo = __builtin_amdgcn_readlane(mine, 0) | __builtin_amdgcn_readlane(mine, 32);
            while (todo) {
                const int m = __builtin_ctz(todo);
                todo &= todo - 1;
                const double coef = (mine >> m) & 1u ? om * vbuf[m] : 0.0;
                const double add = coef * v_own;
                int s_at = -1;
                if (!((ovf_rows >> m) & 1u)) {
#pragma unroll
                    for (int s = 0; s < kSideSlots; ++s) if (side_pillar[s] == m) s_at = s;
                    if (s_at < 0) {
#pragma unroll
                        for (int s = kSideSlots - 1; s >= 0; --s) if (side_pillar[s] < 0) s_at = s;
                        if (s_at >= 0) {
#pragma unroll
                            for (int s = 0; s < kSideSlots; ++s) if (s == s_at) side_pillar[s] = m;
                        }
                    }
                }
                if (s_at >= 0) {
#pragma unroll
                    for (int s = 0; s < kSideSlots; ++s) side[s] += (s == s_at) ? add : 0.0;
                } else {                                   // no register slot left: the wave's scratch row (zero between trades)
                    ovf_rows |= 1u << m;
                    scratch_ovf[m * 64] += add;
                }
            }
        };

        auto pair_has_mini = [](int pair) { return min(static_cast<int>(static_cast<int16_t>(pair & 0xffff)), pair >> 16) <= -3; };
        double vacc = 0.0;                        // the ratio node under construction (its parts walked so far)
        int prev_word = kNullPair;                // classes of the record that left vacc
        unsigned part_mask = 0;                   // (part walk) short-end pillars of the parts so far

        // ---------------------------------------------------------------- part walk: records {om, wa, wb, classes | flags}
        // every record adds its part to vacc, does its first-order and convexity work (both linear in the record), and a
        // record flagged kFire runs the rank-one update with the whole vacc.  FIRE_ALL: ordinary single-time nodes.
        auto part_walk = [&](unsigned long long rows) {       // `rows`: the group's record indices (both groups walk the union)
            int n = __builtin_ctzll(rows);
            rows &= rows - 1;
            const double2* rec_g = reinterpret_cast<const double2*>(rec + gbase * 4);
            double2 nx0 = rec_g[2 * n], nx1 = rec_g[2 * n + 1];
            __builtin_amdgcn_s_setprio(ADR_WALK_PRIO);
            while (true) {
                const bool has_next = rows != 0;
                const int n_next = has_next ? __builtin_ctzll(rows) : n;
                rows &= rows - 1;
                const double om = nx0.x, wa = nx0.y, wb = nx1.x;
                const int word = __double2loint(nx1.y), flags = __double2hiint(nx1.y);
                const int ca = static_cast<int16_t>(word & 0xffff), cb = word >> 16;
                nx0 = rec_g[2 * n_next]; nx1 = rec_g[2 * n_next + 1];
                n = n_next;
                __builtin_amdgcn_sched_barrier(0);
                const int ra = ca >= 0 ? ca : zero_row, rb = cb >= 0 ? cb : zero_row;
                double ua, ub;
                const bool any_mini = jacobian_rows(ca, cb, ua, ub);
                const double v = fma(wb, ub, wa * ua);
                vacc += v;
                dacc = fma(om, v, dacc);
                const double coa = convexity_coef(ra, rb, om * wa, om * wb);
                const bool fire = __builtin_amdgcn_readfirstlane(flags) & kFire;        // (the kind is the same for both groups)
                if (any_mini) part_mask |= pillar_mask(ca, cb);
                if (fire) {
                    rank_one(std::true_type{}, om, vacc, coa, ra);
                    const bool is_special = (flags & kSpecial) != 0;
                    if (__ballot(is_special)) side_update(is_special, om, vacc, part_mask);
                    vacc = 0.0;
                    part_mask = 0;
                } else {
                    lc_row_pass(ra, coa);
                }
                if (any_mini) mini_convexity(ca, cb, om * wa, om * wb);
                if (!has_next) break;
            }
            __builtin_amdgcn_s_setprio(0);
        };

        // ---------------------------------------------------------------- the float coupons, sixteen at a time
        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            const int q = chunk_coupon(lead_row, chunk, l >> 1);      // this lane pair's coupon
            const bool lead = (lead_row || chunk > 0) && (l >> 1) == 0;   // the lead pair: only the accrual start of coupon q counts
            if (chunk > 0) {
                if (chunk > 1) load_chunk(chunk);             // (a third chunk: rows of 32 coupons only)
                c_tp = d_tp; c_ts = d_ts; c_te = d_te; c_al = d_al; c_w = d_w; c_xtp = d_xtp; c_xpay = d_xpay;
            }
            const double tp_q = c_tp, te_q = lead ? c_ts : c_te, ts_q = c_ts, al_q = c_al, w_q = c_w, xtp_q = c_xtp, xpay_q = c_xpay;
            const bool have = live && q < n_flt && q < kRowSlots;
            const bool cin = have && tp_q >= 0.0, accr = al_q > 0.0;
            // even lane: tp, then te from the same bracket when it holds; odd lane: ts.  (A lead pair: the odd lane alone.)
            const double t1 = odd ? ts_q : tp_q;
            const bool look1 = cin && (odd ? accr : !lead);
            int cls1a = -2, cls1b = -2, cls2a = -2, cls2b = -2;
            double b1a = 0.0, b1b = 0.0, b2a = 0.0, b2b = 0.0, ell1 = 0.0, ell2 = 0.0;
            int j1 = 0;
            if (look1) {
                j1 = curve_first_later(c, t1);
                const Lookup lq = curve_lookup_at(c, t1, j1);
                b1a = lq.ba; b1b = lq.bb;
                cls1a = c.knot_class[lq.ka];
                cls1b = b1b != 0.0 ? c.knot_class[lq.kb] : -2;
                ell1 = fma(b1a, c.log_df[lq.ka], b1b * c.log_df[lq.kb]);
            }
            const bool look2 = !odd && cin && accr && !lead;
            if (look2) {
                const int j2 = curve_first_later_is(c, te_q, j1) ? j1 : curve_first_later(c, te_q);
                const Lookup lq = curve_lookup_at(c, te_q, j2);
                b2a = lq.ba; b2b = lq.bb;
                cls2a = c.knot_class[lq.ka];
                cls2b = b2b != 0.0 ? c.knot_class[lq.kb] : -2;
                ell2 = fma(b2a, c.log_df[lq.ka], b2b * c.log_df[lq.kb]);
            }
            const int pair1 = (cls1a & 0xffff) | (cls1b << 16);
            const int pair_e = (cls2a & 0xffff) | (cls2b << 16);
            const int pair_p = dpp_mov_i<kFromEven>(pair1), pair_s = dpp_mov_i<kFromOdd>(pair1);
            const double ell_p = dpp_mov_d<kFromEven>(ell1), ell_s = dpp_mov_d<kFromOdd>(ell1);
            // (a payment time on the value-time knot snaps to it with weight 1 on a knot that carries no sensitivity: the
            // record's weights refer to the DATE's knots, so they are zero then)
            const double pa_raw = dpp_mov_d<kFromEven>(b1a), pb_raw = dpp_mov_d<kFromEven>(b1b);     // (DPP outside the conditional)
            const double sa_raw = dpp_mov_d<kFromOdd>(b1a), sb_raw = dpp_mov_d<kFromOdd>(b1b);
            const double pa = lead ? sa_raw : (pair_p != kNullPair ? pa_raw : 0.0), pb = lead ? sb_raw : (pair_p != kNullPair ? pb_raw : 0.0);
            const double w_not = sl * N * w_q;
            // even lane: the ratio node; odd lane: the payment node (with the fixed coupon of the date)
            double om = 0.0;
            if (!odd) {
                om = (cin && accr && !lead) ? w_not * exp(ell_s - ell2 + ell_p) : 0.0;
            } else if (!lead) {
                double a_q = cin ? w_not * (spread * al_q - (accr ? 1.0 : 0.0)) : 0.0;
                if (have && q < n_fix && xtp_q == tp_q && xtp_q > 0.0) a_q = fma(sf, xpay_q, a_q);
                om = a_q * exp(ell_p);
            }
            pv += om;
            {   // a fixed coupon that found no float payment date of its own gets a pass below
                const bool own = !odd && !lead && live && q < n_fix && q < kRowSlots && !(have && xtp_q == tp_q) && xtp_q > 0.0 && sf * xpay_q != 0.0;
                const unsigned long long ob = __ballot(own);
                if (chunk == 0) fixed_b[0] = ob; else if (chunk == 1) fixed_b[1] = ob; else fixed_b[2] = ob;
            }
            // the date of the coupon: the payment time's knots; the accrual end's when the payment time carries no
            // sensitivity (the value-time knot, where the weighted coupons of DESIGN.md section 9 are "paid")
            const bool fold_e = !lead && pair_e == pair_p, fold_e0 = !lead && pair_p == kNullPair && pair_e != kNullPair;
            const int dpair = lead ? pair_s : (fold_e0 ? pair_e : pair_p);       // (a lead pair's "date": the accrual start's knots)
            const bool e_folded = fold_e || fold_e0;
            const bool e_ok = e_folded || pair_e == kNullPair;
            // can the accrual start ride on the previous coupon's date record?  Same time as that coupon's accrual end (so
            // the same lookup), which was folded into its date
            const double prev_te = dpp_mov_d<kShr1>(dpp_mov_d<kShr1>(te_q));
            const int prev_ok = dpp_mov_i<kShr1>(dpp_mov_i<kShr1>((cin && accr && (e_folded || lead)) ? 1 : 0));
            const int prev_dpair = dpp_mov_i<kShr1>(dpp_mov_i<kShr1>(dpair));
            const bool s_null = pair_s == kNullPair;
            const bool tile_prev = l >= 2 && prev_ok != 0 && ts_q == prev_te && prev_dpair != kNullPair;
            const bool ratio_on = !odd && om != 0.0;
            const bool regular = !ratio_on || ((s_null || tile_prev) && e_ok);
            const bool irregular = ratio_on && !regular;
            const bool uses_vacc = ratio_on && regular && !s_null;
            // even lane: x = the ratio node's weights on the date's knots
            double xa = pa, xb = pb;
            if (fold_e) { xa = pa - b2a; xb = pb - b2b; }
            if (fold_e0) { xa = -b2a; xb = -b2b; }
            if (lead) { xa = 0.0; xb = 0.0; }          // (the record then hands v_P - v_D = s . (ua, ub) to the next one)
            const double om_r = irregular ? 0.0 : (odd ? 0.0 : om);
            // odd lane: the payment node's Greeks (none on the value-time knot)
            const double om_p = (odd && pair_p != kNullPair) ? om : 0.0;
            // what follows this coupon on the same date: the next coupon's accrual start
            // (DPP outside any conditional: in a conditional arm the compiler may run it with the other lanes - its sources - masked off)
            const int next_uses_raw = dpp_mov_i<kShl1>(dpp_mov_i<kShl1>(uses_vacc ? 1 : 0));
            const int next_uses = l + 2 < L ? next_uses_raw : 0;
            const bool tile_next = !odd && next_uses != 0;
            const bool has_mini = pair_has_mini(dpair) || pair_has_mini(prev_dpair);
            const bool special = uses_vacc && has_mini && prev_dpair != dpair;
            const double om_p_pair = dpp_mov_d<kFromOdd>(om_p);
            const bool active = !odd && dpair != kNullPair && (om_r != 0.0 || om_p_pair != 0.0 || tile_next);
            const unsigned long long irregular_b = __ballot(irregular);      // (bits at even lanes; nothing else of the build survives the walk)

            LAG_STAMP(1);   // chunk build: lookups, exponentials, pair logic
            // ---- date records: the even lane's half {w_r, x, classes | flags}, the odd lane's {w_p, p, -}
            __builtin_amdgcn_wave_barrier();
            {
                double2* wp = reinterpret_cast<double2*>(rec + lane * 4);
                const int flags = (tile_next ? kTileNext : 0) | (special ? kSpecial : 0);
                if (!odd) {
                    wp[0] = make_double2(active ? om_r : 0.0, xa);
                    wp[1] = make_double2(xb, __hiloint2double(flags, dpair));
                } else {
                    wp[0] = make_double2(om_p, pa);
                    wp[1] = make_double2(pb, 0.0);
                }
            }
            wave_lds_sync();
            unsigned long long rows = __ballot(active);
            rows |= rows >> 32;
            rows &= kGroupMask;
            if (rows) {
                __builtin_amdgcn_s_setprio(ADR_WALK_PRIO);
                int n = __builtin_ctzll(rows);
                rows &= rows - 1;
                const double2* rec_g = reinterpret_cast<const double2*>(rec + gbase * 4);
                double2 a0 = rec_g[2 * n], a1 = rec_g[2 * n + 1];
                while (true) {
                    const bool has_next = rows != 0;
                    const int n_next = has_next ? __builtin_ctzll(rows) : n;
                    rows &= rows - 1;
                    const double w_r = a0.x, x_a = a0.y, x_b = a1.x;
                    const int word = __double2loint(a1.y), flags = __double2hiint(a1.y);
                    const int ca = static_cast<int16_t>(word & 0xffff), cb = word >> 16;
                    const double2 b0 = rec_g[2 * n + 2], b1 = rec_g[2 * n + 3];      // this record's payment half
                    a0 = rec_g[2 * n_next]; a1 = rec_g[2 * n_next + 1];               // the next record's ratio half, one record ahead
                    const double w_p = b0.x, p_a = b0.y, p_b = b1.x;
                    n = n_next;
                    __builtin_amdgcn_sched_barrier(0);
                    const int ra = ca >= 0 ? ca : zero_row, rb = cb >= 0 ? cb : zero_row;
                    double ua, ub;
                    const bool any_mini = jacobian_rows(ca, cb, ua, ub);
                    const bool t_next = (flags & kTileNext) != 0;
                    const double v_d = fma(x_b, ub, x_a * ua), v_p = fma(p_b, ub, p_a * ua);
                    const double v_r = vacc + v_d;
                    vacc = t_next ? v_p - v_d : 0.0;
                    dacc = fma(w_r, v_r, fma(w_p, v_p, dacc));
                    // first-order weights on the date's two knots: the ratio node's date part, the payment node, and the
                    // next ratio node's accrual start (= this accrual end), whose weight the next record holds
                    const double w_nx = t_next ? a0.x : 0.0;
                    const double coef_a = fma(w_r, x_a, fma(w_p, p_a, w_nx * (p_a - x_a)));
                    const double coef_b = fma(w_r, x_b, fma(w_p, p_b, w_nx * (p_b - x_b)));
                    const double coa = convexity_coef(ra, rb, coef_a, coef_b);
                    rank_one(std::true_type{}, w_r, v_r, coa, ra);
                    const bool is_special = (flags & kSpecial) != 0;
                    if (__ballot(is_special)) {
                        const unsigned mask = is_special ? (pillar_mask(ca, cb) | pillar_mask(static_cast<int16_t>(prev_word & 0xffff), prev_word >> 16)) : 0u;
                        side_update(is_special, w_r, v_r, mask);
                    }
                    prev_word = word;
                    if (any_mini) mini_convexity(ca, cb, coef_a, coef_b);
                    if (__ballot(w_p != 0.0)) rank_one(std::false_type{}, w_p, v_p, 0.0, ra);
                    if (!has_next) break;
                }
                __builtin_amdgcn_s_setprio(0);
            }
            vacc = 0.0;          // (a chunk's last record never feeds the next chunk: its first coupon is irregular)
            LAG_STAMP(2);   // date walk

            // ---- irregular ratio nodes of this chunk: three single-time parts each (accrual start +, accrual end -,
            // payment time + and fire), ten coupons per round.  Rare: the coupon's times are fetched and looked up again
            // (both lanes of the pair), so that nothing of the build has to live through the date walk
            unsigned long long todo = irregular_b;
            while (todo) {
                const unsigned mine_bits = static_cast<unsigned>((todo >> gbase) & kGroupMask);
                const int rank = __builtin_popcount(mine_bits & ((1u << (l & ~1)) - 1u));       // irregular coupons of the group before this one
                const bool mine = rank < 10 && ((todo >> (lane & ~1)) & 1ull);
                int pair_r = kNullPair, pair_e2 = kNullPair;
                double ra_w = 0.0, rb_w = 0.0, ea_w = 0.0, eb_w = 0.0, ell_r = 0.0, ell_e2 = 0.0, w_full = 0.0;
                if (mine) {
                    const int64_t at = (unit * G + g) * kRowSlots + q;
                    const double t_r = odd ? tr.row_ts[at] : tr.row_tp[at];
                    const Lookup lq = curve_lookup(c, t_r);
                    ra_w = lq.ba; rb_w = lq.bb;
                    const int ka_c = c.knot_class[lq.ka], kb_c = rb_w != 0.0 ? c.knot_class[lq.kb] : -2;
                    pair_r = (ka_c & 0xffff) | (kb_c << 16);
                    ell_r = fma(ra_w, c.log_df[lq.ka], rb_w * c.log_df[lq.kb]);
                    if (!odd) {
                        const Lookup le = curve_lookup(c, tr.row_te[at]);
                        ea_w = le.ba; eb_w = le.bb;
                        const int ea_c = c.knot_class[le.ka], eb_c = eb_w != 0.0 ? c.knot_class[le.kb] : -2;
                        pair_e2 = (ea_c & 0xffff) | (eb_c << 16);
                        ell_e2 = fma(ea_w, c.log_df[le.ka], eb_w * c.log_df[le.kb]);
                        w_full = sl * N * (tr.row_w ? tr.row_w[at] : 1.0);
                    }
                }
                const double ell_s2 = dpp_mov_d<kFromOdd>(ell_r);
                const int pair_s2 = dpp_mov_i<kFromOdd>(pair_r);
                const double w_even = (mine && !odd) ? w_full * exp(ell_s2 - ell_e2 + ell_r) : 0.0;
                const double w_r = dpp_mov_d<kFromEven>(w_even);             // the ratio node's weight, for both lanes of the pair
                __builtin_amdgcn_wave_barrier();
                if (mine) {
                    double2* base = reinterpret_cast<double2*>(rec + (gbase + 3 * rank) * 4);
                    if (odd) {
                        base[0] = make_double2(w_r, ra_w);
                        base[1] = make_double2(rb_w, __hiloint2double(0, pair_r));
                    } else {
                        base[2] = make_double2(w_r, -ea_w);
                        base[3] = make_double2(-eb_w, __hiloint2double(0, pair_e2));
                        const bool sp = (pair_has_mini(pair_s2) || pair_has_mini(pair_e2) || pair_has_mini(pair_r));
                        base[4] = make_double2(w_r, ra_w);
                        base[5] = make_double2(rb_w, __hiloint2double(kFire | (sp ? kSpecial : 0), pair_r));
                    }
                }
                wave_lds_sync();
                int cnt = min(__builtin_popcount(mine_bits), 10);
                cnt = max(cnt, __shfl_xor(cnt, 32, 64));
                const int n_rec = 3 * __builtin_amdgcn_readfirstlane(cnt);
                // a group with fewer irregular coupons walks null records for the rest
                __builtin_amdgcn_wave_barrier();
                {
                    const int own = 3 * min(__builtin_popcount(mine_bits), 10);
                    if (l >= own && l < n_rec) {
                        double2* wp = reinterpret_cast<double2*>(rec + lane * 4);
                        wp[0] = make_double2(0.0, 0.0);
                        wp[1] = make_double2(0.0, __hiloint2double((l % 3 == 2) ? kFire : 0, kNullPair));
                    }
                }
                wave_lds_sync();
                part_walk((1ull << n_rec) - 1ull);
                todo &= ~__ballot(mine && !odd);
            }

        }

        // ---------------------------------------------------------------- fixed coupons on dates of their own
        if (fixed_b[0] | fixed_b[1] | fixed_b[2]) {
            // lane = slot l of the row: which chunk / lane pair had it
            int f_chunk, f_pair;
            if (lead_row) { f_chunk = l / 15; f_pair = l % 15 + 1; }
            else if (l < 16) { f_chunk = 0; f_pair = l; }
            else { f_chunk = 1 + (l - 16) / 15; f_pair = (l - 16) % 15 + 1; }
            const unsigned long long fb = f_chunk == 0 ? fixed_b[0] : (f_chunk == 1 ? fixed_b[1] : fixed_b[2]);
            const bool own_fixed = f_chunk < 3 && ((fb >> (gbase + 2 * f_pair)) & 1ull);
            int cls_a = -2, cls_b = -2;
            double ba = 0.0, bb = 0.0, om = 0.0;
            if (own_fixed) {
                const int64_t at = (unit * G + g) * kRowSlots + l;
                const double xtp = tr.row_xtp[at], xpay = tr.row_xpay[at];
                const Lookup lq = curve_lookup(c, xtp);
                ba = lq.ba; bb = lq.bb;
                cls_a = c.knot_class[lq.ka];
                cls_b = bb != 0.0 ? c.knot_class[lq.kb] : -2;
                om = sf * xpay * exp(fma(ba, c.log_df[lq.ka], bb * c.log_df[lq.kb]));
                pv += om;
            }
            const bool greeks = own_fixed && !(cls_a == -2 && cls_b == -2);
            __builtin_amdgcn_wave_barrier();
            {
                double2* wp = reinterpret_cast<double2*>(rec + lane * 4);
                wp[0] = make_double2(greeks ? om : 0.0, greeks ? ba : 0.0);
                wp[1] = make_double2(greeks ? bb : 0.0, __hiloint2double(kFire, greeks ? ((cls_a & 0xffff) | (cls_b << 16)) : kNullPair));
            }
            wave_lds_sync();
            unsigned long long rows = __ballot(greeks);
            rows |= rows >> 32;
            rows &= kGroupMask;
            if (rows) part_walk(rows);
        }
        if (__ballot(carry_row != zero_row)) lc_row_pass(carry_row, carry_w);
        LAG_STAMP(3);   // irregular coupons, fixed coupons of their own
        if (LONG) {
            fresh = !more;
            if (more) {
                load_unit(unit + wave_stride, true);
                pin_next();
                continue;
            }
        }

        // ---------------------------------------------------------------- results
        // Everything that LOADS comes first - the output map, the running totals' read-modify-writes, the side rows'
        // table lookups, the next unit's inputs: vector memory operations retire in order, so a load issued behind this
        // unit's 16 KB of matrix stores would wait for those to drain.
#pragma unroll
        for (int off = 1; off < L; off <<= 1) pv += __shfl_xor(pv, off, 64);
        if (live && l == 0) {
            if (out.pv) out.pv[t] = pv;
        }
        int mm[8], beyond;
        load_out_map(mm, beyond);
        if (want_agg) {
            scratch_run[0] += (live && l == 0) ? pv : 0.0;
            scratch_run[64] += dacc;
#pragma unroll
            for (int i = 0; i < EPG; ++i) scratch_run[(2 + i) * 64] += acc[i] * 1e-8;      // (this lane's packed entries l + 32 i, its group's trade)
        }
        // side rows in register slots: value and whether element (m, l) has no packed entry
        double side_val[kSideSlots];
        bool side_ne[kSideSlots];
#pragma unroll
        for (int s_ = 0; s_ < kSideSlots; ++s_) {
            const int m = side_pillar[s_];                       // (wave-uniform)
            side_val[s_] = side[s_] * 1e-8;
            side_ne[s_] = false;
            if (m >= 0) {
                side_ne[s_] = live && l < P && m < P && cv.out_map[m * kPillarPad + l] < 0;
                if (want_agg) {
                    if (side_ne[s_]) scratch_tot[m * 64] += side_val[s_];
                    tot_rows |= 1u << m;
                }
            }
        }
        if (live && l < P && out.delta) __builtin_nontemporal_store(dacc * 1e-4, out.delta + static_cast<int64_t>(t) * P + l);
        int group_trade[G];
#pragma unroll
        for (int gg = 0; gg < G; ++gg) group_trade[gg] = __builtin_amdgcn_readfirstlane(__shfl(t, gg * L, 64));
        const int64_t my_gamma = static_cast<int64_t>(t) * (P * P);      // (this lane's own trade: the side-row stores)

        load_unit(unit + wave_stride, false);   // before the (large) gamma stores
        LAG_STAMP(4);   // results: loads, totals, pv / delta

        __builtin_amdgcn_s_setprio(ADR_OUT_PRIO);
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
            const int tt = group_trade[gg];
            __builtin_amdgcn_wave_barrier();
            if (g == gg) {
#pragma unroll
                for (int i = 0; i < EPG; ++i) slot[l + L * i] = acc[i] * 1e-8;
            }
            if (lane == 0) slot[kZeroEntry] = 0.0;
            wave_lds_sync();
            double* gm = (tt >= 0 ? out.gamma + static_cast<int64_t>(tt) * (P * P) : out.dump) + 2 * lane;
            double* sink = out.dump + 2 * lane;
#pragma unroll
            for (int part = 0; part < kOutParts; ++part) {
                constexpr int kBands = 8 / kOutParts;
                double gv[2 * kBands];
#pragma unroll
                for (int b = 0; b < kBands; ++b) {
                    gv[2 * b] = slot[mm[kBands * part + b] & 0xffff];
                    gv[2 * b + 1] = slot[mm[kBands * part + b] >> 16];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (STORE) {
#pragma unroll
                    for (int b = 0; b < kBands; ++b) {
                        const int band = kBands * part + b;
                        nt_pair pr; pr.x = gv[2 * b]; pr.y = gv[2 * b + 1];
                        __builtin_nontemporal_store(pr, reinterpret_cast<nt_pair*>(((beyond >> band) & 1 ? sink : gm) + band * 128));
                    }
                }
            }
        }
        // ---- side rows of this unit's special nodes: elements (m, q) and (q, m) without a packed entry take the side row
        // (stored after the matrices: same wave, same addresses, issue order)
#pragma unroll
        for (int s_ = 0; s_ < kSideSlots; ++s_) {
            const int m = side_pillar[s_];
            if (m >= 0) {
                if (STORE && out.gamma && side_ne[s_]) {
                    out.gamma[my_gamma + m * P + l] = side_val[s_];
                    out.gamma[my_gamma + l * P + m] = side_val[s_];
                }
            }
            side[s_] = 0.0; side_pillar[s_] = -1;
        }
        // rows beyond the register slots (rare: more than four distinct short-end pillars in a unit's special nodes): from
        // the wave's scratch, behind the stores
        while (ovf_rows) {
            const int m = __builtin_ctz(ovf_rows);
            ovf_rows &= ovf_rows - 1;
            const double val = scratch_ovf[m * 64] * 1e-8;
            scratch_ovf[m * 64] = 0.0;
            const bool no_entry = live && l < P && m < P && cv.out_map[m * kPillarPad + l] < 0;
            if (no_entry) {
                if (STORE && out.gamma) {
                    out.gamma[my_gamma + m * P + l] = val;
                    out.gamma[my_gamma + l * P + m] = val;
                }
                if (want_agg) scratch_tot[m * 64] += val;
            }
            if (want_agg) tot_rows |= 1u << m;
        }
        __builtin_amdgcn_s_setprio(0);
        LAG_STAMP(5);   // results: expansion and stores
        pin_next();
    }
#ifdef ADR_STAMPS
    if (out.stamps && lane == 0) {
        unsigned long long* dst = out.stamps + (static_cast<size_t>(blockIdx.x) * kWaves + wave) * 8;
        for (int i = 0; i < 8; ++i) dst[i] = stamp_sum[i];
    }
#endif

    // ------------------------------------------------------------------------ block partial of the aggregate
    if (out.block_partials) {
        double blk_gamma[kGammaPerLane];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < EPG; ++i) {
            double v = scratch_run[(2 + i) * 64];
            v += __shfl_xor(v, 32, 64);                       // both groups' trades
            if (g == 0) slot[l + L * i] = v;
        }
        wave_lds_sync();
#pragma unroll
        for (int e = 0; e < kGammaPerLane; ++e) {
            const int m = cv.out_map[(4 * bi + (e >> 2)) * kPillarPad + 4 * bj + (e & 3)];
            blk_gamma[e] = m >= 0 ? slot[m] : 0.0;
        }
        double tot_pv = scratch_run[0], tot_delta = scratch_run[64];
        tot_pv += __shfl_xor(tot_pv, 32, 64);
        tot_delta += __shfl_xor(tot_delta, 32, 64);
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem_raw);
        double* mine = red + wave * kAggStride;
        if (lane == 0) mine[0] = tot_pv;
        if (g == 0) mine[1 + l] = tot_delta * 1e-4;
#pragma unroll
        for (int e = 0; e < kGammaPerLane; ++e) {
            const int r = 4 * bi + (e >> 2), q = 4 * bj + (e & 3);
            mine[1 + kPillarPad + r * kPillarPad + q] = blk_gamma[e];
        }
        __syncthreads();
        // the side rows' totals: row m of the wave's scratch holds, per lane (group, pillar q), the sum over the lane's
        // trades of the elements (m, q) without a packed entry; both groups and the mirror element.  A pair of two
        // short-end pillars appears in both pillars' rows (equal values): the row of the smaller pillar adds it.
        {
            unsigned rowsm = tot_rows;
            while (rowsm) {
                const int m = __builtin_ctz(rowsm);
                rowsm &= rowsm - 1;
                double val = scratch_tot[m * 64];
                val += __shfl_xor(val, 32, 64);
                const bool twice = short_end_lane && l < m;      // (q, m) is also in row q's total: that row adds both mirrors
                if (g == 0 && l < P && m < P && !twice && val != 0.0) {
                    mine[1 + kPillarPad + m * kPillarPad + l] += val;
                    mine[1 + kPillarPad + l * kPillarPad + m] += val;
                }
            }
        }
        __syncthreads();
        double* dst = out.block_partials + static_cast<size_t>(blockIdx.x) * kAggStride;
        for (int i = threadIdx.x; i < kAggStride; i += kThreads) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += red[w * kAggStride + i];
            dst[i] = s;
        }
    }
}

#undef scratch_tot
#undef scratch_ovf
#undef scratch_run

using KernelFn = void (*)(CurveDev, TradesDev, OutputsDev);

template <bool STORE, bool LONG>
KernelFn lag_kernel_for(int epg) {
    switch (epg) {
        case 7: return &price_lag_kernel<STORE, LONG, 7, 5>;
        case 8: return &price_lag_kernel<STORE, LONG, 8, 6>;
        case 12: return &price_lag_kernel<STORE, LONG, 12, 10>;
        default: return &price_lag_kernel<STORE, LONG, 18, 16>;
    }
}

}  // namespace

bool lag_kernel_takes(const CurveDev& cv) {
    return cv.packed_ok && cv.hub && cv.cpg == cv.epg - 2 && cv.method != 2 && cv.P % 2 == 0;
}

int lag_kernel_threads() { return kThreads; }

size_t lag_kernel_scratch_bytes(int n_blocks) {
    return sizeof(double) * static_cast<size_t>(n_blocks) * kWaves * kScratchPerWave;
}

size_t lag_kernel_lds_bytes(const CurveDev& cv) {
    const size_t slot = lag_slot_doubles(cv.epg);
    const size_t slack = kGroupLanes * cv.cpg > cv.Ec + 1 ? kGroupLanes * cv.cpg - (cv.Ec + 1) : 0;
    size_t doubles = static_cast<size_t>(cv.K) + 2 * cv.Kc + static_cast<size_t>(cv.Kcore + 1) * cv.pc_pad +
                     static_cast<size_t>(cv.Kcore + 1) * (cv.Ec + 1) + slack;
    doubles += doubles & 1;
    doubles += kWaves * slot;
    const size_t tables = sizeof(MiniKnot) * cv.n_mini + sizeof(double) * doubles +
                          sizeof(int16_t) * (2 * static_cast<size_t>(cv.K) + cv.Kc + 2 * static_cast<size_t>(cv.n_lut));
    const size_t reduce = sizeof(double) * kWaves * kAggStride;
    const size_t need = tables > reduce ? tables : reduce;
    return (need + 15) & ~static_cast<size_t>(15);
}

hipError_t set_lag_kernel_lds_limit(size_t bytes) {
    const KernelFn fns[] = {lag_kernel_for<true, false>(7), lag_kernel_for<true, false>(8), lag_kernel_for<true, false>(12), lag_kernel_for<true, false>(18),
                            lag_kernel_for<false, false>(7), lag_kernel_for<false, false>(8), lag_kernel_for<false, false>(12), lag_kernel_for<false, false>(18),
                            lag_kernel_for<true, true>(7), lag_kernel_for<true, true>(8), lag_kernel_for<true, true>(12), lag_kernel_for<true, true>(18),
                            lag_kernel_for<false, true>(7), lag_kernel_for<false, true>(8), lag_kernel_for<false, true>(12), lag_kernel_for<false, true>(18)};
    for (KernelFn f : fns) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_price_lag(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, int n_blocks, hipStream_t stream) {
    if (!tr.rows_lagged || !lag_kernel_takes(cv) || !out.lag_scratch) return hipErrorInvalidValue;
    const size_t lds = lag_kernel_lds_bytes(cv);
    KernelFn fn;
    if (tr.rows_chained) fn = out.gamma != nullptr ? lag_kernel_for<true, true>(cv.epg) : lag_kernel_for<false, true>(cv.epg);
    else fn = out.gamma != nullptr ? lag_kernel_for<true, false>(cv.epg) : lag_kernel_for<false, false>(cv.epg);
    hipLaunchKernelGGL(fn, dim3(n_blocks), dim3(kThreads), lds, stream, cv, tr, out);
    return hipGetLastError();
}

}  // namespace adr
