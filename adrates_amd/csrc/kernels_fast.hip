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
// knots that depend on at most two par rates (curve_tables.cpp, build_packed_layout).
//
// Mapping.  OIS trades are short (15.5 coupons on average), so a 64-lane wavefront prices G = 2 trades at a
// time, L = 32 lanes each.  The host lays the eligible trades out as a table of rows sorted by coupon count
// (one row = one trade = 32 padded cash-flow slots per array), so the two trades of a wavefront have similar
// lengths, every input load is a full-width coalesced row read whose address depends only on the row number,
// and the next rows can be requested before the current results are stored.  Inside a group: lanes = coupons
// while nodes are built (coalesced loads, binary search in LDS, exp); every lane then leaves its node
// (omega, the two weights, the two knot classes) as a 32-byte record in the wave's LDS slot and the groups
// walk their nodes in lockstep: node n is two broadcast b128 reads issued one node ahead, lane l builds v
// for pillar l, v goes to the group's LDS buffer, and every lane updates its packed gamma entries l + L*i -
// the rank-1 term omega*v[p]*v[q] from LDS reads of v (hub layout: the core entries of a lane share p, read
// once), the convexity term from the left knot's LC row (the right knot's weight is carried to the next node,
// whose left knot it usually is).
// The short-end knots' one to three convexity numbers are added by the lanes that own those entries.  Each
// trade's packed ladder is then expanded through the wave's LDS slot to the symmetric P x P matrix and written
// as 1 KB-contiguous stores.  No atomics; the aggregate is a fixed-order reduction.  Legs of 33-128 coupons are
// walked as chains of rows (LONG variants).
//
// The kernel is bound by the CU's LDS pipe (about 20 LDS instructions per node pair) at 3 waves/SIMD - the
// tables fill the 160 KB of LDS, so there is one 768-thread block per CU; see DESIGN.md section 7.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>
#include <utility>
#include <vector>

#include "curve_lookup.hpp"
#include "kernels.hpp"

// This file is compiled twice: as is, and from kernels_fast_lindf.hip with ADR_FAST_LINDF = 1 - the same kernels with the
// LINEAR_FWD_RATES node arithmetic (see `lindf` in the kernel) compiled in, kept out of the log-linear instantiations
// so that their register budget (3 waves per SIMD) is untouched.
#ifndef ADR_FAST_LINDF
#define ADR_FAST_LINDF 0
#endif
#ifndef ADR_FAST_LJ_PREFETCH
#define ADR_FAST_LJ_PREFETCH 0    // 1: the next node's Jacobian entries are requested with the second batch of the rank-one update
#endif
#ifndef ADR_FAST_ASM_ROWS
#define ADR_FAST_ASM_ROWS 1       // 1: the convexity rows are read with single ds_read_b64 instructions (inline assembly; see lds_read_f64)
#endif
#ifndef ADR_GAMMA_STORE_NT
#define ADR_GAMMA_STORE_NT 1      // the per-trade gamma matrices are written with non-temporal stores
#endif
#ifndef ADR_FAST_BOTH_ROWS
#define ADR_FAST_BOTH_ROWS 2      // 0: carry the right knot's convexity weight; 1: both rows per node (plain kernels); 2: the payment-lag variant too
#endif

namespace adr {

namespace {

typedef double nt_pair __attribute__((ext_vector_type(2)));   // operand type of the non-temporal gamma stores

constexpr int kBlockThreads = kFastThreads;
#ifndef ADR_OUT_PARTS
#define ADR_OUT_PARTS 2
#endif
constexpr int kOutParts = ADR_OUT_PARTS;      // the 8 output bands of a trade are gathered in this many batches
[[maybe_unused]] constexpr int kWavesPerBlock = kBlockThreads / 64;
#ifndef ADR_LAG_THREADS
#define ADR_LAG_THREADS 512
#endif
constexpr int kLagThreads = ADR_LAG_THREADS;  // block size of the payment-lag variant

__device__ __forceinline__ double shfl_d(double x, int src) { return __shfl(x, src, 64); }

// Neighbour lanes through DPP (VALU, not the LDS pipe): lane i receives lane i + 1 / i - 1 of the wavefront;
// the last / first lane keeps its own value.
__device__ __forceinline__ double from_next_lane(double x) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), 0x130, 0xf, 0xf, false);   // wave_shl:1
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int from_prev_lane_i(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); }   // wave_shr:1
__device__ __forceinline__ double from_prev_lane(double x) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), 0x138, 0xf, 0xf, false);   // wave_shr:1
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Broadcast of one lane of every quad to the quad (DPP quad_perm: 0x00, 0x55, 0xAA, 0xFF = lane 0, 1, 2, 3).
template <int CTRL>
__device__ __forceinline__ int quad_bcast_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ double quad_bcast_d(double x) {
    return __hiloint2double(quad_bcast_i<CTRL>(__double2hiint(x)), quad_bcast_i<CTRL>(__double2loint(x)));
}

// Same-wave LDS hand-off.  The DS instructions of one wavefront are executed in issue order, so data written
// by one lane is visible to a later read of another lane of the same wave without waiting for the write to
// retire; all that is needed is that the compiler keeps the program order of the accesses (it does for
// may-aliasing LDS accesses; the empty asm is a belt-and-braces compiler barrier with no instructions).
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}); the index is a constant
// expression inside f (an immediate operand of inline assembly).
template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// One 8-byte LDS read as its own instruction.  Two reads off one base register that the compiler can see are merged into
// ds_read2_b64, which the LDS serves at HALF the rate of two ds_read_b64 (8 against 2 x 2 LDS cycles per wave-instruction on
// gfx950, MI355X guide, LDS table); the convexity rows are ten such reads per node.  The compiler does not count this read
// in its s_waitcnt bookkeeping: its own waits stay correct (LDS operations return in issue order, so a count computed
// without this read only waits for more), and the value must be passed through lds_reads_done() before its first use.
template <int BYTE_OFFSET>
__device__ __forceinline__ double lds_read_f64(unsigned addr) {
    double x;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(BYTE_OFFSET));
    return x;
}
__device__ __forceinline__ void lds_reads_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_read_done(double& x) { asm volatile("" : "+v"(x)); }   // (orders the use behind the wait)

#ifdef ADR_STAMPS
#define ADR_STAMP(slot_) do { const unsigned long long now_ = clock64(); stamp_sum[slot_] += now_ - stamp_t; stamp_t = now_; } while (0)
#else
#define ADR_STAMP(slot_) do {} while (0)
#endif

struct CurveLds {
    const double* x;            // [K]
    const double* log_df;       // [Kc]
    const double* inv_x;        // [Kc]
    const double* ljc;          // [Kcore + 1][pc_pad]
    const double* lcc;          // [Kcore + 1][Ec + 1]
    const MiniKnot* mini;       // [n_mini]
    const int16_t* lut;         // [n_lut][2]
    int n_lut;
    const int16_t* first_of;    // [K]
    const int16_t* compact_of;  // [K]
    const int16_t* knot_class;  // [Kc]
    int K, method, pc_pad, ec_stride;
};

// Doubles of LDS per wavefront (see the carve-up in the kernel).
__host__ __device__ constexpr int slot_doubles(bool gamma, int epg, int groups) {
    const int epl = (epg * (64 / groups) + 63) / 64;   // 64-entry slices of the packed ladder
    const int hand_off = 64 * 4 + (gamma ? groups * kPillarPad : 0);
    const int staging = gamma ? 64 * epl + 2 : 0;      // + the zero entry structural zeros are read from
    return hand_off > staging ? hand_off : staging;
}

// DELTA/GAMMA: what to compute; STORE: per-trade gamma matrices are written; EPG_: packed entries per group
// lane (entry l + L*i, i < EPG_) and CPG_: how many of those slots hold core pairs and read a convexity slice
// (both curve dependent; compile-time so that the loops below have no branches); G: trades per wavefront.
// CPG_ == EPG_ is the universal variant: it reads a convexity slice for every slot and zeroes the coefficient of
// the slots that are not core pairs at run time.
// LONG: the table holds trades with more than 32 coupons per leg as chains of rows (32 coupons each) that one
// group walks in consecutive iterations; a row whose meta word has bit 18 set is followed by another row of
// the same trade, and the results are written after the last one.  (Cutting a leg into rows only forgoes the
// merge of a start node into the previous payment node at the cut - same nodes, same sums.)
//
// LAG: the table holds trades whose coupons accrue to a date other than their payment date (payment lag) and / or
// carry a per-coupon notional multiplier (`row_te`, `row_w`).  Such a coupon is N w (D(ts) / D(te) - 1 + spread a)
// D(tp): a RATIO node N w D(ts) D(tp) / D(te) - one exponential, up to six knots - plus the payment node
// -N w (1 - spread a) D(tp).  Four lanes work on a coupon (eight coupons per pass): lanes 0-2 of a quad look up ts,
// te and tp, lane 3 looks up tp again for the payment node; the quad shares its log discount factors through DPP
// and every lane leaves ONE record, as in the plain kernel.  In the walk, the records of a ratio node only add to
// `vacc` (and do their first-order and convexity work, both linear in the record) until the third one, which runs
// the rank-one update with the whole v = v_s - v_e + v_p; the payment node follows as an ordinary node.  A ratio
// node can couple a short-end start interval with a later payment interval - pairs of pillars the packed ladder
// has no entry for (it holds the pairs one knot interval can create).  Such nodes ("special": at most a few per
// trade) also leave {omega, v} in a per-wave scratch in global memory, and the output phase adds omega v_r v_c to
// the elements without a packed entry while it expands the ladder.
template <bool DELTA, bool GAMMA, bool STORE, bool LONG, bool LAG, int EPG_, int CPG_, int G>
__global__ __launch_bounds__(LAG ? kLagThreads : kBlockThreads) void price_fast_kernel(CurveDev cv, TradesDev tr, OutputsDev out) {
    // the payment-lag variant keeps more state per lane (the patched elements' totals, the node under construction):
    // 512-thread blocks, i.e. two waves per SIMD and up to 256 VGPRs, instead of 768 / three / 168
    constexpr int kThreads = LAG ? kLagThreads : kBlockThreads;
    constexpr int kWaves = kThreads / 64;
    static_assert(!LAG || (GAMMA && DELTA), "the payment-lag variant exists for gamma requests");
    constexpr int L = 64 / G;                          // lanes per trade
    constexpr int PPL = kPillarPad / L;                // pillars per lane: l + L*k
    constexpr int EPG = GAMMA ? EPG_ : 1;              // packed entries per lane: l + L*i
    constexpr int CPG = GAMMA ? CPG_ : 0;              // of which core pairs (convexity rows are read for these)
    constexpr int EPL = GAMMA ? (EPG * L + 63) / 64 : 1;   // 64-entry slices of the packed ladder (staging, totals)
    static_assert(kPillarPad % L == 0 && PPL >= 1, "a group must cover the pillars evenly");
    static_assert(L == kRowSlots, "one lane per cash-flow slot of a row");
    constexpr unsigned long long kGroupMask = (L == 64) ? ~0ull : ((1ull << L) - 1);

    extern __shared__ __align__(16) unsigned char smem_raw[];
    // LDS carve-up: 64-byte records, doubles, then the int16 tables
    const int ec_stride = cv.Ec + 1;
    const int n_ljc = (cv.Kcore + 1) * cv.pc_pad;            // + the all-zero row
    const int n_lcc = GAMMA ? (cv.Kcore + 1) * ec_stride : 0;
    // zeros behind the last row: a row has Ec + 1 entries and is read L*CPG entries wide
    const int n_slack = GAMMA ? (L * CPG > cv.Ec + 1 ? L * CPG - (cv.Ec + 1) : 0) : 0;
    // per-wave slot: the lanes' 32-byte node records {omega, ba, bb, packed knot classes}, then (gamma) the
    // groups' v hand-off buffers; the front is reused as the packed-ladder staging area at output time
    constexpr int kRecDoubles = 64 * 4;
    constexpr int kSlotDoubles = slot_doubles(GAMMA, EPG, G);
    MiniKnot* s_mini = reinterpret_cast<MiniKnot*>(smem_raw);
    double* s_x = reinterpret_cast<double*>(s_mini + cv.n_mini);
    double* s_log = s_x + cv.K;
    double* s_invx = s_log + cv.Kc;
    double* s_ljc = s_invx + cv.Kc;
    double* s_lcc = s_ljc + n_ljc;
    const int n_front = cv.K + 2 * cv.Kc + n_ljc + n_lcc + n_slack;
    double* s_slot = s_x + n_front + (n_front & 1);          // 16-byte aligned (the records are read as b128)
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
    const int gbase = g * L;                      // first lane of this lane's group
    double* slot = s_slot + wave * kSlotDoubles;
    double* rec = slot;                           // node records, 4 doubles per lane
    double* vbuf = slot + kRecDoubles + g * kPillarPad;       // this group's v, 32 doubles
    const int P = cv.P;
    // LINEAR_FWD_RATES interpolates the discount factor itself, D(t) = (1 - w) d_a + w d_b: a node is TWO single-knot
    // exponentials.  Its record carries the two amounts {1, omega_a, omega_b, classes}; the first-order sum reads it as any
    // other record (v = omega_a u_a + omega_b u_b), and in the second-order sum the amount of a knot multiplies that
    // knot's u u^T + LC - the same coefficient the convexity row of the knot gets, so the rank-one update runs on the
    // knot's Jacobian row with the (carried) convexity coefficient instead of on v
    constexpr bool lindf = ADR_FAST_LINDF != 0;
    const int bi = lane >> 3, bj = lane & 7;
    const int zero_row = cv.Kcore;
    const int core_entries = GAMMA ? (cv.Ec + L - 1) / L : 0;   // slots i < this hold core pairs

    // per-lane constants
    int col[PPL];                                 // ljc column of pillar l + L*k
#pragma unroll
    for (int k = 0; k < PPL; ++k) col[k] = cv.pillar_to_core[l + L * k];
    // Exact variants (CPG < EPG) use the hub layout: the CPG core pairs of a lane share their first pillar,
    // whose v is read once per node; their convexity values sit at per-lane positions of the compact row.
    constexpr bool HUB = GAMMA && CPG < EPG;
    // BOTH_ROWS: a node's two convexity rows are read in the rank-one update's batches.  The alternative - the right knot's
    // weight carried to the next node, whose left knot it is on a grid without duplicate knots - never matches on the
    // reference's grids off the knots (a time is bracketed by the LAST knot of a run on the left and the FIRST of a run on
    // the right: different knots of one date), so every node paid a separate row pass for the flush.
    constexpr bool BOTH_ROWS = GAMMA && (!LAG || ADR_FAST_BOTH_ROWS > 1) && ADR_FAST_LINDF == 0 && ADR_FAST_BOTH_ROWS != 0;
    static_assert(!HUB || PPL == 1, "exact variants: group lane l holds pillar l");
    int up[EPG], vq[EPG];                         // the two pillars of packed entry l + L*i (0, 0 if none)
#pragma unroll
    for (int i = 0; i < EPG; ++i) {
        const int e = l + L * i;
        const bool on = GAMMA && e < cv.Eu;
        up[i] = on ? cv.ent_pq[2 * e] : 0;
        vq[i] = on ? cv.ent_pq[2 * e + 1] : 0;
    }
    const int hub_p = HUB ? up[0] : 0;            // == up[i] for every core slot i < CPG

    // Output map of this lane: it writes elements 2*lane, 2*lane + 1 of each 128-element band of a trade's
    // flat [P][P] matrix (4 rows when P = 32), so every store instruction covers 1 KB of consecutive addresses;
    // two packed entry indices per band; structural zeros point at a staging entry that holds 0.0, pairs beyond
    // P*P (P is even, so a pair never straddles the end) are flagged and stored to the sink.
    constexpr int kZeroEntry = 64 * EPL;          // staging index that always holds 0.0
    int mm[GAMMA ? 8 : 1];
    int beyond = 0;                               // bit b: band b's pair lies beyond the P x P matrix
    if (GAMMA) {
#pragma unroll
        for (int band = 0; band < 8; ++band) {
            const int raw = *reinterpret_cast<const int*>(cv.store_map + 2 * lane + band * 128);
            const int m0 = static_cast<int16_t>(raw & 0xffff), m1 = raw >> 16;
            if (m0 == -2) beyond |= 1 << band;
            mm[band] = (m0 < 0 ? kZeroEntry : m0) | ((m1 < 0 ? kZeroEntry : m1) << 16);
        }
    }

    // LAG: row and first column of this lane's pair of elements in each band (flat index 128 band + 2 lane of the
    // [P][P] matrix), and the running totals of what the output phase adds to elements without a packed entry
    int band_rc[LAG ? 8 : 1];
    double tot_patch[LAG ? 16 : 1];
    if (LAG) {
#pragma unroll
        for (int band = 0; band < 8; ++band) {
            const int flat = band * 128 + 2 * lane;
            band_rc[band] = (flat / P) | ((flat % P) << 8);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) tot_patch[i] = 0.0;
    }
    double* lag_scratch = LAG ? out.lag_scratch + (static_cast<size_t>(blockIdx.x) * kWaves + wave) * (G * kLagScratchNodes * kLagStashDoubles) +
                                    g * (kLagScratchNodes * kLagStashDoubles) : nullptr;

    // running portfolio sums of this wave: pv per group (lane l == 0), delta per group lane/pillar,
    // gamma in the 64-lane packed layout (entry lane + 64 s)
    double tot_pv = 0.0, tot_delta[PPL], tot_gamma[GAMMA ? EPL : 1];
#pragma unroll
    for (int k = 0; k < PPL; ++k) tot_delta[k] = 0.0;
#pragma unroll
    for (int s = 0; s < (GAMMA ? EPL : 1); ++s) tot_gamma[s] = 0.0;

#ifdef ADR_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = clock64();
#endif
    // ---------------------------------------------------------------------------------------------------
    // Main loop.  A unit is G consecutive rows of the sorted, padded trade table (one row = one trade, 32
    // cash-flow slots); group g of the wave takes row G*unit + g.  All loads of a unit depend only on the
    // unit number, and the loads of the NEXT unit are issued before this unit's result stores: vector
    // memory operations of a wave retire in order, so a load issued behind 16 KB of stores would also wait
    // for those stores.
    const int64_t n_units = (tr.n_rows + G - 1) / G;
    const int64_t wave_stride = static_cast<int64_t>(gridDim.x) * kWaves;
    int64_t unit = static_cast<int64_t>(blockIdx.x) * kWaves + wave;

    double nx_tp = 0.0, nx_ts = 0.0, nx_al = 0.0, nx_xtp = 0.0, nx_xpay = 0.0, nx_N = 0.0, nx_spread = 0.0;
    int nx_meta = 0, nx_trade = -1;
    auto load_unit = [&](int64_t u) {
        const int64_t row = u * G + g;
        nx_tp = nx_ts = nx_al = nx_xtp = nx_xpay = nx_N = nx_spread = 0.0;
        nx_meta = 0; nx_trade = -1;
        if (u < n_units && row < tr.n_rows) {
            const int64_t at = row * kRowSlots + l;
            // read-once input stream and write-once outputs: non-temporal, they should not displace anything in L2
            nx_tp = __builtin_nontemporal_load(tr.row_tp + at);
            if (!LAG) { nx_ts = __builtin_nontemporal_load(tr.row_ts + at); nx_al = __builtin_nontemporal_load(tr.row_alpha + at); }
            nx_xtp = __builtin_nontemporal_load(tr.row_xtp + at); nx_xpay = __builtin_nontemporal_load(tr.row_xpay + at);
            nx_N = tr.row_notional[row]; nx_spread = tr.row_spread[row];
            nx_meta = tr.row_meta[row]; nx_trade = tr.row_trade[row];
        }
    };
    // The wait for a unit's inputs is placed by hand at the END of the previous iteration, where exactly the
    // unit's 16 result stores are younger than the loads (vector memory operations retire in order, so the
    // wait is "all but the last 16").  Left to the top of the loop, the wait would have to be correct for the
    // entry edge as well and would drain the stores too.
    auto pin_next = [&]() {
        asm volatile("" : "+v"(nx_tp), "+v"(nx_ts), "+v"(nx_al), "+v"(nx_xtp), "+v"(nx_xpay), "+v"(nx_N),
                          "+v"(nx_spread), "+v"(nx_meta), "+v"(nx_trade));
    };
    load_unit(unit);
    pin_next();

    // LONG: the accumulators live across the iterations of a chain; otherwise they are per iteration
    double pv_chain = 0.0, dacc_chain[PPL], acc_chain[EPG];
    bool fresh = true;                         // LONG: false while a trade's chain of rows is being walked
    int n_special = 0;                         // LAG: special nodes of this group's trade so far (over the whole chain)
    for (; unit < n_units; unit += wave_stride) {
        double pv_unit = 0.0, dacc_unit[PPL], acc_unit[EPG];
        double& pv = LONG ? pv_chain : pv_unit;
        double (&dacc)[PPL] = LONG ? dacc_chain : dacc_unit;
        double (&acc)[EPG] = LONG ? acc_chain : acc_unit;
        // ------------------------------------------------------------------ this group's trade
        const double tp = nx_tp, ts = nx_ts, al = nx_al, xtp = nx_xtp, xpay = nx_xpay;
        const double N = nx_N, spread = nx_spread;
        const int t = nx_trade;
        const bool live = t >= 0;
        const int n_flt = nx_meta & 0xff, n_fix = (nx_meta >> 8) & 0xff;
        const double sl = (nx_meta & 0x10000) ? -1.0 : 1.0, sf = (nx_meta & 0x20000) ? -1.0 : 1.0;
        // both groups' chains have the same length (the host pads the shorter one with empty rows)
        const bool more = LONG && (__builtin_amdgcn_readfirstlane(nx_meta) & 0x40000) != 0;
        ADR_STAMP(0);   // waiting for the unit's inputs
        if (!GAMMA) load_unit(unit + wave_stride);   // small kernels have the registers to fetch a whole unit ahead

        if (!LONG || fresh) {
            pv = 0.0;
#pragma unroll
            for (int k = 0; k < PPL; ++k) dacc[k] = 0.0;
#pragma unroll
            for (int i = 0; i < EPG; ++i) acc[i] = 0.0;
        }

        // ---- fold the coupons into nodes (lane l = coupon l of the group's trade)
        const bool in = live && l < n_flt;
        const double ntp = from_next_lane(tp), nts = from_next_lane(ts), nal = from_next_lane(al);
        const double ptp = from_prev_lane(tp);
        const bool valid = in && tp >= 0.0;
        const bool accrues = al > 0.0;            // te == tp for every coupon of a fast-path trade
        // payment node P_l: -N(1 - spread*a) D(tp)   (N*spread*a*D(tp) when nothing accrues)
        double a_pay = valid ? sl * N * (spread * al - (accrues ? 1.0 : 0.0)) : 0.0;
        // the next coupon's start node lands here when its accrual starts on this payment time
        if (in && l + 1 < n_flt && nal > 0.0 && ntp >= 0.0 && nts == tp) a_pay += sl * N;
        // the fixed coupon paid at the same time joins the node
        const bool fix_in = live && l < n_fix;
        const bool fix_merged = fix_in && in && xtp == tp;
        if (fix_merged && xtp > 0.0) a_pay = fma(sf, xpay, a_pay);
        // own start node S_l unless it coincides with the previous payment node
        bool own_start = !LAG && valid && accrues && !(l > 0 && ptp == ts);
        const bool own_fixed = fix_in && !fix_merged && xtp > 0.0 && sf * xpay != 0.0;

        double qt = tp, qa = a_pay;               // this lane's query: time and coefficient
        bool qon = in && a_pay != 0.0;
        if (!LAG) {   // the group's first start node moves to the group's first spare lane, if there is one
            const unsigned long long mine = (__ballot(own_start) >> gbase) & kGroupMask;
            const bool move = mine != 0 && n_flt < L;
            const int src = gbase + (mine ? __builtin_ctzll(mine) : 0);
            const double st = shfl_d(ts, src);
            if (move && l == n_flt) { qt = st; qa = sl * N; qon = true; }
            if (move && lane == src) own_start = false;
        }
        // LAG: the float coupons are walked eight at a time, four lanes per coupon (chunk passes), then the fixed
        // coupons that did not merge (lane = coupon again)
        int n_chunks = 0;
        if (LAG) {
            int m = live ? n_flt : 0;
#pragma unroll
            for (int off = L; off < 64; off <<= 1) m = max(m, __shfl_xor(m, off, 64));
            n_chunks = (__builtin_amdgcn_readfirstlane(m) + 7) >> 3;
        }
        double vacc[PPL];                         // LAG: v of the ratio node under construction
#pragma unroll
        for (int k = 0; k < PPL; ++k) vacc[k] = 0.0;
        if (!LONG || fresh) n_special = 0;
        const bool more_starts = __ballot(own_start) != 0, more_fixed = __ballot(own_fixed) != 0;
#if ADR_BUILD_PRIO
        if (GAMMA) __builtin_amdgcn_s_setprio(ADR_BUILD_PRIO);
#endif
        ADR_STAMP(1);   // node folding

        // convexity row whose weight is still to be added (see the consume loop), per group
        int carry_row = zero_row;
        double carry_w = 0.0;
        // (Carrying the right knot's Jacobian entries to the next node as well - one LDS read less per node - was
        // measured 7 % SLOWER: the select and its branch sit on the node's critical path.  Requesting the NEXT node's
        // Jacobian entries inside the current node's rank-one update, so that the record -> rows round trip leaves the
        // head of a node's chain: 1-4 % slower, at the same 167 registers.  Reading the right knot's convexity row in the
        // rank-one update's batches as well, instead of carrying it: 5 % slower, 12 bytes of scratch.)
        auto lc_row_pass = [&](int row, double w) {
            double lr[CPG > 0 ? CPG : 1];
            {   // entry l + L*i of a row sits at position l + L*i (the hub layout too: curve_tables.cpp, hub_layout)
                const double* src = c.lcc + __mul24(row, c.ec_stride) + l;
#if ADR_FAST_ASM_ROWS
                // single ds_read_b64 instructions (see lds_read_f64: a merged ds_read2_b64 runs at half rate)
                const unsigned addr = static_cast<unsigned>(reinterpret_cast<size_t>(src));
                static_for<(CPG > 0 ? CPG : 1)>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    if constexpr (i < CPG) lr[i] = lds_read_f64<8 * L * i>(addr);
                });
                lds_reads_wait();
                static_for<(CPG > 0 ? CPG : 1)>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    if constexpr (i < CPG) lds_read_done(lr[i]);
                });
#else
#pragma unroll
                for (int i = 0; i < CPG; ++i) lr[i] = src[L * i];
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CPG; ++i) {
                const bool core = CPG < EPG || i < core_entries;
                acc[i] = fma(core ? w : 0.0, lr[i], acc[i]);
            }
        };

        const int n_pass = LAG ? n_chunks + 1 : 3;
        for (int pass = 0; pass < n_pass; ++pass) {
            const bool chunk_pass = LAG && pass < n_chunks;
            if (!LAG) {
                if (pass == 1) {            // fixed coupons that did not merge into a float payment node
                    if (!more_fixed) continue;
                    qt = xtp; qa = sf * xpay; qon = own_fixed;
                } else if (pass == 2) {     // start nodes that found no spare lane
                    if (!more_starts) break;
                    qt = ts; qa = sl * N; qon = own_start;
                }
            } else if (!chunk_pass) {
                if (!more_fixed) break;
                qt = xtp; qa = sf * xpay; qon = own_fixed;
            }
            // ---- build: lookup + exp in the lanes that own a query
            int cls_a = -2, cls_b = -2;
            double ba = 0.0, bb = 0.0, omega = 0.0;
            bool greeks_lag = false, special = false, taken_by_date = false;
            if (chunk_pass) {
                // coupon q of the row, role 0: accrual start, 1: accrual end, 2: payment time (ratio node),
                // 3: payment time (payment node); the quad's lanes read the same words (one L1 line per array)
                const int q = 8 * pass + (l >> 2), role = l & 3;
                const int64_t at = (unit * G + g) * kRowSlots + q;
                const bool have = live && q < n_flt;
                double ctp = 0.0, t_q = 0.0, cal = 0.0, cw = 1.0, cxtp = 0.0, cxpay = 0.0;
                if (have) {
                    // this lane's query time comes from the array of its role (one load instead of three)
                    const double* times = role == 0 ? tr.row_ts : (role == 1 ? tr.row_te : tr.row_tp);
                    t_q = times[at];
                    ctp = role >= 2 ? t_q : tr.row_tp[at];
                    cal = tr.row_alpha[at];
                    if (tr.row_w) cw = tr.row_w[at];
                    if (role == 3 && q < n_fix) { cxtp = tr.row_xtp[at]; cxpay = tr.row_xpay[at]; }
                }
                const bool cin = have && ctp >= 0.0, accr = cal > 0.0;
                const bool look = cin && (role == 3 || accr);
                int ka = 0, kb = 0;
                double ell = 0.0;
                if (look) {
                    const Lookup lq = curve_lookup(c, t_q);
                    ba = lq.ba; bb = lq.bb; ka = lq.ka; kb = lq.kb;
                    cls_a = c.knot_class[ka];
                    cls_b = bb != 0.0 ? c.knot_class[kb] : -2;
                    ell = fma(ba, c.log_df[ka], bb * c.log_df[kb]);
                }
                // the quad shares its three log discount factors, classes and the accrual-end weights
                const double ls = quad_bcast_d<0x00>(ell), le = quad_bcast_d<0x55>(ell), lp = quad_bcast_d<0xAA>(ell);
                const int pair = (cls_a << 16) | (cls_b & 0xffff);
                const int pair_s = quad_bcast_i<0x00>(pair), pair_e = quad_bcast_i<0x55>(pair), pair_p = quad_bcast_i<0xAA>(pair);
                const double e_ba = quad_bcast_d<0x55>(ba), e_bb = quad_bcast_d<0x55>(bb);
                const int null_pair = static_cast<int>(0xfffefffeu);
                const double w_not = sl * N * cw;
                const double om_r = (cin && accr) ? w_not * exp(ls - le + lp) : 0.0;
                // accrual end and payment time a few days apart share their knots: the end weights join the payment
                // record (one walk of those rows instead of two)
                const bool fold_e = pair_e == pair_p;
                // a payment time without sensitivity (the value-time knot - where the weighted coupons of a leg projected
                // on another curve are "paid"): the date record takes the accrual end's two knots, so
                // that such a coupon is one record too (and its successor's accrual start can ride along)
                const bool fold_e0 = pair_p == null_pair && pair_e != null_pair;
                const int date_pair = fold_e0 ? pair_e : pair_p;             // the knots of the quad's date record
                if (role == 1) { ba = -ba; bb = -bb; }
                if (role == 2 && fold_e) { ba -= e_ba; bb -= e_bb; }
                if (role == 2 && fold_e0) {
                    ba = -e_ba; bb = -e_bb;
                    cls_a = pair_e >> 16; cls_b = static_cast<int>(static_cast<int16_t>(pair_e & 0xffff));
                }
                if (role == 3) {
                    double a_q = cin ? w_not * (spread * cal - (accr ? 1.0 : 0.0)) : 0.0;
                    if (have && q < n_fix && cxtp == ctp && cxtp > 0.0) a_q = fma(sf, cxpay, a_q);   // the fixed coupon of the date
                    omega = a_q * exp(ell);
                    pv += omega;
                    greeks_lag = omega != 0.0 && pair != null_pair;
                } else {
                    omega = om_r;
                    if (role == 2) pv += omega;
                    const bool any_part = pair_s != null_pair || pair_e != null_pair || pair_p != null_pair;
                    greeks_lag = om_r != 0.0 && (role == 2 ? any_part : (pair != null_pair && !(role == 1 && (fold_e || fold_e0))));
                    // special: a short-end knot is involved and the parts do not all sit on one knot interval - the
                    // rank-one term then reaches pairs of pillars the packed ladder has no entry for
                    const bool has_mini = min(min(pair_s >> 16, static_cast<int>(static_cast<int16_t>(pair_s))),
                                              min(min(pair_e >> 16, static_cast<int>(static_cast<int16_t>(pair_e))),
                                                  min(pair_p >> 16, static_cast<int>(static_cast<int16_t>(pair_p))))) <= -3;
                    const bool one_interval = (pair_s == null_pair || pair_s == date_pair) && (pair_e == null_pair || pair_e == date_pair);
                    special = role == 2 && om_r != 0.0 && has_mini && !one_interval;
                }
                if (!greeks_lag) { omega = 0.0; ba = bb = 0.0; cls_a = cls_b = -2; }
                // records the date record of a quad takes along (see the walk): the quad's payment node, and the accrual
                // start of the NEXT quad when it sits on the same two knots as this quad's payment time
                const bool pr_live = quad_bcast_i<0xAA>(greeks_lag ? 1 : 0) != 0;
                const int prev_pair = from_prev_lane_i(from_prev_lane_i(date_pair)), prev_live = from_prev_lane_i(from_prev_lane_i(greeks_lag ? 1 : 0));
                taken_by_date = greeks_lag && ((role == 3 && pr_live) || (role == 0 && l >= 4 && prev_live != 0 && prev_pair == pair));
            } else
            if (qon) {
                const Lookup q = curve_lookup(c, qt);
                ba = q.ba; bb = q.bb;
                cls_a = c.knot_class[q.ka];
                cls_b = bb != 0.0 ? c.knot_class[q.kb] : -2;
                if (!LAG && lindf) {
                    ba = qa * ba * exp(c.log_df[q.ka]);
                    bb = bb != 0.0 ? qa * bb * exp(c.log_df[q.kb]) : 0.0;
                    pv += ba + bb;
                    omega = 1.0;
                } else {
                    omega = qa * exp(fma(ba, c.log_df[q.ka], bb * c.log_df[q.kb]));
                    pv += omega;
                }
            }
            ADR_STAMP(2);   // lookup + exp
            if (!DELTA) continue;
            const bool greeks = chunk_pass ? greeks_lag : (qon && !(cls_a == -2 && cls_b == -2));
            if (!greeks) { omega = 0.0; cls_a = -2; cls_b = -2; }
            // ---- consume: the groups walk their nodes in lockstep.  Every lane leaves its node as a 32-byte
            // record {omega, ba, bb, the two knot classes} in LDS; node n of a group is then two broadcast
            // b128 reads (instead of eight ds_bpermute), issued one node ahead so that the round trip hides
            // behind the previous node's work.
            __builtin_amdgcn_wave_barrier();
            {
                double2* wp = reinterpret_cast<double2*>(rec + lane * 4);
                wp[0] = make_double2(omega, ba);
                const int cls_b_word = LAG ? ((cls_b & 0xffff) | (special ? 0x10000 : 0) | (taken_by_date ? 0x20000 : 0)) : cls_b;
                wp[1] = make_double2(bb, __hiloint2double(cls_b_word, cls_a));   // both words are read back: a dead
                // half would be reused as a scratch register while the prefetch of the record is still in flight
            }
            wave_lds_sync();
            unsigned long long any_row = __ballot(greeks && !taken_by_date);
#pragma unroll
            for (int off = L; off < 64; off <<= 1) any_row |= any_row >> off;
            any_row &= kGroupMask;
            if (!any_row) continue;
#ifdef ADR_DEBUG_SKIP_WALK             // diagnostic build: nodes are built but not walked (ladders stay zero) - the output phase alone
            continue;
#endif
#if ADR_WALK_PRIO
            if (GAMMA) __builtin_amdgcn_s_setprio(ADR_WALK_PRIO);
#endif
            int n = __builtin_ctzll(any_row);
            any_row &= any_row - 1;
            const double2* rec_g = reinterpret_cast<const double2*>(rec + gbase * 4);
            double2 nx0 = rec_g[2 * n], nx1 = rec_g[2 * n + 1];
            constexpr bool LJPRE = GAMMA && !LAG && PPL == 1 && ADR_FAST_LINDF == 0 && ADR_FAST_LJ_PREFETCH != 0;
            double pre_ua = 0.0, pre_ub = 0.0;            // LJPRE: the Jacobian entries of the node about to be walked
            auto request_rows = [&]() {                   // ... requested from the record in nx0 / nx1
                const int n_ca = __double2loint(nx1.y), n_cb = __double2hiint(nx1.y);
                const int n_ra = n_ca >= 0 ? n_ca : zero_row, n_rb = n_cb >= 0 ? n_cb : zero_row;
                pre_ua = c.ljc[__mul24(n_ra, c.pc_pad) + col[0]];
                pre_ub = c.ljc[__mul24(n_rb, c.pc_pad) + col[0]];
            };
            if (LJPRE) request_rows();
            while (true) {
                const bool has_next = any_row != 0;
                const int n_next = has_next ? __builtin_ctzll(any_row) : n;   // nothing left: read n again
                any_row &= any_row - 1;
                const double om = nx0.x, wa = nx0.y, wb = nx1.x;
                const int cb_word = __double2hiint(nx1.y);
                const int ca = __double2loint(nx1.y), cb = LAG ? static_cast<int>(static_cast<int16_t>(cb_word & 0xffff)) : cb_word;
                // LAG chunk passes (the lane index is the same for both groups, so the role is wave-uniform): records 0
                // and 1 of a quad (accrual start / end) only add to the ratio node's v; record 2 is the DATE record - it
                // completes the ratio node and takes the quad's payment node (record 3) and the NEXT coupon's accrual
                // start (record 0 of the next quad) along when those sit on the same two knots (flag 0x20000 in their
                // class word): one walk of the date's rows, two rank-one updates
                const int n_cur = n;
                const int role = chunk_pass ? (n_cur & 3) : 3;
                const bool is_special = LAG && (cb_word & 0x10000) != 0;
                const bool taken = LAG && (cb_word & 0x20000) != 0;     // served by a date record: nothing to do on its own
                nx0 = rec_g[2 * n_next]; nx1 = rec_g[2 * n_next + 1];
                n = n_next;
                __builtin_amdgcn_sched_barrier(0);
                const int ra = ca >= 0 ? ca : zero_row, rb = cb >= 0 ? cb : zero_row;
                const bool mini_a = ca <= -3, mini_b = cb <= -3;
                const bool any_mini = __ballot(mini_a || mini_b) != 0;

                // LJ at this lane's pillars for the record's two knots: core rows (the zero row for anything else),
                // the short-end knots' one or two entries from their records
                double ua[PPL], ub[PPL];
                if (LJPRE) {
                    ua[0] = pre_ua; ub[0] = pre_ub;
                } else {
                    // (24-bit multiplies: full rate, v_mul_lo_u32 is quarter rate)
                    const double* lja = c.ljc + __mul24(ra, c.pc_pad);
                    const double* ljb = c.ljc + __mul24(rb, c.pc_pad);
#pragma unroll
                    for (int k = 0; k < PPL; ++k) { ua[k] = lja[col[k]]; ub[k] = ljb[col[k]]; }
                }
                if (any_mini) {
                    if (mini_a) {
                        const MiniKnot& m = c.mini[-3 - ca];
#pragma unroll
                        for (int k = 0; k < PPL; ++k) {
                            const int p = l + L * k;
                            ua[k] = p == m.p[0] ? m.lj[0] : (p == m.p[1] ? m.lj[1] : 0.0);
                        }
                    }
                    if (mini_b) {
                        const MiniKnot& m = c.mini[-3 - cb];
#pragma unroll
                        for (int k = 0; k < PPL; ++k) {
                            const int p = l + L * k;
                            ub[k] = p == m.p[0] ? m.lj[0] : (p == m.p[1] ? m.lj[1] : 0.0);
                        }
                    }
                }
                ADR_STAMP(5);   // (walk) record decode, Jacobian rows
                // convexity of short-end knots: one to three numbers (the symmetric 2x2 block on pillars p0, p1), added
                // by the lanes that own the packed entries (p0,p0), (p0,p1), (p1,p1)
                auto mini_convexity = [&](double coef_a, double coef_b) {
                    if (any_mini) {
#pragma unroll
                        for (int side = 0; side < 2; ++side) {
                            const bool mine = side == 0 ? mini_a : mini_b;
                            if (!__ballot(mine)) continue;
                            const MiniKnot& m = c.mini[mine ? (-3 - (side == 0 ? ca : cb)) : 0];
                            const double coef = mine ? (side == 0 ? coef_a : coef_b) : 0.0;
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                const int e = m.e[j];
                                if (j > 0 && !__ballot(mine && e >= 0)) continue;   // single-pillar knots: one entry
                                const bool owner = mine && e >= 0 && (e % L) == l;
                                const double add = owner ? coef * m.lc[j] : 0.0;
                                const int at = e / L;
                                if constexpr (CPG < EPG) {
                                    // exact variants: a short-end knot's pairs are fringe pairs - the last slots - unless its second
                                    // pillar is a core pillar; the core slots are visited only then
#pragma unroll
                                    for (int i = CPG; i < EPG; ++i) acc[i] += (i == at) ? add : 0.0;
                                    if (__ballot(owner && at < CPG)) {
#pragma unroll
                                        for (int i = 0; i < CPG; ++i) acc[i] += (i == at) ? add : 0.0;
                                    }
                                } else {
#pragma unroll
                                    for (int i = 0; i < EPG; ++i) acc[i] += (i == at) ? add : 0.0;
                                }
                            }
                        }
                    }
                };
                // Convexity rows.  Consecutive nodes of a swap usually share a knot (the right neighbour of one payment
                // time is the left neighbour of the next), so the right-hand row is not read with its node: its weight
                // is carried to the next node and joins that node's left-hand weight when the rows agree; a carried row
                // that does not match is added on its own first.  Returns the left row's coefficient.
                int flush_row = zero_row;        // LINEAR_FWD_RATES: a carried knot that did not match (see below)
                double flush_w = 0.0;
                auto convexity_coef = [&](double coef_a, double coef_b) {
                    if (__ballot(carry_row != zero_row && carry_row != ra)) {
                        const bool flush = carry_row != ra;
                        if (!LAG && lindf) { flush_row = flush ? carry_row : zero_row; flush_w = flush ? carry_w : 0.0; }
                        else lc_row_pass(flush ? carry_row : zero_row, flush ? carry_w : 0.0);
                        if (flush) { carry_row = zero_row; carry_w = 0.0; }
                    }
                    const double coa = coef_a + (carry_row == ra ? carry_w : 0.0);
                    carry_row = rb; carry_w = coef_b;
                    return coa;
                };
                // rank-one update om_r * vv vv^T through the group's LDS slot, with the left row's convexity term
                // (coefficient coa) folded into the same batches of LDS reads when WITH_ROW
                auto rank_one_rows = [&](auto with_row, auto two_rows, double om_r, const double (&vv_)[PPL], double coa, int row, double cob, int row_b) {
                    constexpr bool WITH_ROW = decltype(with_row)::value;
                    constexpr bool TWO_ROWS = WITH_ROW && decltype(two_rows)::value;       // both knots' convexity rows in these batches (no carry)
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < PPL; ++k) vbuf[l + L * k] = vv_[k];
                    wave_lds_sync();
#if ADR_RANK_PRIO
                    __builtin_amdgcn_s_setprio(ADR_RANK_PRIO);
#endif
                    const double* rowa = c.lcc + __mul24(row, c.ec_stride) + l;
                    const double* rowb = c.lcc + __mul24(TWO_ROWS ? row_b : row, c.ec_stride) + l;
                    // All operands of a batch of entries are fetched before any of them is used: the scheduling barrier
                    // keeps the compiler from pairing each LDS read with its FMA (which would expose one LDS round trip
                    // per entry).
                    constexpr int kBatch = EPG < ADR_FAST_BATCH ? EPG : ADR_FAST_BATCH;
                    double hub_v = 0.0;
                    constexpr bool ASM_ROWS = ADR_FAST_ASM_ROWS != 0 && WITH_ROW;
                    if constexpr (ASM_ROWS) {
                        // the rows as single ds_read_b64 (lds_read_f64), issued AHEAD of the batch's gathers: the compiler's waits
                        // for the gathers then cover the rows as well (in-order return), and the explicit wait below is free
                        const unsigned addr_a = static_cast<unsigned>(reinterpret_cast<size_t>(rowa));
                        const unsigned addr_b = static_cast<unsigned>(reinterpret_cast<size_t>(rowb));
                        constexpr int kBatches = (EPG + kBatch - 1) / kBatch;
                        static_for<kBatches>([&](auto b_) {
                            constexpr int i0 = decltype(b_)::value * kBatch;
                            double uu[kBatch], vv[kBatch], la[kBatch], lb[TWO_ROWS ? kBatch : 1];
                            static_for<kBatch>([&](auto i_) {
                                constexpr int i = decltype(i_)::value;
                                if constexpr (i0 + i < CPG) {
                                    la[i] = lds_read_f64<8 * L * (i0 + i)>(addr_a);
                                    if constexpr (TWO_ROWS) lb[i] = lds_read_f64<8 * L * (i0 + i)>(addr_b);
                                }
                            });
                            if (HUB && i0 == 0) hub_v = vbuf[hub_p];
#pragma unroll
                            for (int i = 0; i < kBatch; ++i) {
                                if (i0 + i >= EPG) continue;
                                if (!HUB) uu[i] = vbuf[up[i0 + i]];
                                vv[i] = vbuf[vq[i0 + i]];
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < kBatch; ++i) {
                                if (i0 + i >= EPG) continue;
                                const double u_i = HUB ? (i0 + i < CPG ? hub_v : vv_[0]) : uu[i];
                                acc[i0 + i] = fma(om_r * u_i, vv[i], acc[i0 + i]);
                            }
                            if constexpr (i0 < CPG) {
                                lds_reads_wait();
                                static_for<kBatch>([&](auto i_) {
                                    constexpr int i = decltype(i_)::value;
                                    if constexpr (i0 + i < CPG) {
                                        lds_read_done(la[i]);
                                        if constexpr (TWO_ROWS) lds_read_done(lb[i]);
                                        const bool core = CPG < EPG || (i0 + i) < core_entries;   // compile-time true unless universal
                                        double gsum = fma(core ? coa : 0.0, la[i], acc[i0 + i]);
                                        if constexpr (TWO_ROWS) gsum = fma(core ? cob : 0.0, lb[i], gsum);
                                        acc[i0 + i] = gsum;
                                    }
                                });
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        });
                    } else {
#pragma unroll
                    for (int i0 = 0; i0 < EPG; i0 += kBatch) {
                        double uu[kBatch], vv[kBatch], la[kBatch], lb[TWO_ROWS ? kBatch : 1];
                        if (HUB && i0 == 0) hub_v = vbuf[hub_p];
                        if (LJPRE && WITH_ROW && i0 + kBatch >= EPG) request_rows();      // (the last batch: the next record has long arrived)
#pragma unroll
                        for (int i = 0; i < kBatch; ++i) {
                            if (i0 + i >= EPG) continue;
                            // exact variants: the fringe slots' first pillar is the lane's own (curve_tables.cpp) - no gather
                            if (!HUB) uu[i] = vbuf[up[i0 + i]];
                            vv[i] = vbuf[vq[i0 + i]];
                            // convexity rows: entry l + L*i of a row sits at row[l + L*i] (hub layout: row[pos[i]])
                            if (WITH_ROW && i0 + i < CPG) la[i] = rowa[L * (i0 + i)];
                            if (TWO_ROWS && i0 + i < CPG) lb[i] = rowb[L * (i0 + i)];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < kBatch; ++i) {
                            if (i0 + i >= EPG) continue;
                            const double u_i = HUB ? (i0 + i < CPG ? hub_v : vv_[0]) : uu[i];
                            double gsum = fma(om_r * u_i, vv[i], acc[i0 + i]);
                            if (WITH_ROW && i0 + i < CPG) {
                                const bool core = CPG < EPG || (i0 + i) < core_entries;   // compile-time true unless universal
                                gsum = fma(core ? coa : 0.0, la[i], gsum);
                                if (TWO_ROWS) gsum = fma(core ? cob : 0.0, lb[i], gsum);
                            }
                            acc[i0 + i] = gsum;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    }
#if ADR_RANK_PRIO
                    __builtin_amdgcn_s_setprio(ADR_WALK_PRIO);
#endif
                };
                auto rank_one_row = [&](auto with_row, double om_r, const double (&vv_)[PPL], double coa, int row) {
                    rank_one_rows(with_row, std::false_type{}, om_r, vv_, coa, row, 0.0, 0);
                };
                auto rank_one = [&](auto with_row, double om_r, const double (&vv_)[PPL], double coa) {
                    rank_one_row(with_row, om_r, vv_, coa, ra);
                };
                // LINEAR_FWD_RATES: amount w on core row `row` - w (u u^T + LC) with u the row's Jacobian entries
                auto lindf_knot = [&](int row, double w) {
                    double u[PPL];
                    const double* lj = c.ljc + __mul24(row, c.pc_pad);
#pragma unroll
                    for (int k = 0; k < PPL; ++k) u[k] = lj[col[k]];
                    rank_one_row(std::true_type{}, w, u, w, row);
                };
                auto stash_special = [&](const double (&vv_)[PPL], double om_r) {   // {v, omega} for the pairs without a packed entry
                    if (__ballot(is_special)) {
                        if (is_special) {
                            double* dst = lag_scratch + n_special * kLagStashDoubles;
#pragma unroll
                            for (int k = 0; k < PPL; ++k) dst[l + L * k] = vv_[k];
                            if (l == 0) dst[kPillarPad] = om_r;
                            ++n_special;
                        }
                    }
                };

                if (LAG && chunk_pass && role == 2) {
                    // ---- date record: ratio node (its last part), payment node, next coupon's accrual start
                    const double2 p0 = rec_g[2 * (n_cur + 1)], p1 = rec_g[2 * (n_cur + 1) + 1];
                    const int s_at = n_cur + 2 < L ? n_cur + 2 : n_cur;              // last quad of the row: nothing follows
                    const double2 s0 = rec_g[2 * s_at], s1 = rec_g[2 * s_at + 1];
                    const bool use_p = (__double2hiint(p1.y) & 0x20000) != 0;
                    const bool use_s = n_cur + 2 < L && (__double2hiint(s1.y) & 0x20000) != 0;
                    const double om_p = use_p ? p0.x : 0.0, wpa = use_p ? p0.y : 0.0, wpb = use_p ? p1.x : 0.0;
                    const double om_s = use_s ? s0.x : 0.0, wsa = use_s ? s0.y : 0.0, wsb = use_s ? s1.x : 0.0;
                    double v1[PPL], vp[PPL];
#pragma unroll
                    for (int k = 0; k < PPL; ++k) {
                        const double vr = fma(wb, ub[k], wa * ua[k]);
                        vp[k] = fma(wpb, ub[k], wpa * ua[k]);
                        const double vs = fma(wsb, ub[k], wsa * ua[k]);
                        dacc[k] = fma(om, vr, fma(om_p, vp[k], fma(om_s, vs, dacc[k])));
                        v1[k] = vacc[k] + vr;
                        vacc[k] = vs;                                   // the next ratio node starts with its accrual start
                    }
                    stash_special(v1, om);
                    if (GAMMA) {
                        const double coef_a = fma(om, wa, fma(om_p, wpa, om_s * wsa)), coef_b = fma(om, wb, fma(om_p, wpb, om_s * wsb));
                        ADR_STAMP(6);   // (walk, date record) the three v, first-order sums, convexity coefficient
                        if (BOTH_ROWS) {
                            rank_one_rows(std::true_type{}, std::true_type{}, om, v1, coef_a, ra, coef_b, rb);
                        } else {
                            const double coa = convexity_coef(coef_a, coef_b);
                            rank_one(std::true_type{}, om, v1, coa);
                        }
                        mini_convexity(coef_a, coef_b);
                        if (__ballot(om_p != 0.0)) rank_one(std::false_type{}, om_p, vp, 0.0);
                        ADR_STAMP(7);   // (walk, date record) the two rank-one updates
                    }
                } else if (LAG && chunk_pass && role < 2) {
                    // ---- accrual start / end on knots of their own: adds to the node's v; first-order and convexity part
                    const double om_e = taken ? 0.0 : om;               // (a start record the previous date record has served)
#pragma unroll
                    for (int k = 0; k < PPL; ++k) {
                        const double vk = taken ? 0.0 : fma(wb, ub[k], wa * ua[k]);
                        dacc[k] = fma(om_e, vk, dacc[k]);
                        vacc[k] += vk;
                    }
                    if (GAMMA) {
                        if (BOTH_ROWS) {
                            lc_row_pass(ra, om_e * wa);
                            if (__ballot(rb != zero_row)) lc_row_pass(rb, om_e * wb);
                        } else {
                            const double coa = convexity_coef(om_e * wa, om_e * wb);
                            lc_row_pass(ra, coa);
                        }
                        mini_convexity(om_e * wa, om_e * wb);
                    }
                } else {
                    // ---- an ordinary node (LAG: a payment node on its own, or a fixed coupon)
                    const double om_n = taken ? 0.0 : om;
                    double v[PPL];
#pragma unroll
                    for (int k = 0; k < PPL; ++k) {
                        v[k] = fma(wb, ub[k], wa * ua[k]);
                        dacc[k] = fma(om_n, v[k], dacc[k]);
                    }
                    if (GAMMA && BOTH_ROWS) {
                        ADR_STAMP(6);
                        // (a branch for nodes on a single knot - one row - was measured: slower, it costs registers)
                        rank_one_rows(std::true_type{}, std::true_type{}, om_n, v, om_n * wa, ra, om_n * wb, rb);
                        ADR_STAMP(7);
                        mini_convexity(om_n * wa, om_n * wb);
                    } else if (GAMMA) {
                        const double coa = convexity_coef(om_n * wa, om_n * wb);
                        ADR_STAMP(6);   // (walk) v, first-order sums, convexity coefficient
                        if (!LAG && lindf) {
                            if (__ballot(flush_row != zero_row)) lindf_knot(flush_row, flush_w);
                            rank_one(std::true_type{}, coa, ua, coa);            // knot a (core row, or a short-end knot's entries)
                            if (__ballot(mini_b)) {                              // a short-end knot b is not carried
                                rank_one(std::false_type{}, mini_b ? wb : 0.0, ub, 0.0);
                                if (mini_b) carry_w = 0.0;
                            }
                        } else {
                            rank_one(std::true_type{}, om_n, v, coa);
                        }
                        ADR_STAMP(7);   // (walk) the rank-one update with the left knot's convexity row
                        mini_convexity(om_n * wa, om_n * wb);
                    }
                }
                if (!has_next) {
                    // LINEAR_FWD_RATES: the knot still carried gets its u u^T + LC here (the flush behind the passes adds
                    // convexity rows only)
                    if (!LAG && GAMMA && lindf && __ballot(carry_row != zero_row)) {
                        lindf_knot(carry_row, carry_w);
                        carry_row = zero_row; carry_w = 0.0;
                    }
                    break;
                }
            }

#if ADR_WALK_PRIO || ADR_BUILD_PRIO
            if (GAMMA) __builtin_amdgcn_s_setprio(ADR_BUILD_PRIO);
#endif
            ADR_STAMP(3);   // node consumption
        }
        if (GAMMA && __ballot(carry_row != zero_row)) lc_row_pass(carry_row, carry_w);
        // LAG: the scratch stores of this unit's special nodes must have landed before the output phase reads them back;
        // waited for HERE, where nothing else is in flight (later, the wait would also cover the next unit's input loads)
        if (LAG && __ballot(n_special > 0)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (LONG) {
            fresh = !more;
            if (more) {                        // the trade continues in this wave's next unit: no results yet
                if (GAMMA) load_unit(unit + wave_stride);
                pin_next();
                continue;
            }
        }

        // ------------------------------------------------------------------ results
        // pv and delta first: after them nothing refers to the per-lane trade index any more, so the next
        // unit's inputs can be loaded straight into the registers this unit used (a copy at the loop's back
        // edge would have to wait for the loads - and, vector memory operations retiring in order, for
        // every store issued behind them).
#pragma unroll
        for (int off = 1; off < L; off <<= 1) pv += __shfl_xor(pv, off, 64);
        if (live && l == 0) {
            if (out.pv) out.pv[t] = pv;
            tot_pv += pv;
        }
        if (DELTA) {
#pragma unroll
            for (int k = 0; k < PPL; ++k) {
                const int p = l + L * k;
                if (live && p < P && out.delta) __builtin_nontemporal_store(dacc[k] * 1e-4, out.delta + static_cast<int64_t>(t) * P + p);
                tot_delta[k] += dacc[k];
            }
        }
        int group_trade[G];
#pragma unroll
        for (int gg = 0; gg < G; ++gg) group_trade[gg] = __builtin_amdgcn_readfirstlane(__shfl(t, gg * L, 64));

        // ---- inputs of the wave's next unit, requested before this unit's (large) gamma stores
        if (GAMMA) load_unit(unit + wave_stride);

#if ADR_OUT_PRIO
        if (GAMMA) __builtin_amdgcn_s_setprio(ADR_OUT_PRIO);
#endif
        if (GAMMA) {
#pragma unroll
            for (int gg = 0; gg < G; ++gg) {
                const int tt = group_trade[gg];   // < 0: idle slot of the last unit, stored to the sink
                const int n_sp_gg = LAG ? __builtin_amdgcn_readfirstlane(__shfl(n_special, gg * L, 64)) : 0;
                __builtin_amdgcn_wave_barrier();
                if (g == gg) {
#pragma unroll
                    for (int i = 0; i < EPG; ++i) slot[l + L * i] = acc[i] * 1e-8;     // per bp^2
                }
                if (lane == 0) slot[kZeroEntry] = 0.0;
                wave_lds_sync();
                // LDS reads of a trade in kOutParts batches: the running-total slice with the first
#ifdef ADR_DEBUG_STORE_TO_DUMP      // diagnostic build: every gamma store goes to the 8 KB sink (L2), nothing to HBM
                double* gm = out.dump + 2 * lane;
#else
                double* gm = (tt >= 0 ? out.gamma + static_cast<int64_t>(tt) * (P * P) : out.dump) + 2 * lane;
#endif
                double* sink = out.dump + 2 * lane;            // pairs beyond a P < 32 matrix go here
#pragma unroll
                for (int part = 0; part < kOutParts; ++part) {
                    constexpr int kBands = 8 / kOutParts;
                    double ts_[EPL], gv[2 * kBands];
                    int mbs[kBands];
                    if (part == 0) {
#pragma unroll
                        for (int s = 0; s < EPL; ++s) ts_[s] = slot[lane + 64 * s];
                    }
#pragma unroll
                    for (int b = 0; b < kBands; ++b) mbs[b] = mm[kBands * part + b];
#pragma unroll
                    for (int b = 0; b < kBands; ++b) {
                        gv[2 * b] = slot[mbs[b] & 0xffff];
                        gv[2 * b + 1] = slot[mbs[b] >> 16];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (part == 0) {
#pragma unroll
                        for (int s = 0; s < EPL; ++s) tot_gamma[s] += ts_[s];
                    }
                    if (LAG && n_sp_gg > 0) {
                        // special ratio nodes of this trade: omega v_r v_c for this lane's elements that have no packed
                        // entry (their staging index is the zero entry), from the wave's scratch
                        const double* sv = out.lag_scratch + (static_cast<size_t>(blockIdx.x) * kWaves + wave) * (G * kLagScratchNodes * kLagStashDoubles) +
                                           gg * (kLagScratchNodes * kLagStashDoubles);
                        for (int sp = 0; sp < n_sp_gg; ++sp, sv += kLagStashDoubles) {
                            // all of a node's operands for this part's bands are requested together (one L2 round trip)
                            double vr[kBands];
                            nt_pair vc[kBands];
                            const double om_sp = __builtin_nontemporal_load(sv + kPillarPad) * 1e-8;
#pragma unroll
                            for (int b = 0; b < kBands; ++b) {
                                const int rc_ = band_rc[kBands * part + b];
                                vr[b] = __builtin_nontemporal_load(sv + (rc_ & 31));                 // (rows beyond P: masked below)
                                vc[b] = __builtin_nontemporal_load(reinterpret_cast<const nt_pair*>(sv + ((rc_ >> 8) & 30)));
                            }
#pragma unroll
                            for (int b = 0; b < kBands; ++b) {
                                const int band = kBands * part + b;
                                const bool inside = ((beyond >> band) & 1) == 0;
                                const bool miss0 = inside && (mbs[b] & 0xffff) == kZeroEntry, miss1 = inside && (mbs[b] >> 16) == kZeroEntry;
                                const double a0 = miss0 ? om_sp * vr[b] * vc[b].x : 0.0, a1 = miss1 ? om_sp * vr[b] * vc[b].y : 0.0;
                                gv[2 * b] += a0; gv[2 * b + 1] += a1;
                                if (tt >= 0) { tot_patch[2 * band] += a0; tot_patch[2 * band + 1] += a1; }
                            }
                        }
                    }
                    if (STORE) {
#pragma unroll
                        for (int b = 0; b < kBands; ++b) {
                            const int band = kBands * part + b;
                            // write-once output stream: non-temporal 16-byte stores (-3 % on the bench pass)
                            nt_pair pr; pr.x = gv[2 * b]; pr.y = gv[2 * b + 1];
#if defined(ADR_GAMMA_STORE_BITS)     // diagnostic builds: the store with explicit cache-policy bits (e.g. "sc1 nt")
                            {
                                const double* at_ = ((beyond >> band) & 1 ? sink : gm) + band * 128;
                                asm volatile("global_store_dwordx4 %0, %1, off " ADR_GAMMA_STORE_BITS :: "v"(at_), "v"(pr) : "memory");
                            }
#elif ADR_GAMMA_STORE_NT
                            __builtin_nontemporal_store(pr, reinterpret_cast<nt_pair*>(((beyond >> band) & 1 ? sink : gm) + band * 128));
#else
                            *reinterpret_cast<nt_pair*>(((beyond >> band) & 1 ? sink : gm) + band * 128) = pr;
#endif
                        }
                    }
                }
                // odd pillar count: the matrix's last element has no partner inside it (its pair went to the sink above) and is
                // stored on its own; the 16-byte stores of odd-numbered trades are 8-byte aligned only, which the hardware takes
                if (STORE && cv.odd_last != -3) {
                    const double x_last = cv.odd_last >= 0 ? slot[cv.odd_last] : 0.0;
                    if (lane == 0 && tt >= 0) __builtin_nontemporal_store(x_last, out.gamma + static_cast<int64_t>(tt) * (P * P) + (P * P - 1));
                }
            }
        }
#if ADR_OUT_PRIO
        if (GAMMA) __builtin_amdgcn_s_setprio(0);
#endif
        ADR_STAMP(4);   // outputs
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
        // gamma: packed 64-lane layout -> 4x4 blocks through the wave's slot
        double blk_gamma[GAMMA ? kGammaPerLane : 1];
        if (GAMMA) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < EPL; ++s) slot[lane + 64 * s] = tot_gamma[s];
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < kGammaPerLane; ++e) {
                const int m = cv.out_map[(4 * bi + (e >> 2)) * kPillarPad + 4 * bj + (e & 3)];
                blk_gamma[e] = m >= 0 ? slot[m] : 0.0;            // already per bp^2
            }
        }
        // delta and pv: sum the groups (lanes l, l + L, ... hold the same pillar)
#pragma unroll
        for (int off = L; off < 64; off <<= 1) {
            tot_pv += __shfl_xor(tot_pv, off, 64);
#pragma unroll
            for (int k = 0; k < PPL; ++k) tot_delta[k] += __shfl_xor(tot_delta[k], off, 64);
        }
        __syncthreads();   // every wave is done with the curve tables; reuse the LDS for the reduction
        double* red = reinterpret_cast<double*>(smem_raw);   // [waves][kAggStride]
        double* mine = red + wave * kAggStride;
        if (lane == 0) mine[0] = tot_pv;
        if (g == 0) {
#pragma unroll
            for (int k = 0; k < PPL; ++k) mine[1 + l + L * k] = DELTA ? tot_delta[k] * 1e-4 : 0.0;
        }
#pragma unroll
        for (int e = 0; e < kGammaPerLane; ++e) {
            const int r = 4 * bi + (e >> 2), q = 4 * bj + (e & 3);
            mine[1 + kPillarPad + r * kPillarPad + q] = GAMMA ? blk_gamma[GAMMA ? e : 0] : 0.0;
        }
        __syncthreads();
        if (LAG) {      // the patched elements: every wave adds to its own record `mine`, each lane its own 16 elements
            {
                {
#pragma unroll
                    for (int band = 0; band < 8; ++band) {
                        const int r = band_rc[band] & 0xff, q = band_rc[band] >> 8;
                        if (r < P) {
                            mine[1 + kPillarPad + r * kPillarPad + q] += tot_patch[2 * band];
                            mine[1 + kPillarPad + r * kPillarPad + q + 1] += tot_patch[2 * band + 1];
                        }
                    }
                }
            }
            __syncthreads();
        }
        double* dst = out.block_partials + static_cast<size_t>(blockIdx.x) * kAggStride;
        for (int i = threadIdx.x; i < kAggStride; i += kThreads) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += red[w * kAggStride + i];
            dst[i] = s;
        }
    }
}

#if !ADR_FAST_LINDF
// Fixed-order sum of the block partials -> agg[1 + P + P*P]: one wavefront per output, lanes stride over
// the blocks, then a fixed butterfly - the aggregate does not depend on scheduling.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* partials, int n_blocks, int P,
                                                               int has_gamma, double* agg, int ti, int tj) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // slot of the tile-local record
    if (i >= kAggStride) return;
    const bool diag = ti == tj;
    int at = -1, mirror = -1;                                                // where the slot goes in agg[1 + P + P*P]
    if (i == 0) {
        if (ti == 0 && tj == 0) at = 0;
    } else if (i < 1 + kPillarPad) {
        const int p = kPillarPad * ti + i - 1;
        if (diag && p < P) at = 1 + p;
    } else {
        const int r = kPillarPad * ti + (i - 1 - kPillarPad) / kPillarPad, q = kPillarPad * tj + (i - 1 - kPillarPad) % kPillarPad;
        if (r < P && q < P) {
            at = 1 + P + r * P + q;
            if (!diag) mirror = 1 + P + q * P + r;
        }
    }
    if (at < 0) return;
    if (!has_gamma && i >= 1 + kPillarPad) {      // nothing was accumulated there: the launches ran without GAMMA
        if (lane == 0) { agg[at] = 0.0; if (mirror >= 0) agg[mirror] = 0.0; }
        return;
    }
    double s = 0.0;
    for (int b = lane; b < n_blocks; b += 64) s += partials[static_cast<size_t>(b) * kAggStride + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) { agg[at] = s; if (mirror >= 0) agg[mirror] = s; }
}

#endif  // !ADR_FAST_LINDF

constexpr int kGroups = 2;   // trades per wavefront: the row table has 64 / 2 = 32 slots per row

// Kernel variant for a curve: (epg, cpg) as chosen by build_packed_layout - epg in {7, 8, 12, 18} slots per
// group lane, with cpg = epg - 2 (the last two slots hold the fringe pairs) or cpg = epg (universal).
using KernelFn = void (*)(CurveDev, TradesDev, OutputsDev);

template <int EPG, bool STORE, bool LONG, bool LAG>
KernelFn gamma_kernel(int cpg) {
    if (cpg == EPG - 2) return &price_fast_kernel<true, true, STORE, LONG, LAG, EPG, EPG - 2, kGroups>;
    return &price_fast_kernel<true, true, STORE, LONG, LAG, EPG, EPG, kGroups>;
}

template <bool STORE, bool LONG, bool LAG>
KernelFn gamma_kernel_for(const CurveDev& cv) {
    switch (cv.epg) {
        case 7: return gamma_kernel<7, STORE, LONG, LAG>(cv.cpg);
        case 8: return gamma_kernel<8, STORE, LONG, LAG>(cv.cpg);
        case 12: return gamma_kernel<12, STORE, LONG, LAG>(cv.cpg);
        default: return gamma_kernel<18, STORE, LONG, LAG>(cv.cpg);
    }
}

template <bool LONG>
KernelFn pick_kernel(const CurveDev& cv, bool want_delta, bool want_gamma, bool store_gamma) {
    if (!want_gamma)
        return want_delta ? &price_fast_kernel<true, false, false, LONG, false, 1, 1, kGroups>
                          : &price_fast_kernel<false, false, false, LONG, false, 1, 1, kGroups>;
    return store_gamma ? gamma_kernel_for<true, LONG, false>(cv) : gamma_kernel_for<false, LONG, false>(cv);
}

template <bool STORE, bool LONG, bool LAG>
void collect_gamma_kernels(std::vector<const void*>& fns) {
    for (int universal : {0, 1}) {
        fns.push_back(reinterpret_cast<const void*>(gamma_kernel<7, STORE, LONG, LAG>(universal ? 7 : 5)));
        fns.push_back(reinterpret_cast<const void*>(gamma_kernel<8, STORE, LONG, LAG>(universal ? 8 : 6)));
        fns.push_back(reinterpret_cast<const void*>(gamma_kernel<12, STORE, LONG, LAG>(universal ? 12 : 10)));
        fns.push_back(reinterpret_cast<const void*>(gamma_kernel<18, STORE, LONG, LAG>(universal ? 18 : 16)));
    }
}

}  // namespace

#if ADR_FAST_LINDF
// The LINEAR_FWD_RATES build of the kernels above (no payment-lag variant: its ratio nodes are single exponentials).
hipError_t launch_price_fast_lindf(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, bool want_delta,
                                   bool want_gamma, int n_blocks, size_t lds, hipStream_t stream) {
    if (tr.rows_lagged) return hipErrorInvalidValue;
    KernelFn fn = tr.rows_chained ? pick_kernel<true>(cv, want_delta, want_gamma, out.gamma != nullptr)
                                  : pick_kernel<false>(cv, want_delta, want_gamma, out.gamma != nullptr);
    hipLaunchKernelGGL(fn, dim3(n_blocks), dim3(kBlockThreads), lds, stream, cv, tr, out);
    return hipGetLastError();
}

hipError_t set_fast_lindf_lds_limit(size_t bytes) {
    std::vector<const void*> fns;
    collect_gamma_kernels<true, false, false>(fns);
    collect_gamma_kernels<false, false, false>(fns);
    collect_gamma_kernels<true, true, false>(fns);
    collect_gamma_kernels<false, true, false>(fns);
    fns.push_back(reinterpret_cast<const void*>(&price_fast_kernel<true, false, false, false, false, 1, 1, kGroups>));
    fns.push_back(reinterpret_cast<const void*>(&price_fast_kernel<false, false, false, false, false, 1, 1, kGroups>));
    fns.push_back(reinterpret_cast<const void*>(&price_fast_kernel<true, false, false, true, false, 1, 1, kGroups>));
    fns.push_back(reinterpret_cast<const void*>(&price_fast_kernel<false, false, false, true, false, 1, 1, kGroups>));
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
#else
hipError_t launch_price_fast_lindf(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, bool want_delta,
                                   bool want_gamma, int n_blocks, size_t lds, hipStream_t stream);
hipError_t set_fast_lindf_lds_limit(size_t bytes);

int fast_kernel_groups() { return kGroups; }

size_t fast_kernel_lag_scratch_bytes(int n_blocks) {
    return sizeof(double) * static_cast<size_t>(n_blocks) * (kLagThreads / 64) * kGroups * kLagScratchNodes * kLagStashDoubles;
}
int fast_kernel_threads(bool lagged) { return lagged ? kLagThreads : kBlockThreads; }

size_t fast_kernel_lds_bytes(const CurveDev& cv, bool gamma, bool lagged) {
    const size_t kWavesPerBlock = static_cast<size_t>(fast_kernel_threads(lagged)) / 64;
    const size_t slot = slot_doubles(gamma, cv.epg, kGroups);
    const size_t slack = kGroupLanes * cv.cpg > cv.Ec + 1 ? kGroupLanes * cv.cpg - (cv.Ec + 1) : 0;
    size_t doubles = static_cast<size_t>(cv.K) + 2 * cv.Kc + static_cast<size_t>(cv.Kcore + 1) * cv.pc_pad +
                     (gamma ? static_cast<size_t>(cv.Kcore + 1) * (cv.Ec + 1) + slack : 0);
    doubles += doubles & 1;   // the slots start on a 16-byte boundary
    doubles += kWavesPerBlock * slot;
    size_t tables = sizeof(MiniKnot) * cv.n_mini + sizeof(double) * doubles +
                    sizeof(int16_t) * (2 * static_cast<size_t>(cv.K) + cv.Kc + 2 * static_cast<size_t>(cv.n_lut));
    size_t reduce = sizeof(double) * kWavesPerBlock * kAggStride;
    size_t need = tables > reduce ? tables : reduce;
    return (need + 15) & ~static_cast<size_t>(15);
}

hipError_t launch_price_fast(const CurveDev& cv, const TradesDev& tr, const OutputsDev& out, bool want_delta,
                             bool want_gamma, int n_blocks, hipStream_t stream) {
    const size_t lds = fast_kernel_lds_bytes(cv, want_gamma, tr.rows_lagged != 0);
    if (cv.method == 2) return launch_price_fast_lindf(cv, tr, out, want_delta, want_gamma, n_blocks, lds, stream);
    KernelFn fn;
    if (tr.rows_lagged) {
        if (!want_gamma) return hipErrorInvalidValue;        // the payment-lag rows exist for gamma requests only
        if (tr.rows_chained) fn = out.gamma != nullptr ? gamma_kernel_for<true, true, true>(cv) : gamma_kernel_for<false, true, true>(cv);
        else fn = out.gamma != nullptr ? gamma_kernel_for<true, false, true>(cv) : gamma_kernel_for<false, false, true>(cv);
    } else {
        fn = tr.rows_chained ? pick_kernel<true>(cv, want_delta, want_gamma, out.gamma != nullptr)
                             : pick_kernel<false>(cv, want_delta, want_gamma, out.gamma != nullptr);
    }
    hipLaunchKernelGGL(fn, dim3(n_blocks), dim3(fast_kernel_threads(tr.rows_lagged != 0)), lds, stream, cv, tr, out);
    return hipGetLastError();
}

hipError_t launch_reduce_partials(const double* partials, int n_blocks, int P, bool has_gamma, double* agg,
                                  hipStream_t stream, int tile_i, int tile_j) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((kAggStride + 3) / 4), dim3(256), 0, stream, partials, n_blocks, P,
                       has_gamma ? 1 : 0, agg, tile_i, tile_j);
    return hipGetLastError();
}

hipError_t set_general_kernel_lds_limit(size_t bytes);
hipError_t set_lite_kernel_lds_limit(size_t bytes);

hipError_t set_kernel_lds_limits(size_t general_bytes, size_t fast_bytes) {
    hipError_t e = set_general_kernel_lds_limit(general_bytes);
    if (e != hipSuccess) return e;
    e = set_lite_kernel_lds_limit(fast_bytes);
    if (e != hipSuccess) return e;
    e = set_fast_lindf_lds_limit(fast_bytes);
    if (e != hipSuccess) return e;
    std::vector<const void*> fns;
    collect_gamma_kernels<true, false, false>(fns);
    collect_gamma_kernels<false, false, false>(fns);
    collect_gamma_kernels<true, true, false>(fns);
    collect_gamma_kernels<false, true, false>(fns);
    collect_gamma_kernels<true, false, true>(fns);
    collect_gamma_kernels<false, false, true>(fns);
    collect_gamma_kernels<true, true, true>(fns);
    collect_gamma_kernels<false, true, true>(fns);
    fns.push_back(reinterpret_cast<const void*>(&price_fast_kernel<true, false, false, false, false, 1, 1, kGroups>));
    fns.push_back(reinterpret_cast<const void*>(&price_fast_kernel<false, false, false, false, false, 1, 1, kGroups>));
    fns.push_back(reinterpret_cast<const void*>(&price_fast_kernel<true, false, false, true, false, 1, 1, kGroups>));
    fns.push_back(reinterpret_cast<const void*>(&price_fast_kernel<false, false, false, true, false, 1, 1, kGroups>));
    for (const void* f : fns) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(fast_bytes));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
#endif  // ADR_FAST_LINDF

}  // namespace adr
