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

constexpr int kPillarPad = 32;                                  // ladders are padded to this on chip (one pillar tile)
constexpr int kMaxPillars = 64;                                 // more than 32 pillars: the wide kernel (one launch), or tiles of 32
constexpr int kWidePad = 64;                                    // wide kernel: ladders padded to one wavefront of pillars
constexpr int kWideMaxBlocks = 3;                               // ... 4x4 gamma blocks per lane: 136 upper blocks of a 64 x 64 matrix
constexpr int kAggWide = 1 + kWidePad + kWidePad * kWidePad;    // ... padded [pv, delta, gamma] record of a block partial
inline int pillar_tiles(int P) { return (P + kPillarPad - 1) / kPillarPad; }
inline int tile_pair(int ti, int tj) { return tj * (tj + 1) / 2 + ti; }       // ti <= tj
constexpr int kGammaPerLane = kPillarPad * kPillarPad / 64;     // 16 gamma entries per lane (4x4 block)
constexpr int kMinCorePillars = 8;   // fewer core pillars than this: no packed layout (general kernel)
constexpr int kGroupLanes = 32;                                 // lanes per trade in the fast kernel
constexpr double kLutPerYear = 4.0;                             // buckets per year of the knot-search table
constexpr int kLutMax = 512;                                    // at most this many buckets (128 years)

// A knot whose log-DF depends on at most two par rates (single-period calibration swaps of the short
// end): its whole first/second-derivative information is 2 + 3 numbers.
struct MiniKnot {
    int32_t p[2];      // pillars, -1 when unused
    int32_t e[3];      // packed entries of (p0,p0), (p0,p1), (p1,p1), -1 when unused
    int32_t pad;
    double lj[2];
    double lc[3];
};
static_assert(sizeof(MiniKnot) == 64, "MiniKnot is read as 64-byte records");

struct CurveTables {
    int K = 0;    // knots of the caller's grid
    int P = 0;    // pillars
    int Kc = 0;   // knots kept
    bool has_hess = false;
    bool packed_ok = false;            // packed layout below is usable
    std::vector<double> x;             // [K]   knot times (full grid, for the search)
    // search accelerator: bucket b = [b / kLutPerYear, (b + 1) / kLutPerYear) of time; the first knot later than
    // a time in the bucket has an index in [lut[2b], lut[2b + 1]] = [first knot later than the bucket's start, first
    // knot at or after its end] (bucket 0 also takes negative times, the last bucket everything later)
    std::vector<int16_t> lut;          // [n_lut][2]
    std::vector<int32_t> first_of;     // [K]   first index of the run of equal times containing k
    std::vector<int32_t> compact_of;   // [K]   row of knot k in the compact tables, -1 if unreachable
    std::vector<int32_t> knot_index;   // [Kc]  inverse of compact_of
    std::vector<double> log_df;        // [Kc]
    std::vector<double> inv_x;         // [Kc]  1 / max(x_k, 1e-15)   (linear-zero-rate weights)
    int T = 1;                         // pillar tiles of 32: ceil(P / 32)
    std::vector<double> lj;            // [T][Kc][kPillarPad], zero padded: pillar 32 t + j of knot c at (t * Kc + c) * 32 + j
    std::vector<double> lc;            // [Kc][P][P] row-major (plain layout, for checking)
    std::vector<double> lc_lanes;      // [pairs][Kc][64][16] lane-major 32x32 tiles read by the general gamma kernel; tile pair
                                       //                     (ti <= tj) at index tile_pair(ti, tj), pair 0 = the only one for P <= 32
    std::vector<uint64_t> lc_block_mask;  // [pairs][Kc] bit l: lane l's 4x4 block of that tile of LC_k has a non-zero entry

    // ---- wide layout (33-64 pillars, kernels_general.hip WIDE variants): one wavefront = 64 pillars, the upper triangle
    // of the gamma matrix in 4x4 blocks dealt to the lanes, block u of the row-major upper triangle to lane u % 64, slot u / 64
    int wide_bpl = 0;                  // blocks per lane (1..kWideMaxBlocks); 0: no wide tables (P <= 32)
    std::vector<double> lj64;          // [Kc][64] zero padded
    std::vector<int16_t> wide_blk;     // [wide_bpl][64] block row | block column << 8 of the lane's block in that slot, -1: none
    std::vector<double> lcw;           // [Kc][wide_bpl][64][16] LC_k on the lanes' blocks (entry 4 i + j = row 4 bi + i, column 4 bj + j)
    std::vector<uint64_t> lcw_mask;    // [Kc][wide_bpl] bit l: lane l's block in that slot has a structural non-zero of LC_k

    // ---- packed layout of the fast kernels (see build_packed_layout) ----
    int Pc = 0;                        // pillars in the core set
    int pc_pad = 0;                    // row stride of ljc: >= Pc + 1 (column Pc is all zero), even
    int Ec = 0;                        // Pc*(Pc+1)/2 core x core pairs = the first Ec packed entries
    int Eu = 0;                        // all packed entries: core pairs, padding to a multiple of 32, fringe pairs
    int epg = 0;                       // packed entries per group lane the kernel is instantiated for (Eu <= 32*epg)
    int cpg = 0;                       // of which slots that read convexity rows: epg - 2 (exact) or epg (universal)
    int fringe_start = 0;              // first packed entry of the fringe pairs
    int n_fringe = 0;                  // number of fringe pairs
    bool fringe_own = false;           // every fringe pair sits in the lane of its first pillar (entry % 32 == that pillar)
    std::vector<int> fringe_pos;       // [32*32] ordinal of the fringe pair (p, q), -1 if none (scratch of the layout builder)
    int Kcore = 0;                     // rows of ljc / lcc
    int n_mini = 0;                    // knots with at most two pillars outside the core
    std::vector<int16_t> pillar_to_core;   // [32]   core column of pillar p; Pc (the zero column) outside the core
    std::vector<int16_t> knot_class;       // [Kc]   >= 0: core row; -2: all-zero knot; <= -3: mini record -3 - m
    std::vector<double> ljc;               // [Kcore + 1][pc_pad]  LJ on the core pillars; last row all zero
    std::vector<double> lcc;               // [Kcore + 1][Ec + 1]  LC on the packed core pairs, then a 0; last row zero
    std::vector<uint8_t> ent_pq;           // [Eu][2]          pillars of packed entry e (hub layout: hub first)
    bool hub = false;                      // core slots of a lane share their first pillar (see hub_layout)
    std::vector<int16_t> core_pos;         // [32*cpg]         hub layout: position of core entry e in a lcc row
    std::vector<char> core_real;           // [32*cpg]         hub layout: entry e is a real pair (not padding)
    std::vector<uint8_t> lcc_pq;           // [Ec][2]          pillars of the pair stored at position pos of a lcc row
    std::vector<int16_t> lcc_pos;          // [32*32]          flat index of pair (r, c), either order: position in a lcc row (core),
                                           //                  Ec + 1 + ordinal of the fringe pair, -1 for pairs no node creates
    std::vector<int16_t> out_map;          // [32*32]          packed entry feeding gamma[r][c] (32-wide rows), -1 if none
    std::vector<int16_t> store_map;        // [32*32]          the same by flat index r*P + c; -2 beyond P*P
    std::vector<MiniKnot> mini;            // [n_mini]
};

// Row/column of gamma entry e (0..15) held by lane l (0..63): a 4x4 block at (4*(l/8), 4*(l%8)).
inline int gamma_row(int lane, int e) { return 4 * (lane >> 3) + (e >> 2); }
inline int gamma_col(int lane, int e) { return 4 * (lane & 7) + (e & 3); }

// Packed index of core pair (i <= j) among Pc core pillars, row-major upper triangle.
inline int packed_index(int i, int j, int Pc) { return i * Pc - i * (i - 1) / 2 + (j - i); }

// Returns false when the curve does not have the sparse structure the packed layout needs (the fast
// kernels are then not used for it).
bool build_packed_layout(CurveTables& t);

// Returns an empty string on success, otherwise the reason the inputs were rejected.
std::string build_curve_tables(int K, int P, const double* times, const double* dfs, const double* jac,
                               const double* hess, CurveTables& out);

}  // namespace adr
