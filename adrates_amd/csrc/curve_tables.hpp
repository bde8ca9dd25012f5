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
constexpr int kMaxPillars = 256;                                // 33-64 pillars: the wide kernel (one launch) or tiles of 32; beyond: tiles
constexpr int kWidePad = 64;                                    // wide kernel: ladders padded to one wavefront of pillars
constexpr int kWideChunk = 128;                                 // ... packed gamma entries per chunk: two per lane
constexpr int kWideMaxChunks = 17;                              // ... 64 * 65 / 2 = 2080 packed entries
// chunks per row of the packed triangle (columns padded to an even length), rounded up to the kernel variants
// (7: up to 41 pillars, 10: up to 49, 17: up to 64)
inline int wide_chunks(int P) {
    int e = 0;
    for (int b = 0; b < P; ++b) e += 2 * ((b + 2) / 2);
    const int n = (e + kWideChunk - 1) / kWideChunk;
    return n <= 7 ? 7 : (n <= 10 ? 10 : kWideMaxChunks);
}
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

    // ---- wide layout (33-64 pillars, kernels_general.hip WIDE variants): one wavefront = 64 pillars for v and the delta
    // ladder; the gamma matrix as its packed upper triangle in POSITION space - position a of `wide_order` = the pillars
    // sorted by how many knots depend on them, most first -, column by column (column b: rows a = 0 .. b, padded to an even
    // length): the pairs a knot's convexity touches sit at the front of the array, and a knot's row is read in the few
    // 128-entry chunks its `wide_knot_chunks` bit mask names.  Lane l of the wavefront holds entries 128 c + 2 l and
    // 128 c + 2 l + 1 of chunk c: rows a (even) and a + 1 of one column.
    int wide_nch = 0;                  // chunks of 128 packed entries: ceil(P (P + 1) / 2 / 128); 0: no wide tables (P <= 32)
    std::vector<int32_t> wide_order;   // [P] pillar at position a
    std::vector<double> lj64;          // [Kc][64] zero padded
    std::vector<int32_t> wide_pos;     // [64] position of pillar p (inverse of wide_order; identity beyond P)
    std::vector<uint32_t> wide_ent;    // [wide_nch][64] the lane's pair: row a | column b << 8 | (entry 0 inside the triangle) << 16 | (entry 1) << 17
    std::vector<double> lcflat;        // [Kc][wide_nch * 128] LC_k on the packed entries, zero beyond the triangle
    std::vector<uint32_t> wide_knot_chunks;   // [Kc] bit c: chunk c of the knot's row has a structural non-zero
    std::vector<uint8_t> wide_pq;             // [wide_nch * 128][2] pillars of the packed entry, 255 where the slot is padding
    std::vector<uint32_t> wide_store_map;     // [ceil(P * P / 128)][64] packed entry of element 128 band + 2 lane | the next one's << 16
                                              //                         of the row-major P x P matrix, 0xffff beyond it

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
    int odd_last = -3;                     // odd P: packed entry of element (P-1, P-1) (-1: none), whose pair of the flat array
                                           //        straddles the end of the matrix - its slot in store_map says -2 and the
                                           //        kernels store it on its own; -3 for even P
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
