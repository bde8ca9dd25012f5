#include "curve_tables.hpp"
#include <cstdio>
#include <functional>
#include <cstdlib>

#include <algorithm>
#include <cmath>

namespace adr {

std::string build_curve_tables(int K, int P, const double* times, const double* dfs, const double* jac,
                               const double* hess, CurveTables& out) {
    if (K < 2) return "curve needs at least two knots";
    if (P < 1 || P > kMaxPillars) return "pillar count must be in 1.." + std::to_string(kMaxPillars);
    if (K > 32767) return "too many knots (int16 index tables)";
    if (!times || !dfs || !jac) return "times, dfs and jac must not be null";
    for (int k = 0; k < K; ++k) {
        if (!(dfs[k] > 0.0) || !std::isfinite(dfs[k])) return "discount factors must be positive and finite";
        if (!std::isfinite(times[k])) return "knot times must be finite";
        if (k > 0 && times[k] < times[k - 1]) return "knot times must be non-decreasing";
    }

    // The reference prices relative to the discount factor at the value time, D(tp)/D(0) (engine.py:2426-2435, 2669-2692);
    // its grids always start with the value-time point (t = 0, D = 1, no sensitivity), which is what lets the
    // kernels leave that division out.  A table that does not start there would be priced differently: refuse it.
    if (times[0] != 0.0 || dfs[0] != 1.0) return "the first knot must be the value time (t = 0, discount factor 1)";
    for (int p = 0; p < P; ++p)
        if (jac[p] != 0.0) return "the value-time knot (t = 0) must have no sensitivity";

    out = CurveTables();
    out.K = K;
    out.P = P;
    out.has_hess = hess != nullptr;
    out.x.assign(times, times + K);
    out.first_of.resize(K);
    out.compact_of.assign(K, -1);

    {   // knot-search table
        const double t_last = std::max(times[K - 1], 0.0);
        const int n_lut = std::min(kLutMax, static_cast<int>(t_last * kLutPerYear) + 2);
        out.lut.assign(2 * static_cast<size_t>(n_lut), 0);
        auto first_later = [&](double tau) { return static_cast<int>(std::upper_bound(times, times + K, tau) - times); };
        auto first_at_or_later = [&](double tau) { return static_cast<int>(std::lower_bound(times, times + K, tau) - times); };
        // A time t in bucket b satisfies b/4 <= t < (b+1)/4 (the scaling by 4 is exact), so the first knot later than t
        // is no earlier than the first knot later than b/4 and no later than the first knot AT OR AFTER (b+1)/4.  The
        // upper end matters: pillar dates sit on whole years = bucket boundaries and are runs of up to 17 duplicate
        // knots - bounded by "later than (b+1)/4" every search in the quarter before a pillar date had to walk the run.
        for (int b = 0; b < n_lut; ++b) {
            out.lut[2 * b] = static_cast<int16_t>(b == 0 ? 0 : first_later(b / kLutPerYear));
            out.lut[2 * b + 1] = static_cast<int16_t>(b + 1 == n_lut ? K : first_at_or_later((b + 1) / kLutPerYear));
        }
    }

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

    const int T = out.T = pillar_tiles(P), n_pairs = T * (T + 1) / 2;
    out.log_df.resize(Kc);
    out.inv_x.resize(Kc);
    out.lj.assign(static_cast<size_t>(T) * Kc * kPillarPad, 0.0);
    if (out.has_hess) {
        out.lc.assign(static_cast<size_t>(Kc) * P * P, 0.0);
        out.lc_lanes.assign(static_cast<size_t>(n_pairs) * Kc * 64 * kGammaPerLane, 0.0);
        out.lc_block_mask.assign(static_cast<size_t>(n_pairs) * Kc, 0);
    }

    std::vector<double> ljrow(static_cast<size_t>(P));
    for (int c = 0; c < Kc; ++c) {
        const int k = out.knot_index[c];
        const double d = dfs[k];
        out.log_df[c] = std::log(d);
        out.inv_x[c] = 1.0 / std::max(times[k], 1e-15);
        for (int p = 0; p < P; ++p) {
            ljrow[p] = jac[static_cast<size_t>(k) * P + p] / d;
            out.lj[(static_cast<size_t>(p / kPillarPad) * Kc + c) * kPillarPad + p % kPillarPad] = ljrow[p];
        }
        if (!out.has_hess) continue;
        const double* hk = hess + static_cast<size_t>(k) * P * P;
        double* lck = &out.lc[static_cast<size_t>(c) * P * P];
        for (int p = 0; p < P; ++p)
            for (int q = 0; q < P; ++q) lck[p * P + q] = hk[p * P + q] / d - ljrow[p] * ljrow[q];
        for (int tj = 0; tj < T; ++tj)
            for (int ti = 0; ti <= tj; ++ti) {
                const size_t pair = static_cast<size_t>(tile_pair(ti, tj));
                double* lanes = &out.lc_lanes[(pair * Kc + c) * 64 * kGammaPerLane];
                uint64_t mask = 0;
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < kGammaPerLane; ++e) {
                        const int r = kPillarPad * ti + gamma_row(lane, e), q = kPillarPad * tj + gamma_col(lane, e);
                        const double x = (r < P && q < P) ? lck[r * P + q] : 0.0;
                        lanes[lane * kGammaPerLane + e] = x;
                        if (x != 0.0) mask |= 1ull << lane;
                    }
                out.lc_block_mask[pair * Kc + c] = mask;
            }
    }
    if (T > 1 && P <= kWidePad) {   // wide layout (33-64 pillars): the whole ladder in one launch; beyond, the pillar tiles above
        const int nch = out.wide_nch = wide_chunks(P), row = nch * kWideChunk;
        out.lj64.assign(static_cast<size_t>(Kc) * kWidePad, 0.0);
        for (int c = 0; c < Kc; ++c)
            for (int p = 0; p < P; ++p)
                out.lj64[static_cast<size_t>(c) * kWidePad + p] = out.lj[(static_cast<size_t>(p / kPillarPad) * Kc + c) * kPillarPad + p % kPillarPad];
        // pillar order: the pillars most knots depend on first (stable: equal counts keep the pillar order)
        std::vector<int> uses(P, 0);
        for (int c = 0; c < Kc; ++c)
            for (int p = 0; p < P; ++p) {
                bool nz = out.lj64[static_cast<size_t>(c) * kWidePad + p] != 0.0;
                if (!nz && out.has_hess)
                    for (int q = 0; q < P && !nz; ++q) nz = out.lc[(static_cast<size_t>(c) * P + p) * P + q] != 0.0;
                if (nz) ++uses[p];
            }
        out.wide_order.resize(P);
        for (int p = 0; p < P; ++p) out.wide_order[p] = p;
        std::stable_sort(out.wide_order.begin(), out.wide_order.end(), [&](int x, int y) { return uses[x] > uses[y]; });
        out.wide_pos.assign(kWidePad, 0);
        for (int aq = 0; aq < P; ++aq) out.wide_pos[out.wide_order[aq]] = aq;
        for (int p = P; p < kWidePad; ++p) out.wide_pos[p] = p;              // lanes beyond the pillars keep their own (zero) slot
        // columns of the packed triangle, each padded to an even length: a lane's two entries are rows a, a + 1 of ONE column
        std::vector<int> col_off(P + 1, 0);
        for (int bq = 0; bq < P; ++bq) col_off[bq + 1] = col_off[bq] + 2 * ((bq + 2) / 2);
        const int E = col_off[P];                                             // <= row (wide_chunks)
        std::vector<int> ea(row, 0), eb(row, 0);
        std::vector<char> live(row, 0);
        for (int bq = 0; bq < P; ++bq)
            for (int aq = 0; aq < col_off[bq + 1] - col_off[bq]; ++aq) {
                ea[col_off[bq] + aq] = aq; eb[col_off[bq] + aq] = bq;
                live[col_off[bq] + aq] = aq <= bq;
            }
        out.wide_pq.assign(static_cast<size_t>(row) * 2, 255);
        for (int e = 0; e < E; ++e)
            if (live[e]) {
                out.wide_pq[2 * e] = static_cast<uint8_t>(out.wide_order[ea[e]]);
                out.wide_pq[2 * e + 1] = static_cast<uint8_t>(out.wide_order[eb[e]]);
            }
        out.wide_ent.assign(static_cast<size_t>(nch) * 64, 0u);
        for (int ch = 0; ch < nch; ++ch)
            for (int lane = 0; lane < 64; ++lane) {
                const int e0 = ch * kWideChunk + 2 * lane;
                if (e0 >= E) continue;
                out.wide_ent[static_cast<size_t>(ch) * 64 + lane] = static_cast<uint32_t>(ea[e0]) | (static_cast<uint32_t>(eb[e0]) << 8) |
                                                                    (live[e0] ? 1u << 16 : 0u) | (live[e0 + 1] ? 1u << 17 : 0u);
            }
        {
            const int bands = (P * P + 127) / 128;
            out.wide_store_map.assign(static_cast<size_t>(bands) * 64, 0xffffffffu);
            auto entry_of = [&](int f) -> uint32_t {
                if (f >= P * P) return 0xffffu;
                const int pr = out.wide_pos[f / P], pq = out.wide_pos[f % P];
                return static_cast<uint32_t>(col_off[std::max(pr, pq)] + std::min(pr, pq));
            };
            for (int band = 0; band < bands; ++band)
                for (int lane = 0; lane < 64; ++lane)
                    out.wide_store_map[static_cast<size_t>(band) * 64 + lane] = entry_of(band * 128 + 2 * lane) | (entry_of(band * 128 + 2 * lane + 1) << 16);
        }
        if (out.has_hess) {
            out.lcflat.assign(static_cast<size_t>(Kc) * row, 0.0);
            out.wide_knot_chunks.assign(Kc, 0u);
            for (int c = 0; c < Kc; ++c) {
                const double* lck = &out.lc[static_cast<size_t>(c) * P * P];
                for (int e = 0; e < E; ++e) {
                    if (!live[e]) continue;
                    const double x = lck[out.wide_order[ea[e]] * P + out.wide_order[eb[e]]];
                    out.lcflat[static_cast<size_t>(c) * row + e] = x;
                    if (x != 0.0) out.wide_knot_chunks[c] |= 1u << (e / kWideChunk);
                }
            }
        }
    }
    out.packed_ok = build_packed_layout(out);
    return std::string();
}

// Hub layout of the core pairs: every group lane's `cpg` core slots hold pairs that share one pillar (the
// lane's hub), so the kernel fetches the hub's value once instead of once per pair.  Each core pair {a, b} is
// given to one of its two pillars (a "star decomposition" of the complete graph on the core pillars, loops
// (a, a) staying with a); hub h gets k_h lanes, i.e. room for cpg * k_h pairs, the k_h as even as 32 lanes allow.
// The assignment is a bipartite matching with capacities, found with augmenting paths.  Returns false (layout
// untouched) when the pairs do not fit.
static bool hub_layout(int Pc, const std::vector<int>& core_pillars, int cpg, CurveTables& t) {
    if (Pc < 1 || Pc > kGroupLanes) return false;
    // IDENTITY positions: entry e = lane + 32 * slot of a convexity row sits at position e of the (compact) row, so a lane
    // reads its slots at one base address plus constants (no per-slot address arithmetic in the kernels).  The compact row
    // holds the Ec real pairs at positions 0 .. Ec - 1, so the padding slots must be exactly the entries e >= Ec: the LAST
    // slot of the last `n_short` lanes.  Hence: `n_full` lanes carry cpg pairs, the last `n_short` lanes cpg - 1, and every
    // hub's pair count (its loop included) is cpg * (its full lanes) + (cpg - 1) * (its short lanes) - exactly.
    const int Ec = Pc * (Pc + 1) / 2;
    const int n_short = cpg * kGroupLanes - Ec, n_full = kGroupLanes - n_short;
    if (n_short < 0 || n_full < 0) return false;
    std::vector<int> lanes_of(Pc, kGroupLanes / Pc);
    for (int h = 0; h < kGroupLanes % Pc; ++h) ++lanes_of[h];
    // short lanes: one per hub, hubs with the most lanes first (keeps the pair counts - the out-degrees of the star
    // decomposition - as even as the lane granularity allows), round robin beyond that
    std::vector<int> short_of(Pc, 0);
    {
        std::vector<int> order(Pc);
        for (int h = 0; h < Pc; ++h) order[h] = h;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lanes_of[a] > lanes_of[b]; });
        int left = n_short;
        for (int round = 0; left > 0 && round < kGroupLanes; ++round)
            for (int k = 0; k < Pc && left > 0; ++k) {
                const int h = order[k];
                if (short_of[h] < lanes_of[h]) { ++short_of[h]; --left; }
            }
        if (left > 0) return false;
    }
    std::vector<int> cap(Pc);
    int cap_sum = 0;
    for (int h = 0; h < Pc; ++h) {
        cap[h] = cpg * (lanes_of[h] - short_of[h]) + (cpg - 1) * short_of[h] - 1;   // one slot is the loop (h, h)
        if (cap[h] < 0) return false;
        cap_sum += cap[h];
    }
    if (cap_sum != Ec - Pc) return false;                          // (exact: every slot but the designated padding is a real pair)
    struct Edge { int a, b, at; };
    std::vector<Edge> edges;
    for (int a = 0; a < Pc; ++a)
        for (int b = a + 1; b < Pc; ++b) edges.push_back({a, b, -1});
    std::vector<int> load(Pc, 0);
    std::vector<std::vector<int>> held(Pc);                        // edges currently assigned to a hub
    std::vector<char> visited(Pc);
    // try to make room at hub h by moving one of its edges to that edge's other end (recursively)
    auto make_room = [&](auto&& self, int h) -> bool {
        if (load[h] < cap[h]) return true;
        if (visited[h]) return false;
        visited[h] = 1;
        for (size_t k = 0; k < held[h].size(); ++k) {
            const int ei = held[h][k];
            const int other = edges[ei].a == h ? edges[ei].b : edges[ei].a;
            if (self(self, other)) {
                held[h].erase(held[h].begin() + static_cast<long>(k));
                --load[h];
                edges[ei].at = other;
                held[other].push_back(ei);
                ++load[other];
                return true;
            }
        }
        return false;
    };
    for (size_t ei = 0; ei < edges.size(); ++ei) {
        bool placed = false;
        for (int h : {edges[ei].a, edges[ei].b}) {
            std::fill(visited.begin(), visited.end(), 0);
            if (make_room(make_room, h)) {
                edges[ei].at = h;
                held[h].push_back(static_cast<int>(ei));
                ++load[h];
                placed = true;
                break;
            }
        }
        if (!placed) return false;
    }
    for (int h = 0; h < Pc; ++h)
        if (load[h] != cap[h]) return false;

    // lanes: every hub's full lanes first (lanes 0 .. n_full - 1), then the short lanes (their last slot is the padding)
    const int n_core_entries = cpg * kGroupLanes;
    std::vector<uint8_t> pq(2 * static_cast<size_t>(n_core_entries));
    std::vector<char> real(n_core_entries, 0);
    int lane_full = 0, lane_short = n_full;
    for (int h = 0; h < Pc; ++h) {
        std::vector<int> partners{h};                              // the loop first
        for (int ei : held[h]) partners.push_back(edges[ei].a == h ? edges[ei].b : edges[ei].a);
        size_t k = 0;
        for (int j = 0; j < lanes_of[h]; ++j) {
            const bool is_short = j >= lanes_of[h] - short_of[h];
            const int lane = is_short ? lane_short++ : lane_full++;
            const int n_here = is_short ? cpg - 1 : cpg;
            for (int i = 0; i < cpg; ++i) {
                const int e = lane + kGroupLanes * i;
                const bool on = i < n_here;
                pq[2 * e] = static_cast<uint8_t>(core_pillars[h]);
                pq[2 * e + 1] = static_cast<uint8_t>(core_pillars[on ? partners[k + static_cast<size_t>(i)] : h]);
                real[e] = on;
            }
            k += static_cast<size_t>(n_here);
        }
        if (k != partners.size()) return false;
    }
    if (lane_full != n_full || lane_short != kGroupLanes) return false;
    // row positions: the identity (a padding slot - entry e >= Ec - reads the row's trailing zero or the head of the next
    // row: its accumulator is never output)
    t.core_pos.assign(n_core_entries, 0);
    t.lcc_pq.assign(2 * static_cast<size_t>(Ec), 0);
    for (int e = 0; e < n_core_entries; ++e) {
        t.core_pos[e] = static_cast<int16_t>(e);
        if (real[e]) {
            if (e >= Ec) return false;
            t.lcc_pq[2 * e] = pq[2 * e];
            t.lcc_pq[2 * e + 1] = pq[2 * e + 1];
        } else if (e < Ec) {
            return false;
        }
    }
    t.core_real.assign(real.begin(), real.end());
    t.ent_pq.assign(pq.begin(), pq.end());
    return true;
}

// Pillar-support analysis for the fast kernels.
//
// A bootstrapped knot DF depends only on the par rates of the swaps in its bootstrap chain, so most of
// LJ and of every LC matrix is structurally zero (README curve: at most 17 of 32 pillars per knot, always
// from the same 17 "long" pillars, while the 15 short pillars only touch their own single-period knots).
//   core set C   = union of all supports with at least three pillars;
//   core knot    = support inside C: a row of ljc (LJ on C) and of lcc (LC on the packed pairs of C x C);
//   mini knot    = any other knot - by construction its support has at most two pillars: a MiniKnot record;
//   null knot    = empty support (t = 0).
// A node (one or two knots) only creates gamma entries inside S x S with S the union of its knots'
// supports.  The packed entry list is all core pairs followed by every non-core pair some *possible* node
// can create - possible nodes are the adjacent (last-of-run, first-of-run) knot pairs and the single
// knots, which the curve alone determines.  Trades with ratio terms (payment lag) can couple three
// intervals and are not covered; they go to the general kernel.
bool build_packed_layout(CurveTables& t) {
    const int P = t.P, Kc = t.Kc;
    if (P > kPillarPad) return false;          // the packed layout (fast kernels) is for one pillar tile
    std::vector<uint32_t> support(Kc, 0u);
    for (int c = 0; c < Kc; ++c) {
        uint32_t m = 0;
        for (int p = 0; p < P; ++p) {
            bool nz = t.lj[static_cast<size_t>(c) * kPillarPad + p] != 0.0;
            if (!nz && t.has_hess)
                for (int q = 0; q < P && !nz; ++q) nz = t.lc[(static_cast<size_t>(c) * P + p) * P + q] != 0.0;
            if (nz) m |= 1u << p;
        }
        support[c] = m;
    }
    uint32_t core = 0;
    for (int c = 0; c < Kc; ++c)
        if (__builtin_popcount(support[c]) >= 3) core |= support[c];

    const int Pc = t.Pc = __builtin_popcount(core);
    // Curves with fewer than kMinCorePillars core pillars go to the general kernel.  For 1 <= Pc < 8 this is a
    // performance choice only (a handful of core pillars gains nothing from the LDS-resident layout).  For Pc == 0
    // (every knot depends on at most two pillars: a two- or three-pillar toy curve) it used to be a fault: with no
    // core pairs the slot choice below still picks an "exact" variant (cpg = epg - 2), `hub_layout` refuses Pc < 1 and
    // leaves `core_pos` EMPTY, the upload then passes a null `CurveDev::core_pos`, and the exact kernel variants
    // read `cv.core_pos[l + 32 * i]` for their per-lane row positions (kernels_fast.hip, "pos[i] = HUB ? ...") -
    // a null-pointer load on every lane.  The slot choice now also falls back to the universal variant (which never
    // reads core_pos) whenever the hub layout cannot be built, so the guard is no longer what prevents the fault.
    if (Pc < kMinCorePillars) return false;
    t.pc_pad = (Pc + 2) & ~1;
    t.pillar_to_core.assign(kPillarPad, static_cast<int16_t>(Pc));
    std::vector<int> core_pillars;
    for (int p = 0; p < P; ++p)
        if (core & (1u << p)) {
            t.pillar_to_core[p] = static_cast<int16_t>(core_pillars.size());
            core_pillars.push_back(p);
        }
    const int Ec = t.Ec = Pc * (Pc + 1) / 2;

    // packed entries: core pairs first (in packed_index order), then padding, then the fringe pairs
    std::vector<int> entry_of(kPillarPad * kPillarPad, -1);   // [p*32+q], p <= q
    t.ent_pq.clear();
    for (int i = 0; i < Pc; ++i)
        for (int j = i; j < Pc; ++j) {
            entry_of[core_pillars[i] * kPillarPad + core_pillars[j]] = static_cast<int>(t.ent_pq.size() / 2);
            t.ent_pq.push_back(static_cast<uint8_t>(core_pillars[i]));
            t.ent_pq.push_back(static_cast<uint8_t>(core_pillars[j]));
        }
    std::vector<std::pair<int, int>> fringe;
    std::vector<char> seen(kPillarPad * kPillarPad, 0);
    auto add_pairs = [&](uint32_t S) {
        for (int p = 0; p < P; ++p) {
            if (!(S & (1u << p))) continue;
            for (int q = p; q < P; ++q) {
                if (!(S & (1u << q)) || entry_of[p * kPillarPad + q] >= 0 || seen[p * kPillarPad + q]) continue;
                seen[p * kPillarPad + q] = 1;
                fringe.emplace_back(p, q);
            }
        }
    };
    for (int c = 0; c < Kc; ++c) {
        add_pairs(support[c]);                                  // snapped / extrapolated single knots
        if (c + 1 < Kc && t.x[t.knot_index[c]] != t.x[t.knot_index[c + 1]])
            add_pairs(support[c] | support[c + 1]);             // (last of a run, first of the next run)
    }
    // A group lane (32 lanes per trade) holds entries l + 32*i.  Fringe entries start on a 32-entry boundary
    // so that no lane slot mixes core pairs (whose convexity rows are read as contiguous 32-wide slices)
    // with fringe pairs.  The kernels exist for epg = 7, 8, 12, 18 slots per lane with the last two slots
    // fringe-only ("exact": core slots = epg - 2) or with every slot treated as core ("universal").
    const int core_slots_min = (Ec + kGroupLanes - 1) / kGroupLanes;
    const int n_fringe = static_cast<int>(fringe.size());
    static const int kEpgChoices[] = {7, 8, 12, 18};
    t.epg = 0;
    for (int epg : kEpgChoices) {
        if (n_fringe <= 2 * kGroupLanes && core_slots_min <= epg - 2) { t.epg = epg; t.cpg = epg - 2; break; }
        if (core_slots_min * kGroupLanes + n_fringe <= epg * kGroupLanes) { t.epg = epg; t.cpg = epg; break; }
    }
    if (t.epg == 0) return false;

    // position of a core pair in a convexity row: packed_index order unless the hub layout rearranges it
    t.lcc_pq.assign(t.ent_pq.begin(), t.ent_pq.end());
    t.core_pos.clear();
    t.hub = false;
    // Exact variants (two fringe slots per lane): a fringe pair is given to the lane of one of its own pillars - group
    // lane l holds pillar l's v in a register, so the rank-one term of these slots needs one LDS gather instead of two
    // (the kernels' exact instantiations rely on it).  Greedy: pairs with one candidate first, then the less loaded of
    // the two lanes; a curve it cannot place takes a universal variant, like one without a star decomposition.
    std::vector<int> own_slot(n_fringe, -1);                  // slot * 32 + lane
    bool own_ok = t.cpg < t.epg;
    if (own_ok) {
        // bipartite matching, pairs -> (lane, slot) with lane one of the pair's pillars: augmenting paths (Kuhn)
        std::vector<int> holder(2 * kGroupLanes, -1);         // pair held by slot * 32 + lane
        std::vector<char> visited;
        std::function<bool(int)> place = [&](int i) -> bool {
            const int cand[2] = {fringe[i].first, fringe[i].second};
            for (int c = 0; c < (cand[0] == cand[1] ? 1 : 2); ++c) {
                if (cand[c] >= kGroupLanes) continue;
                for (int slot = 0; slot < 2; ++slot) {
                    const int at = slot * kGroupLanes + cand[c];
                    if (visited[at]) continue;
                    visited[at] = 1;
                    if (holder[at] < 0 || place(holder[at])) { holder[at] = i; return true; }
                }
            }
            return false;
        };
        for (int i = 0; i < n_fringe && own_ok; ++i) {
            visited.assign(2 * kGroupLanes, 0);
            own_ok = place(i);
        }
        if (own_ok)
            for (int at = 0; at < 2 * kGroupLanes; ++at) {
                const int i = holder[at];
                if (i < 0) continue;
                own_slot[i] = at;
                if (fringe[i].first != at % kGroupLanes) std::swap(fringe[i].first, fringe[i].second);   // own pillar first
            }
    }
    const bool hub_ok = own_ok && hub_layout(Pc, core_pillars, t.cpg, t);
    if (t.cpg < t.epg && !hub_ok) {
        // no star decomposition for this curve: the exact variants would index an empty core_pos (see above).
        // Take the first universal variant the entries fit, or leave the curve to the general kernel.
        t.epg = 0;
        for (int epg : kEpgChoices)
            if (core_slots_min * kGroupLanes + n_fringe <= epg * kGroupLanes) { t.epg = t.cpg = epg; break; }
        if (t.epg == 0) return false;
    }
    const int fringe_start = t.fringe_start = (t.cpg < t.epg ? t.cpg : core_slots_min) * kGroupLanes;
    if (hub_ok) {
        // ent_pq / core_pos / lcc_pq now describe the star decomposition; redo entry_of for the core pairs
        for (int e = 0; e < t.cpg * kGroupLanes; ++e) {
            if (!t.core_real[e]) continue;
            const int p = t.ent_pq[2 * e], q = t.ent_pq[2 * e + 1];
            entry_of[std::min(p, q) * kPillarPad + std::max(p, q)] = e;
        }
        t.hub = true;
    }
    while (static_cast<int>(t.ent_pq.size() / 2) < fringe_start) { t.ent_pq.push_back(0); t.ent_pq.push_back(0); }
    std::vector<int> fringe_entry(n_fringe, -1);
    t.fringe_own = hub_ok;
    for (int i = 0; i < n_fringe; ++i) fringe_entry[i] = fringe_start + (t.fringe_own ? own_slot[i] : i);
    t.fringe_pos.assign(kPillarPad * kPillarPad, -1);
    const int n_entries = t.fringe_own ? fringe_start + 2 * kGroupLanes : fringe_start + n_fringe;
    while (static_cast<int>(t.ent_pq.size() / 2) < n_entries) { t.ent_pq.push_back(0); t.ent_pq.push_back(0); }
    for (int i = 0; i < n_fringe; ++i) {
        const int e = fringe_entry[i], a = fringe[i].first, b = fringe[i].second;
        entry_of[std::min(a, b) * kPillarPad + std::max(a, b)] = e;
        t.ent_pq[2 * e] = static_cast<uint8_t>(a);
        t.ent_pq[2 * e + 1] = static_cast<uint8_t>(b);
        t.fringe_pos[a * kPillarPad + b] = t.fringe_pos[b * kPillarPad + a] = i;
    }
    t.n_fringe = n_fringe;
    t.Eu = n_entries;

    t.out_map.assign(kPillarPad * kPillarPad, -1);
    for (int r = 0; r < P; ++r)
        for (int q = 0; q < P; ++q)
            t.out_map[r * kPillarPad + q] = static_cast<int16_t>(entry_of[std::min(r, q) * kPillarPad + std::max(r, q)]);

    // the same map by flat index r*P + c of the caller's [P][P] matrix (what the store loop walks); -2 beyond it
    t.store_map.assign(kPillarPad * kPillarPad, -2);
    for (int r = 0; r < P; ++r)
        for (int q = 0; q < P; ++q) t.store_map[r * P + q] = t.out_map[r * kPillarPad + q];
    t.odd_last = -3;
    if (P % 2) {     // the last element has no partner inside the matrix: its pair goes to the kernels' sink
        t.odd_last = t.store_map[P * P - 1] < 0 ? -1 : t.store_map[P * P - 1];
        t.store_map[P * P - 1] = -2;
    }

    t.knot_class.assign(Kc, -2);
    t.Kcore = 0;
    t.mini.clear();
    for (int c = 0; c < Kc; ++c) {
        if (support[c] == 0) continue;
        if ((support[c] & ~core) == 0) { t.knot_class[c] = static_cast<int16_t>(t.Kcore++); continue; }
        if (__builtin_popcount(support[c]) > 2) return false;   // cannot happen: such a support is in the core
        MiniKnot m{};
        m.p[0] = m.p[1] = -1;
        m.e[0] = m.e[1] = m.e[2] = -1;
        int n = 0;
        for (int p = 0; p < P; ++p)
            if (support[c] & (1u << p)) { m.p[n] = p; m.lj[n] = t.lj[static_cast<size_t>(c) * kPillarPad + p]; ++n; }
        if (t.has_hess) {
            const double* lck = &t.lc[static_cast<size_t>(c) * P * P];
            m.e[0] = entry_of[m.p[0] * kPillarPad + m.p[0]];
            m.lc[0] = lck[m.p[0] * P + m.p[0]];
            if (n == 2) {
                m.e[1] = entry_of[m.p[0] * kPillarPad + m.p[1]];
                m.lc[1] = lck[m.p[0] * P + m.p[1]];
                m.e[2] = entry_of[m.p[1] * kPillarPad + m.p[1]];
                m.lc[2] = lck[m.p[1] * P + m.p[1]];
            }
        }
        t.knot_class[c] = static_cast<int16_t>(-3 - static_cast<int>(t.mini.size()));
        t.mini.push_back(m);
    }
    t.n_mini = static_cast<int>(t.mini.size());

    t.lcc_pos.assign(kPillarPad * kPillarPad, static_cast<int16_t>(-1));
    for (int pos = 0; pos < Ec; ++pos) {
        const int a = t.lcc_pq[2 * pos], b = t.lcc_pq[2 * pos + 1];
        t.lcc_pos[a * kPillarPad + b] = t.lcc_pos[b * kPillarPad + a] = static_cast<int16_t>(pos);
    }
    for (int i = 0; i < kPillarPad * kPillarPad; ++i)        // fringe pairs: behind the row, in the order they were found
        if (t.fringe_pos[i] >= 0) t.lcc_pos[i] = static_cast<int16_t>(Ec + 1 + t.fringe_pos[i]);

    // one extra all-zero row (index Kcore) stands in for knots nothing depends on, so the kernel's hot loop
    // needs no branch for them
    t.ljc.assign(static_cast<size_t>(t.Kcore + 1) * t.pc_pad, 0.0);
    t.lcc.assign(t.has_hess ? static_cast<size_t>(t.Kcore + 1) * (Ec + 1) : 0, 0.0);   // trailing 0 per row
    for (int c = 0; c < Kc; ++c) {
        const int row = t.knot_class[c];
        if (row < 0) continue;
        for (int i = 0; i < Pc; ++i)
            t.ljc[static_cast<size_t>(row) * t.pc_pad + i] = t.lj[static_cast<size_t>(c) * kPillarPad + core_pillars[i]];
        if (!t.has_hess) continue;
        for (int pos = 0; pos < Ec; ++pos)
            t.lcc[static_cast<size_t>(row) * (Ec + 1) + pos] =
                t.lc[(static_cast<size_t>(c) * P + t.lcc_pq[2 * pos]) * P + t.lcc_pq[2 * pos + 1]];
    }
    return true;
}

}  // namespace adr
