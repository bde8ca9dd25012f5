// Aggregate-only mode, second half: the book's knot-space sums -> the pillar ladders.
//
// The reference assembles a trade's Greeks as  jac.T @ hess_dfs @ jac + sum_k g_k hess[k]  (cavour/market/position/
// engine.py:2551-2567: `jac` the knots' derivatives w.r.t. the par rates, `hess` their second derivatives, g / hess_dfs the
// trade's gradient / Hessian w.r.t. the knot discount factors), and Portfolio.compute adds the trades' ladders up
// (cavour/market/portfolio/portfolio.py:39-66).  Both steps are linear in (g, hess_dfs), so the book's ladder is the same
// expression on the SUMS of the trades' knot-space quantities - which is what the KNOT instantiations of the lite kernel
// (kernels_lite.hip) leave per block, in log space:
//
//   pv,   w_k = sum w b_k,   D_k = sum w b_k^2,   O_k = sum w b_k b_{k+1}      (nodes w = c exp(ba L[ka] + bb L[kb]))
//
// Ratio nodes (payment lag, weighted coupons: w exp(L(ts) - L(te) + L(tp))) couple knots of up to three intervals; their rows
// leave pair BANDS P_d[k] = sum w u_k u_{k+d}, d = 1 .. kKnotBand, and - for pairs farther apart - a dense overflow matrix.
//
// Here: (1) the block records are summed in a fixed order (one wavefront per number, lanes stride over the blocks, fixed
// butterfly - no atomics, the result does not depend on scheduling), (2) one projection per launch
//
//   delta_p  += 1e-4 sum_k w_k LJ[k][p]
//   gamma_pq += 1e-8 sum_k ( D_k LJ[k][p] LJ[k][q] + O_k (LJ[k][p] LJ[k+1][q] + LJ[k+1][p] LJ[k][q]) + w_k LC[k][p][q] )
//
// with LJ = d ln(knot DF) / d r, LC = d2 ln(knot DF) / d r2 read from whichever tables the curve carries (32-wide tiles or
// the wide layout of 33-64 pillars).  The projection ADDS to agg: the other kernel families' reduction (trades the knot
// kernel does not take) has written it before, or the caller has zeroed it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace adr {

namespace {

__global__ __launch_bounds__(256) void knot_reduce_kernel(const double* partials, int n_blocks, int stride, int n_values, double* reduced) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n_values) return;
    double s = 0.0;
    for (int b = lane; b < n_blocks; b += 64) s += partials[static_cast<size_t>(b) * stride + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) reduced[i] = s;
}

__device__ __forceinline__ double lj_at(const CurveDev& cv, int k, int p) {
    if (cv.wide_nch > 0 && cv.lj64) return cv.lj64[static_cast<size_t>(k) * kWidePad + p];
    return cv.lj[(static_cast<size_t>(p / kPillarPad) * cv.Kc + k) * kPillarPad + p % kPillarPad];
}

// LC[k][p][q]: the wide layout's packed triangle in position space, or the 32 x 32 tiles of the general kernel
__device__ __forceinline__ double lc_at(const CurveDev& cv, const int* col_off, int k, int p, int q) {
    if (cv.wide_nch > 0 && cv.lcflat) {
        int a = cv.wide_pos[p], b = cv.wide_pos[q];
        if (a > b) { const int t = a; a = b; b = t; }
        return cv.lcflat[static_cast<size_t>(k) * (cv.wide_nch * kWideChunk) + col_off[b] + a];
    }
    int ti = p / kPillarPad, tj = q / kPillarPad;
    if (ti > tj) { int t = p; p = q; q = t; t = ti; ti = tj; tj = t; }       // (symmetric)
    const int r = p % kPillarPad, c = q % kPillarPad;
    const int lane = (r >> 2) * 8 + (c >> 2), e = (r & 3) * 4 + (c & 3);
    return cv.lc_lanes[((static_cast<size_t>(tj * (tj + 1) / 2 + ti) * cv.Kc + k) * 64 + lane) * kGammaPerLane + e];
}

// Block p < P: row p of the gamma matrix (lane q); block P: pv and the delta ladder (lane p).  Sixteen wavefronts per block, each
// taking every sixteenth knot; their sums are added in wave order.  (One wavefront walking all Kc knots was a chain of Kc L2
// round trips: 85 us per aggregate-only pass of 100 k trades; 51 with four waves, 42 with sixteen.)
// The reduced sums are staged in LDS first (every thread walks all of them: as global loads behind `continue`s they were a
// chain of ~2 000 L2 round trips per thread - 0.7 ms for a 16-band record); the overflow matrix is scanned 64 entries at a
// time, coalesced, and only the non-zero ones (normally none) are visited.
constexpr int kProjectWaves = 16;

__global__ __launch_bounds__(64 * kProjectWaves) void knot_project_kernel(CurveDev cv, const double* reduced, int want_delta, int want_gamma,
                                                                          int bands, const double* overflow, double* agg) {
    extern __shared__ double s_red[];                     // [1 + (2 + bands) Kc]
    __shared__ int col_off[kWidePad + 1];
    __shared__ double s_part[kProjectWaves][64];
    const int P = cv.P, Kc = cv.Kc;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_values = want_gamma ? 1 + (2 + bands) * Kc : 1 + Kc;
    for (int i = threadIdx.x; i < n_values; i += 64 * kProjectWaves) s_red[i] = reduced[i];
    if (threadIdx.x == 0) {
        col_off[0] = 0;
        for (int b = 0; b < kWidePad; ++b) col_off[b + 1] = col_off[b] + 2 * ((b + 2) / 2);
    }
    __syncthreads();
    const double* w = s_red + 1;
    const double* D = w + Kc;
    const double* O = D + Kc;
    auto sum_of_waves = [&](double mine) {                // (block-uniform calls) the four waves' sums, added in wave order
        __syncthreads();
        s_part[wave][lane] = mine;
        __syncthreads();
        double t = s_part[0][lane];
#pragma unroll
        for (int i = 1; i < kProjectWaves; ++i) t += s_part[i][lane];
        return t;
    };
    if (static_cast<int>(blockIdx.x) == P) {
        if (threadIdx.x == 0) agg[0] += s_red[0];
        for (int q0 = 0; want_delta && q0 < P; q0 += 64) {
            const int q = q0 + lane, qq = q < P ? q : 0;
            double s = 0.0;
            for (int k = wave; k < Kc; k += kProjectWaves) s = fma(w[k], lj_at(cv, k, qq), s);
            s = sum_of_waves(s);
            if (wave == 0 && q < P) agg[1 + q] += s * 1e-4;
        }
        return;
    }
    if (!want_gamma) return;
    const int p = blockIdx.x;
    const bool any_overflow = overflow && overflow[static_cast<size_t>(Kc) * Kc] != 0.0;    // set by the kernel that used the matrix
    for (int q0 = 0; q0 < P; q0 += 64) {                   // columns in blocks of one wavefront (more than 64 pillars: several)
        const int q = q0 + lane;
        const int qq = q < P ? q : 0;                          // (lanes beyond the ladder compute a copy of column 0 and do not store)
        double s = 0.0;
        for (int k = wave; k < Kc; k += kProjectWaves) {
            const double wk = w[k], dk = D[k];
            const double ap = lj_at(cv, k, p), aq = lj_at(cv, k, qq);
            s = fma(dk * ap, aq, s);
            for (int d = 1; d <= bands && k + d < Kc; ++d) {             // pairs of knots (k, k + d): band d
                const double ok = O[(d - 1) * Kc + k];
                if (ok == 0.0) continue;                                 // (wave-uniform: the sums are the same for every lane)
                const double bp = lj_at(cv, k + d, p), bq = lj_at(cv, k + d, qq);
                s = fma(ok, fma(ap, bq, bp * aq), s);
            }
            if (any_overflow)                                            // ... and the pairs farther apart (payment-lag rows; rare)
                for (int l0 = k + bands + 1; l0 < Kc; l0 += 64) {
                    const int l = l0 + lane;
                    const double mine = l < Kc ? overflow[static_cast<size_t>(k) * Kc + l] : 0.0;
                    unsigned long long any = __ballot(mine != 0.0);
                    while (any) {
                        const int src = __builtin_ctzll(any);
                        any &= any - 1;
                        const double ok = __shfl(mine, src, 64);
                        const double bp = lj_at(cv, l0 + src, p), bq = lj_at(cv, l0 + src, qq);
                        s = fma(ok, fma(ap, bq, bp * aq), s);
                    }
                }
            if (wk != 0.0) s = fma(wk, lc_at(cv, col_off, k, p, qq), s);
        }
        s = sum_of_waves(s);
        if (wave == 0 && q < P) agg[1 + P + p * P + q] += s * 1e-8;
    }
}

}  // namespace

hipError_t launch_knot_project(const CurveDev& cv, const double* partials, int n_blocks, double* reduced, bool want_delta,
                               bool want_gamma, int bands, const double* overflow, double* agg, hipStream_t stream) {
    const int stride = 1 + (2 + bands) * cv.Kc, n_values = want_gamma ? stride : 1 + cv.Kc;     // (without GAMMA only pv and w were written)
    hipLaunchKernelGGL(knot_reduce_kernel, dim3((n_values + 3) / 4), dim3(256), 0, stream, partials, n_blocks, stride, n_values, reduced);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(knot_project_kernel, dim3(cv.P + 1), dim3(64 * kProjectWaves), sizeof(double) * static_cast<size_t>(stride), stream, cv, reduced,
                       want_delta ? 1 : 0, want_gamma ? 1 : 0, bands, want_gamma ? overflow : nullptr, agg);
    return hipGetLastError();
}

}  // namespace adr
