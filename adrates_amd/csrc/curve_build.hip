// CDNA4 (gfx950) curve builder: bootstrap of the engine knot grid with its first and second par-rate
// derivatives, for a batch of par-rate scenarios on one grid, and the conversion of every scenario to the
// log-space tables the pricing kernels consume.
//
// Replaces, for scenario batches (Model.scenario, cavour/models/models.py:507-557, which rebuilds a model per
// shock), Engine.build_curve_ad's lax.scan (cavour/market/position/engine.py:2336-2360) and the jacrev /
// hessian of Engine._cached_curve (:2388-2389):
//
//   d_i   = (1 - r PV01_prev) / (1 + r a)            r = par rate of the knot's swap, a = accrual fraction
//   PV01_i = PV01_prev + a d_i
//
// with the derivatives propagated in closed form next to the values (the same forward recurrences the host
// builder uses, adrates_amd/market/curves/curve_tables.py::build_engine_curve).
// Floating-point contraction is off in both kernels so that the device results are the IEEE operations of
// the host builder in the same order - device-built and host-built tables agree bit for bit (up to the
// libm `log`).
//
// bootstrap_kernel: one 256-thread block per scenario walks the K knots in order (a knot's predecessor is
// an earlier knot); PV01 and its gradient stay in LDS (or, when K x P doubles exceed it - a 64-pillar grid has
// 1 242 knots -, go through a second scratch buffer), the K x P x P second-derivative state goes through
// a scratch buffer in HBM/L2 (8 KB per knot at 32 pillars); thread t owns entries t, t+256, ... of the P x P matrices.
// pack_kernel: one block per (reachable knot, scenario) converts to log space and scatters into the dense
// tables of the general kernel and the packed LDS tables of the fast kernel - or, for 33-64 pillars, into the
// wide layout's tables (64-wide Jacobian rows, convexity rows on the packed triangle).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace adr {

namespace {

constexpr int kBuildThreads = 256;

__global__ __launch_bounds__(kBuildThreads) void bootstrap_kernel(CurveBuildPlanDev plan, const double* rates,
                                                                 double* dfs_out, double* jac_out, double* hess_out,
                                                                 double* d2pv_scratch, double* dpv_scratch) {
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int K = plan.K, P = plan.P, PP = P * P;
    const int scen = blockIdx.x;
    double* s_pv = reinterpret_cast<double*>(smem_raw);   // [K]
    double* s_dd = s_pv + K;                               // [64] gradient of the current knot
    double* s_cross = s_dd + kWidePad;                     // [64]
    // [K][P] PV01 gradients: in LDS, or in this scenario's slice of the scratch (same-block writes and reads around the
    // barriers below, as for the second-derivative state)
    double* s_dpv = plan.dpv_global ? dpv_scratch + static_cast<size_t>(scen) * K * P : s_cross + kWidePad;

    const double* r_s = rates + static_cast<size_t>(scen) * P;
    double* dfs = dfs_out + static_cast<size_t>(scen) * K;
    double* jac = jac_out + static_cast<size_t>(scen) * K * P;
    double* hess = hess_out ? hess_out + static_cast<size_t>(scen) * K * PP : nullptr;
    double* d2pv = hess_out ? d2pv_scratch + static_cast<size_t>(scen) * K * PP : nullptr;
    const int t = threadIdx.x;

    for (int i = 0; i < K; ++i) {
        const int s = plan.pillar[i];
        const int pi = plan.prev_idx[i];
        const double r = r_s[s], a = plan.acc[i];
        // a predecessor that sorts after the point has not been written yet in the reference's scan: reads 0
        const bool has_prev = pi >= 0 && pi < i;
        const double Pp = has_prev ? s_pv[pi] : 0.0;
        const double v = 1.0 + r * a;
        const double d = pi >= 0 ? (1.0 - r * Pp) / v : 1.0 / v;
        const double f = -r / v;
        if (t < P) {
            const double dPp = has_prev ? s_dpv[static_cast<size_t>(pi) * P + t] : 0.0;
            double dd = f * dPp;
            if (t == s) dd -= (Pp + d * a) / v;
            s_dd[t] = dd;
            s_cross[t] = has_prev ? dd * (a / v) + dPp / v : dd * (a / v);
            jac[static_cast<size_t>(i) * P + t] = dd;
        }
        __syncthreads();
        if (t < P) s_dpv[static_cast<size_t>(i) * P + t] = (has_prev ? s_dpv[static_cast<size_t>(pi) * P + t] : 0.0) + a * s_dd[t];
        if (t == 0) { s_pv[i] = Pp + a * d; dfs[i] = d; }
        if (hess) {
            const double* prev2 = has_prev ? d2pv + static_cast<size_t>(pi) * PP : nullptr;
            for (int e = t; e < PP; e += kBuildThreads) {
                const int p = e / P, q = e - p * P;
                const double h_prev = prev2 ? prev2[e] : 0.0;
                double h = f * h_prev;
                if (p == s) h -= s_cross[q];
                if (q == s) h -= s_cross[p];
                hess[static_cast<size_t>(i) * PP + e] = h;
                d2pv[static_cast<size_t>(i) * PP + e] = h_prev + a * h;
            }
        }
        __syncthreads();   // LDS state and (workgroup-scope) global writes of knot i visible to the next knot
    }
}

__global__ __launch_bounds__(kBuildThreads) void pack_kernel(CurveBuildPlanDev plan, const double* dfs_in,
                                                            const double* jac_in, const double* hess_in,
                                                            CurvePackOut out) {
#pragma clang fp contract(off)
    __shared__ double s_lj[kWidePad];
    const int K = plan.K, P = plan.P, PP = P * P, Kc = plan.Kc;
    const int c = blockIdx.x, scen = blockIdx.y, t = threadIdx.x;
    const int k = plan.knot_index[c];
    const double d = dfs_in[static_cast<size_t>(scen) * K + k];
    const double* jrow = jac_in + (static_cast<size_t>(scen) * K + k) * P;
    const double* hk = hess_in ? hess_in + (static_cast<size_t>(scen) * K + k) * PP : nullptr;

    if (t < kWidePad) s_lj[t] = t < P ? jrow[t] / d : 0.0;
    __syncthreads();

    double* log_df = out.log_df + static_cast<size_t>(scen) * Kc;
    if (t == 0) log_df[c] = log(d);
    if (plan.wide_nch > 0) {      // 33-64 pillars: the wide layout only
        double* lj64 = out.lj64 + (static_cast<size_t>(scen) * Kc + c) * kWidePad;
        if (t < kWidePad) lj64[t] = s_lj[t];
        if (hk) {
            const int row = plan.wide_nch * kWideChunk;
            double* dst = out.lcflat + (static_cast<size_t>(scen) * Kc + c) * row;
            for (int e = t; e < row; e += kBuildThreads) {
                const int p = plan.wide_pq[2 * e], q = plan.wide_pq[2 * e + 1];
                if (p != 255) dst[e] = hk[p * P + q] / d - s_lj[p] * s_lj[q];
            }
        }
        return;
    }
    double* lj = out.lj + (static_cast<size_t>(scen) * Kc + c) * kPillarPad;
    if (t < kPillarPad) lj[t] = s_lj[t];

    const int cls = plan.knot_class ? plan.knot_class[c] : -2;
    if (hk) {
        // dense lane-major layout of the general kernel: lane = 4x4 block (r/4, q/4), element (r%4, q%4)
        double* lanes = out.lc_lanes + (static_cast<size_t>(scen) * Kc + c) * 64 * kGammaPerLane;
        for (int e = t; e < kPillarPad * kPillarPad; e += kBuildThreads) {
            const int r = e / kPillarPad, q = e % kPillarPad;
            const double lc = (r < P && q < P) ? hk[r * P + q] / d - s_lj[r] * s_lj[q] : 0.0;
            const int lane = (r >> 2) * 8 + (q >> 2), slot = (r & 3) * 4 + (q & 3);
            lanes[lane * kGammaPerLane + slot] = lc;
        }
    }
    if (!plan.packed_ok) return;
    if (cls >= 0) {
        double* ljc = out.ljc + (static_cast<size_t>(scen) * (plan.Kcore + 1) + cls) * plan.pc_pad;
        if (t < plan.Pc) ljc[t] = s_lj[plan.core_pillars[t]];
        if (hk) {
            double* lcc = out.lcc + (static_cast<size_t>(scen) * (plan.Kcore + 1) + cls) * (plan.Ec + 1);
            for (int e = t; e < plan.Ec; e += kBuildThreads) {
                const int p = plan.lcc_pq[2 * e], q = plan.lcc_pq[2 * e + 1];
                lcc[e] = hk[p * P + q] / d - s_lj[p] * s_lj[q];
            }
        }
    } else if (cls <= -3 && t == 0) {
        MiniKnot* m = out.mini + static_cast<size_t>(scen) * plan.n_mini + (-3 - cls);
        const int p0 = m->p[0], p1 = m->p[1];          // structure copied from the plan's base curve
        m->lj[0] = s_lj[p0];
        if (p1 >= 0) m->lj[1] = s_lj[p1];
        if (hk) {
            m->lc[0] = hk[p0 * P + p0] / d - s_lj[p0] * s_lj[p0];
            if (p1 >= 0) {
                m->lc[1] = hk[p0 * P + p1] / d - s_lj[p0] * s_lj[p1];
                m->lc[2] = hk[p1 * P + p1] / d - s_lj[p1] * s_lj[p1];
            }
        }
    }
}

}  // namespace

size_t bootstrap_kernel_lds_bytes(int K, int P, bool dpv_global) {
    return sizeof(double) * (static_cast<size_t>(K) * (dpv_global ? 1 : P + 1) + 2 * kWidePad);
}

hipError_t launch_curve_build(const CurveBuildPlanDev& plan, int n_scen, const double* rates_dev, double* dfs,
                              double* jac, double* hess, double* d2pv_scratch, double* dpv_scratch,
                              const CurvePackOut& out, hipStream_t stream) {
    if (n_scen <= 0) return hipSuccess;
    const size_t lds = bootstrap_kernel_lds_bytes(plan.K, plan.P, plan.dpv_global != 0);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bootstrap_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bootstrap_kernel, dim3(n_scen), dim3(kBuildThreads), lds, stream, plan, rates_dev, dfs, jac,
                       hess, d2pv_scratch, dpv_scratch);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pack_kernel, dim3(plan.Kc, n_scen), dim3(kBuildThreads), 0, stream, plan, dfs, jac, hess, out);
    return hipGetLastError();
}

}  // namespace adr
