// CDNA4 (gfx950): batched discount-factor lookups on an uploaded curve - the device counterpart of
// InterpolatorAd.simple_interpolate / curve.df_ad over arrays of times (cavour/market/curves/interpolator_ad.py:186-249).
//
// Used by the cross-currency assembly (adrates_amd/market/position/xccy_engine.py), where the reference evaluates
// D_x(tp_j) on the XCCY curve and D_f(ts_j) / D_f(te_j) on the foreign OIS curve inside its differentiable leg function
// (cavour/market/position/engine.py:1640-1712): one thread per query time, the curve's search arrays read through L2
// (a few KB, shared by every thread), the same snap / +1e-12 / duplicate-knot semantics as the pricing kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve_lookup.hpp"
#include "kernels.hpp"

namespace adr {

namespace {

struct CurveGlobal {
    const double* x;
    const double* log_df;
    const double* inv_x;
    const int16_t* lut;
    int n_lut;
    const int16_t* first_of;
    const int16_t* compact_of;
    int K, method;
};

__global__ __launch_bounds__(256) void curve_df_kernel(CurveDev cv, int64_t n, const double* __restrict__ t,
                                                       double* __restrict__ df) {
    CurveGlobal c;
    c.x = cv.x; c.log_df = cv.log_df; c.inv_x = cv.inv_x; c.lut = cv.lut; c.n_lut = cv.n_lut;
    c.first_of = cv.first_of; c.compact_of = cv.compact_of; c.K = cv.K;
    // LINEAR_FWD_RATES interpolates the knot DFs themselves (interpolator_ad.py:234-235): take the flat-forward
    // search (weights 1 - w, w) and apply them to the DFs instead of their logs
    c.method = cv.method == 2 ? 1 : cv.method;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const Lookup q = curve_lookup(c, t[i]);
        double d;
        if (cv.method == 2) d = q.ba * exp(c.log_df[q.ka]) + q.bb * exp(c.log_df[q.kb]);
        else d = exp(fma(q.ba, c.log_df[q.ka], q.bb * c.log_df[q.kb]));
        df[i] = d;
    }
}

}  // namespace

hipError_t launch_curve_df(const CurveDev& cv, int64_t n, const double* t_dev, double* df_dev, int n_cu, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int64_t need = (n + 255) / 256;
    const int blocks = static_cast<int>(need < static_cast<int64_t>(n_cu) * 8 ? need : static_cast<int64_t>(n_cu) * 8);
    hipLaunchKernelGGL(curve_df_kernel, dim3(blocks), dim3(256), 0, stream, cv, n, t_dev, df_dev);
    return hipGetLastError();
}

}  // namespace adr
