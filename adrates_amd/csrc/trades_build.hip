// Device-side construction of the pricing kernels' trade tables from the uploaded CSR arrays.
//
// adr_trades_upload copies the caller's arrays (the per-trade arrays the reference's engine extracts from the legs,
// cavour/market/position/engine.py:2519-2527, 2858-2877) to the device ONCE; the padded, sorted row tables of the fast
// kernel, the 16-slot rows of the lite kernel and the 32-byte trade headers of the general kernel are then gathered
// from them here - a few GB of HBM traffic (about a millisecond per million trades) instead of several GB written by
// one host thread and pushed over PCIe.  The host keeps what needs the whole batch: validation, the routing classes
// and the (counting-)sorted row order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace adr {

namespace {

__global__ __launch_bounds__(256) void build_headers_kernel(CsrDev csr, TradeHeader* out) {
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= csr.n) return;
    TradeHeader h;
    h.notional = csr.notional[t];
    h.spread = csr.spread[t];
    h.flt_begin = static_cast<int32_t>(csr.flt_off[t]);
    h.fix_begin = static_cast<int32_t>(csr.fix_off[t]);
    h.n_flt = static_cast<int16_t>(csr.flt_off[t + 1] - csr.flt_off[t]);
    h.n_fix = static_cast<int16_t>(csr.fix_off[t + 1] - csr.fix_off[t]);
    h.fix_sign = static_cast<int8_t>(csr.fix_sign[t]);
    h.flt_sign = static_cast<int8_t>(csr.flt_sign[t]);
    h.pad = 0;
    out[t] = h;
}

// One thread per (row, slot) of a 32-slot row table; `piece_*` say which 32 coupons of which trade a row holds.
__global__ __launch_bounds__(256) void build_rows_kernel(CsrDev csr, RowBuildDev rb) {
    const int64_t at = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t r = at / kRowSlots;
    const int j = static_cast<int>(at % kRowSlots);
    if (r >= rb.rows) return;
    const int64_t t = rb.piece_trade[r];
    double tp = 0.0, ts = 0.0, al = 0.0, te = 0.0, w = 1.0, xtp = 0.0, xpay = 0.0;
    int64_t ml = 0, mf = 0;
    if (t >= 0) {
        const int64_t first = rb.piece_first ? rb.piece_first[r] : 0;
        const int64_t l0 = csr.flt_off[t] + first, f0 = csr.fix_off[t] + first;
        ml = min(max(csr.flt_off[t + 1] - l0, static_cast<int64_t>(0)), static_cast<int64_t>(kRowSlots));
        mf = min(max(csr.fix_off[t + 1] - f0, static_cast<int64_t>(0)), static_cast<int64_t>(kRowSlots));
        if (j < ml) {
            tp = csr.flt_tp[l0 + j]; ts = csr.flt_ts[l0 + j]; al = csr.flt_alpha[l0 + j];
            if (rb.row_te) te = csr.flt_te[l0 + j];
            if (rb.row_w) w = csr.flt_weight[l0 + j];
        }
        if (j < mf) { xtp = csr.fix_tp[f0 + j]; xpay = csr.fix_pay[f0 + j]; }
    }
    rb.row_tp[at] = tp; rb.row_ts[at] = ts; rb.row_alpha[at] = al; rb.row_xtp[at] = xtp; rb.row_xpay[at] = xpay;
    if (rb.row_te) rb.row_te[at] = te;
    if (rb.row_w) rb.row_w[at] = w;
    if (j == 0) {
        int32_t meta = (rb.piece_more && rb.piece_more[r]) ? 0x40000 : 0;
        double nn = 0.0, sp = 0.0;
        if (t >= 0) {
            nn = csr.notional[t]; sp = csr.spread[t];
            meta |= static_cast<int32_t>(ml | (mf << 8) | ((csr.flt_sign[t] < 0.0) ? 0x10000 : 0) |
                                         ((csr.fix_sign[t] < 0.0) ? 0x20000 : 0));
        }
        rb.row_notional[r] = nn; rb.row_spread[r] = sp; rb.row_meta[r] = meta;
        rb.row_trade[r] = static_cast<int32_t>(t);
    }
}

// One thread per (row, slot) of the lite table (16 slots = 15 coupons + a spare lane); row `r` belongs to trade slot
// `row_slot[r]` and is that trade's `row_piece[r]`-th row (coupons 15 piece .. 15 piece + 14).
__global__ __launch_bounds__(256) void build_lite_kernel(CsrDev csr, LiteBuildDev lb) {
    const int64_t at = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t r = at / kLiteSlots;
    const int j = static_cast<int>(at % kLiteSlots);
    if (r >= lb.rows) return;
    const int64_t slot = lb.row_slot[r];
    const int64_t t = slot >= 0 ? lb.slot_trade[slot] : -1;
    double tp = 0.0, ts = 0.0, al = 0.0, xtp = 0.0, xpay = 0.0, te = 0.0, w = 0.0;
    if (t >= 0 && j < kLiteCoupons) {
        const int64_t c = static_cast<int64_t>(lb.row_piece[r]) * kLiteCoupons + j;
        const int64_t ml = csr.flt_off[t + 1] - csr.flt_off[t], mf = csr.fix_off[t + 1] - csr.fix_off[t];
        if (c < ml) {
            const int64_t k = csr.flt_off[t] + c;
            tp = csr.flt_tp[k]; ts = csr.flt_ts[k]; al = csr.flt_alpha[k];
            if (lb.te_w) { te = csr.flt_te[k]; w = csr.flt_weight ? csr.flt_weight[k] : 1.0; }
        }
        if (c < mf) { const int64_t k = csr.fix_off[t] + c; xtp = csr.fix_tp[k]; xpay = csr.fix_pay[k]; }
    }
    lb.tp_ts[2 * at] = tp; lb.tp_ts[2 * at + 1] = ts;
    lb.al_xtp[2 * at] = al; lb.al_xtp[2 * at + 1] = xtp;
    lb.xpay[at] = xpay;
    if (lb.te_w) { lb.te_w[2 * at] = te; lb.te_w[2 * at + 1] = w; }
}

__global__ __launch_bounds__(256) void build_lite_slots_kernel(CsrDev csr, LiteBuildDev lb) {
    const int64_t s = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (s >= lb.n_slots) return;
    const int64_t t = lb.slot_trade[s];
    LiteTrade rec{0.0, 0.0, 0, -1, 0};
    if (t >= 0) {
        const int64_t ml = csr.flt_off[t + 1] - csr.flt_off[t], mf = csr.fix_off[t + 1] - csr.fix_off[t];
        rec.notional = csr.notional[t]; rec.spread = csr.spread[t];
        rec.trade = static_cast<int32_t>(t);
        rec.meta = static_cast<int32_t>(ml | (mf << 9) | ((csr.flt_sign[t] < 0.0) ? 0x40000 : 0) |
                                        ((csr.fix_sign[t] < 0.0) ? 0x80000 : 0));
    }
    lb.slot[s] = rec;
}

}  // namespace

hipError_t launch_build_headers(const CsrDev& csr, TradeHeader* out, hipStream_t stream) {
    if (csr.n == 0) return hipSuccess;
    hipLaunchKernelGGL(build_headers_kernel, dim3(static_cast<unsigned>((csr.n + 255) / 256)), dim3(256), 0, stream, csr, out);
    return hipGetLastError();
}

hipError_t launch_build_rows(const CsrDev& csr, const RowBuildDev& rb, hipStream_t stream) {
    if (rb.rows == 0) return hipSuccess;
    const int64_t threads = rb.rows * kRowSlots;
    hipLaunchKernelGGL(build_rows_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, stream, csr, rb);
    return hipGetLastError();
}

hipError_t launch_build_lite(const CsrDev& csr, const LiteBuildDev& lb, hipStream_t stream) {
    if (lb.rows > 0) {
        const int64_t threads = lb.rows * kLiteSlots;
        hipLaunchKernelGGL(build_lite_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, stream, csr, lb);
    }
    if (lb.n_slots > 0)
        hipLaunchKernelGGL(build_lite_slots_kernel, dim3(static_cast<unsigned>((lb.n_slots + 255) / 256)), dim3(256), 0, stream, csr, lb);
    return hipGetLastError();
}

}  // namespace adr
