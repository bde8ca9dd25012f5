/*
 * The drop-in boundary from plain C: upload a curve cache (times, dfs, jac, hess - what the reference's
 * Engine._cached_curve returns, cavour/market/position/engine.py:2362-2412), upload two trades in CSR form (the
 * arrays the engine extracts from the legs, :2519-2527, :2858-2877) and price VALUE / DELTA / GAMMA.
 *
 *   gcc -std=c99 -I include examples/c_abi_example.c -L adrates_amd -ladrates_hip -Wl,-rpath,$PWD/adrates_amd -o c_abi_example
 *
 * The curve is a toy: three knots (t = 0, 1, 2 years), two par rates r1, r2 with the bootstrap
 * d1 = 1/(1 + r1), d2 = (1 - r2 d1)/(1 + r2) and its exact first and second derivatives.
 */
#include <stdio.h>
#include <stdlib.h>

#include "adrates.h"

#define CHECK(call)                                                               \
    do {                                                                          \
        int rc_ = (call);                                                         \
        if (rc_ < 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, adr_last_error()); return 1; } \
    } while (0)

int main(void) {
    const double r1 = 0.04, r2 = 0.045;
    const double d1 = 1.0 / (1.0 + r1), d2 = (1.0 - r2 * d1) / (1.0 + r2);
    enum { K = 3, P = 2 };
    const double times[K] = {0.0, 1.0, 2.0};
    const double dfs[K] = {1.0, d1, d2};
    /* jac[k][p] = d dfs[k] / d r_p */
    const double dd1 = -d1 * d1;                                    /* d d1 / d r1 */
    const double d2_r1 = -r2 * dd1 / (1.0 + r2);                    /* d d2 / d r1 */
    const double d2_r2 = (-d1 * (1.0 + r2) - (1.0 - r2 * d1)) / ((1.0 + r2) * (1.0 + r2));
    const double jac[K * P] = {0.0, 0.0, dd1, 0.0, d2_r1, d2_r2};
    /* hess[k][p][q] */
    const double d1_11 = 2.0 * d1 * d1 * d1;
    const double d2_11 = -r2 * d1_11 / (1.0 + r2);
    const double d2_12 = -dd1 / ((1.0 + r2) * (1.0 + r2));
    const double d2_22 = 2.0 * (1.0 + d1) / ((1.0 + r2) * (1.0 + r2) * (1.0 + r2));
    const double hess[K * P * P] = {0, 0, 0, 0, d1_11, 0, 0, 0, d2_11, d2_12, d2_12, d2_22};

    /* two trades: a 2Y payer at 4.2 % on 10 M and a 1Y receiver at 3.9 % on 5 M, annual coupons */
    const int64_t fix_off[3] = {0, 2, 3}, flt_off[3] = {0, 2, 3};
    const double fix_tp[3] = {1.0, 2.0, 1.0};
    const double fix_pay[3] = {0.042 * 1e7, 0.042 * 1e7, 0.039 * 5e6};
    const double flt_tp[3] = {1.0, 2.0, 1.0}, flt_ts[3] = {0.0, 1.0, 0.0}, flt_te[3] = {1.0, 2.0, 1.0};
    const double flt_alpha[3] = {1.0, 1.0, 1.0};
    const double notional[2] = {1e7, 5e6}, spread[2] = {0.0, 0.0};
    const double fix_sign[2] = {-1.0, 1.0}, flt_sign[2] = {1.0, -1.0};

    adr_ctx* ctx = NULL;
    adr_curve* curve = NULL;
    adr_trades* trades = NULL;
    CHECK(adr_init(0, &ctx));
    CHECK(adr_curve_upload(ctx, ADR_INTERP_FLAT_FWD_RATES, K, P, times, dfs, jac, hess, &curve));
    CHECK(adr_trades_upload(ctx, 2, fix_off, flt_off, fix_tp, fix_pay, flt_tp, flt_ts, flt_te, flt_alpha, notional,
                            spread, fix_sign, flt_sign, &trades));
    double pv[2], delta[2 * P], gamma[2 * P * P], agg[1 + P + P * P];
    CHECK(adr_price(ctx, curve, trades, ADR_REQ_VALUE | ADR_REQ_DELTA | ADR_REQ_GAMMA, pv, delta, gamma, agg));
    for (int t = 0; t < 2; ++t) {
        printf("trade %d pv %.17g delta %.17g %.17g gamma %.17g %.17g %.17g %.17g\n", t, pv[t], delta[t * P],
               delta[t * P + 1], gamma[t * P * P], gamma[t * P * P + 1], gamma[t * P * P + 2], gamma[t * P * P + 3]);
    }
    printf("book pv %.17g delta %.17g %.17g\n", agg[0], agg[1], agg[2]);
    /* closed form for the first trade: PV = N [ (1 - d2) - c (d1 + d2) ] */
    printf("check pv0 %.17g\n", 1e7 * ((1.0 - d2) - 0.042 * (d1 + d2)));
    adr_free_trades(trades);
    adr_free_curve(curve);
    adr_free_ctx(ctx);
    return 0;
}
