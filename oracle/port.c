/*
 * port.c - batched C restatement of the reference's OIS valuation algorithm.  TEST INFRASTRUCTURE ONLY.
 *
 * Used (a) as the fast checker for large batches in tests/ and (b) as the "port" CPU baseline timed by
 * bench.py.  It is never linked into or called by the product package.  Parity unpinned in the
 * absolute sense (see oracle/cavour_oracle.py); this file is itself pinned against that torch.func
 * oracle by tests/test_oracle_port.py.
 *
 * It follows the reference's own formulation - knot-DF space, not the log space the HIP kernels use:
 *   D(t)                InterpolatorAd.simple_interpolate     cavour/market/curves/interpolator_ad.py:186-249
 *   fixed leg PV        Engine._price_fixed_leg_jax           cavour/market/position/engine.py:2414-2448
 *   float leg PV        Engine._float_leg_jax                 cavour/market/position/engine.py:2639-2728
 *   g = dPV/d dfs, H = d2PV/d dfs2, delta = (g . jac) 1e-4,
 *   gamma = (jac^T H jac + sum_k g_k hess_k) 1e-8             cavour/market/position/engine.py:2551-2568, 2909-2926
 * with the derivatives JAX obtains by AD written out analytically ("Closed forms").
 * Every cash flow is priced on its own (no merging of terms), H is kept dense over the knots the trade
 * touches, as the reference keeps it dense over all knots.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXLOC 1024 /* distinct knots one trade may touch */

typedef struct {
    int K, P, method;
    const double *x, *d, *jac, *hess;
} curve_t;

/* one discount factor: value, up to two knots, first and second partials w.r.t. those knots' DFs */
typedef struct {
    double v;
    int n;
    int k[2];
    double d1[2];
    double d2[2][2];
} df_t;

static int upper_bound(const double* x, int K, double t) {
    int lo = 0, hi = K;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (x[mid] > t) hi = mid; else lo = mid + 1;
    }
    return lo;
}

static void set_power(df_t* r, const curve_t* c, int ka, double aa, int kb, double ab) {
    /* D = d_a^aa * d_b^ab  (ab = 0: single knot) */
    const double da = c->d[ka];
    double D = exp(aa * log(da));
    r->n = 1; r->k[0] = ka;
    if (ab != 0.0 || kb >= 0) {
        const double db = c->d[kb];
        D = exp(aa * log(da) + ab * log(db));
        r->n = 2; r->k[1] = kb;
        r->d1[1] = ab * D / db;
        r->d2[1][1] = ab * (ab - 1.0) * D / (db * db);
        r->d2[0][1] = r->d2[1][0] = aa * ab * D / (da * db);
    }
    r->v = D;
    r->d1[0] = aa * D / da;
    r->d2[0][0] = aa * (aa - 1.0) * D / (da * da);
}

static void interpolate(const curve_t* c, double t, df_t* r) {
    const int K = c->K;
    const double* x = c->x;
    /* snap: nearest knot within 1e-10, first index on ties (argmin) */
    int j = upper_bound(x, K, t);
    int best = -1; double bd = 1e300;
    if (j > 0) { int b = j - 1; while (b > 0 && x[b - 1] == x[j - 1]) --b; best = b; bd = fabs(t - x[j - 1]); }
    if (j < K) { double dh = fabs(t - x[j]); if (dh < bd) { bd = dh; best = j; } }
    if (bd < 1e-10) {
        r->v = c->d[best]; r->n = 1; r->k[0] = best; r->d1[0] = 1.0; r->d2[0][0] = 0.0;
        return;
    }
    const double tau = t + 1e-12;
    const int lzr = c->method == 4;
    if (tau < x[0] || tau > x[K - 1]) {
        int k = tau < x[0] ? 0 : K - 1;
        set_power(r, c, k, lzr ? t / fmax(x[k], 1e-15) : 1.0, -1, 0.0);   /* LINEAR_FWD: the end knot's DF, flat */
        return;
    }
    int i = upper_bound(x, K, tau);
    if (i < 1) i = 1;
    if (i > K - 1) i = K - 1;
    const double dx = x[i] - x[i - 1];
    const double w = fabs(dx) <= 0x1p-104 ? 0.0 : (tau - x[i - 1]) / dx;
    if (c->method == 2) {        /* LINEAR_FWD_RATES (interpolator_ad.py:234-235): linear in the knot DFs themselves */
        r->v = c->d[i - 1] + w * (c->d[i] - c->d[i - 1]);
        r->n = 2; r->k[0] = i - 1; r->k[1] = i;
        r->d1[0] = 1.0 - w; r->d1[1] = w;
        r->d2[0][0] = r->d2[0][1] = r->d2[1][0] = r->d2[1][1] = 0.0;
        return;
    }
    if (lzr) set_power(r, c, i - 1, t * (1.0 - w) / fmax(x[i - 1], 1e-15), i, t * w / fmax(x[i], 1e-15));
    else set_power(r, c, i - 1, 1.0 - w, i, w);
}

typedef struct {
    int n;               /* local knots in use */
    int glob[MAXLOC];    /* local -> global knot */
    double g[MAXLOC];
    double* H;           /* [MAXLOC*MAXLOC], row stride MAXLOC */
    int* loc_of;         /* [K] global -> local, -1 */
} work_t;

static int local_of(work_t* w, int k) {
    int l = w->loc_of[k];
    if (l < 0) {
        l = w->n++;
        w->loc_of[k] = l;
        w->glob[l] = k;
        w->g[l] = 0.0;
        for (int i = 0; i <= l; ++i) { w->H[i * MAXLOC + l] = 0.0; w->H[l * MAXLOC + i] = 0.0; }
    }
    return l;
}

/* add a term f(X_0..X_{m-1}) with gradient fX and Hessian fXY to g and H */
static void add_term(work_t* w, int m, const df_t* X, const double* fX, const double* fXY /* m*m */) {
    int loc[4][2];
    for (int a = 0; a < m; ++a)
        for (int i = 0; i < X[a].n; ++i) loc[a][i] = local_of(w, X[a].k[i]);
    for (int a = 0; a < m; ++a)
        for (int i = 0; i < X[a].n; ++i) {
            w->g[loc[a][i]] += fX[a] * X[a].d1[i];
            for (int jx = 0; jx < X[a].n; ++jx) w->H[loc[a][i] * MAXLOC + loc[a][jx]] += fX[a] * X[a].d2[i][jx];
            for (int b = 0; b < m; ++b) {
                const double h = fXY[a * m + b];
                if (h == 0.0) continue;
                for (int jx = 0; jx < X[b].n; ++jx)
                    w->H[loc[a][i] * MAXLOC + loc[b][jx]] += h * X[a].d1[i] * X[b].d1[jx];
            }
        }
}

static double price_one(const curve_t* c, work_t* w, int mf, const double* ftp, const double* fpay, int ml,
                        const double* ltp, const double* lts, const double* lte, const double* lal,
                        const double* lw /* per-coupon weights or NULL */, double N,
                        double spread, double sf, double sl, double* delta, double* gamma, double* tmp /* MAXLOC*P */) {
    const int P = c->P;
    const double tv = 0.0;
    double pv = 0.0;
    w->n = 0;
    df_t V;
    interpolate(c, tv, &V);
    /* fixed leg: s * sum_j pay_j D(tp_j)/D(tv) [tp_j > tv]; principal = 0 */
    for (int j = 0; j < mf; ++j) {
        if (!(ftp[j] > tv)) continue;
        df_t X[2];
        interpolate(c, ftp[j], &X[0]);
        X[1] = V;
        const double cpn = sf * fpay[j];
        const double C = X[0].v, Vv = V.v;
        pv += cpn * C / Vv;
        const double fX[2] = {cpn / Vv, -cpn * C / (Vv * Vv)};
        const double fXY[4] = {0.0, -cpn / (Vv * Vv), -cpn / (Vv * Vv), 2.0 * cpn * C / (Vv * Vv * Vv)};
        add_term(w, 2, X, fX, fXY);
    }
    /* float leg: s * sum_j ((A/B - 1)/alpha + spread) alpha N * C/V [tp_j >= tv] */
    for (int j = 0; j < ml; ++j) {
        if (!(ltp[j] >= tv)) continue;
        const double al = lal[j];
        df_t X[4];
        interpolate(c, ltp[j], &X[2]);
        X[3] = V;
        const double C = X[2].v, Vv = V.v, R = C / Vv;
        if (al > 0.0) {
            interpolate(c, lts[j], &X[0]);
            interpolate(c, lte[j], &X[1]);
            const double A = X[0].v, B = X[1].v, n = sl * N * (lw ? lw[j] : 1.0);
            const double fwd = (A / B - 1.0) / al;
            const double cf = (fwd + spread) * al * n;
            pv += cf * R;
            const double fX[4] = {n / B * R, -n * A / (B * B) * R, cf / Vv, -cf * C / (Vv * Vv)};
            double fXY[16] = {0};
            fXY[0 * 4 + 1] = fXY[1 * 4 + 0] = -n / (B * B) * R;
            fXY[0 * 4 + 2] = fXY[2 * 4 + 0] = n / B / Vv;
            fXY[0 * 4 + 3] = fXY[3 * 4 + 0] = -n / B * C / (Vv * Vv);
            fXY[1 * 4 + 1] = 2.0 * n * A / (B * B * B) * R;
            fXY[1 * 4 + 2] = fXY[2 * 4 + 1] = -n * A / (B * B) / Vv;
            fXY[1 * 4 + 3] = fXY[3 * 4 + 1] = n * A / (B * B) * C / (Vv * Vv);
            fXY[2 * 4 + 3] = fXY[3 * 4 + 2] = -cf / (Vv * Vv);
            fXY[3 * 4 + 3] = 2.0 * cf * C / (Vv * Vv * Vv);
            add_term(w, 4, X, fX, fXY);
        } else {
            const double cf = spread * al * sl * N * (lw ? lw[j] : 1.0);   /* forward is forced to 0 when nothing accrues */
            pv += cf * R;
            df_t Y[2] = {X[2], V};
            const double fX[2] = {cf / Vv, -cf * C / (Vv * Vv)};
            const double fXY[4] = {0.0, -cf / (Vv * Vv), -cf / (Vv * Vv), 2.0 * cf * C / (Vv * Vv * Vv)};
            add_term(w, 2, Y, fX, fXY);
        }
    }
    /* chain rule to the pillars */
    const int m = w->n;
    if (delta) {
        for (int p = 0; p < P; ++p) delta[p] = 0.0;
        for (int a = 0; a < m; ++a) {
            const double* J = c->jac + (size_t)w->glob[a] * P;
            for (int p = 0; p < P; ++p) delta[p] += w->g[a] * J[p];
        }
        for (int p = 0; p < P; ++p) delta[p] *= 1e-4;
    }
    if (gamma) {
        for (int i = 0; i < P * P; ++i) gamma[i] = 0.0;
        /* tmp[a][q] = sum_b H[a][b] J[b][q] */
        for (int a = 0; a < m; ++a) {
            double* ta = tmp + (size_t)a * P;
            for (int q = 0; q < P; ++q) ta[q] = 0.0;
            for (int b = 0; b < m; ++b) {
                const double h = w->H[a * MAXLOC + b];
                if (h == 0.0) continue;
                const double* J = c->jac + (size_t)w->glob[b] * P;
                for (int q = 0; q < P; ++q) ta[q] += h * J[q];
            }
        }
        for (int a = 0; a < m; ++a) {
            const double* J = c->jac + (size_t)w->glob[a] * P;
            const double* ta = tmp + (size_t)a * P;
            const double* Ck = c->hess + (size_t)w->glob[a] * P * P;
            const double ga = w->g[a];
            for (int p = 0; p < P; ++p) {
                const double jp = J[p];
                double* grow = gamma + (size_t)p * P;
                const double* crow = Ck + (size_t)p * P;
                for (int q = 0; q < P; ++q) grow[q] += jp * ta[q] + ga * crow[q];
            }
        }
        for (int i = 0; i < P * P; ++i) gamma[i] *= 1e-8;
    }
    for (int a = 0; a < m; ++a) w->loc_of[w->glob[a]] = -1;
    return pv;
}

int adr_port_price_weighted(int K, int P, int method, const double* times, const double* dfs, const double* jac,
                            const double* hess, int64_t n, const int64_t* fix_off, const int64_t* flt_off,
                            const double* fix_tp, const double* fix_pay, const double* flt_tp, const double* flt_ts,
                            const double* flt_te, const double* flt_alpha, const double* flt_weight,
                            const double* notional, const double* spread, const double* fix_sign,
                            const double* flt_sign, double* pv, double* delta, double* gamma, int n_threads);

/* Returns 0, or -1 on bad arguments / a trade touching more than MAXLOC knots cannot occur silently:
 * the number of knots per trade is bounded by 2 + 6 * flows, checked up front. */
int adr_port_price(int K, int P, int method, const double* times, const double* dfs, const double* jac,
                   const double* hess, int64_t n, const int64_t* fix_off, const int64_t* flt_off,
                   const double* fix_tp, const double* fix_pay, const double* flt_tp, const double* flt_ts,
                   const double* flt_te, const double* flt_alpha, const double* notional, const double* spread,
                   const double* fix_sign, const double* flt_sign, double* pv, double* delta, double* gamma,
                   int n_threads) {
    return adr_port_price_weighted(K, P, method, times, dfs, jac, hess, n, fix_off, flt_off, fix_tp, fix_pay, flt_tp,
                                   flt_ts, flt_te, flt_alpha, NULL, notional, spread, fix_sign, flt_sign, pv, delta,
                                   gamma, n_threads);
}

/* The same with a weight per float coupon multiplying its notional (NULL = all 1): the cross-currency foreign leg,
 * whose coupons are discounted on another curve (engine.py:2639-2728 called with disc != index curve). */
int adr_port_price_weighted(int K, int P, int method, const double* times, const double* dfs, const double* jac,
                            const double* hess, int64_t n, const int64_t* fix_off, const int64_t* flt_off,
                            const double* fix_tp, const double* fix_pay, const double* flt_tp, const double* flt_ts,
                            const double* flt_te, const double* flt_alpha, const double* flt_weight,
                            const double* notional, const double* spread, const double* fix_sign,
                            const double* flt_sign, double* pv, double* delta, double* gamma, int n_threads) {
    if (K < 2 || P < 1 || (method != 1 && method != 2 && method != 4) || (gamma && !hess)) return -1;
    for (int64_t t = 0; t < n; ++t)
        if (2 + 2 * (fix_off[t + 1] - fix_off[t]) + 6 * (flt_off[t + 1] - flt_off[t]) > MAXLOC && K > MAXLOC)
            return -1;
    curve_t c = {K, P, method, times, dfs, jac, hess};
    int rc = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel
    {
        work_t w;
        w.H = (double*)malloc(sizeof(double) * MAXLOC * MAXLOC);
        w.loc_of = (int*)malloc(sizeof(int) * K);
        double* tmp = (double*)malloc(sizeof(double) * MAXLOC * P);
        if (!w.H || !w.loc_of || !tmp) {
#pragma omp atomic write
            rc = -1;
        } else {
            for (int k = 0; k < K; ++k) w.loc_of[k] = -1;
#pragma omp for schedule(dynamic, 64)
            for (int64_t t = 0; t < n; ++t) {
                const int64_t f0 = fix_off[t], l0 = flt_off[t];
                double v = price_one(&c, &w, (int)(fix_off[t + 1] - f0), fix_tp + f0, fix_pay + f0,
                                     (int)(flt_off[t + 1] - l0), flt_tp + l0, flt_ts + l0, flt_te + l0,
                                     flt_alpha + l0, flt_weight ? flt_weight + l0 : NULL, notional[t], spread[t],
                                     fix_sign[t], flt_sign[t],
                                     delta ? delta + (size_t)t * P : NULL,
                                     gamma ? gamma + (size_t)t * P * P : NULL, tmp);
                if (pv) pv[t] = v;
            }
        }
        free(w.H); free(w.loc_of); free(tmp);
    }
    return rc;
}

int adr_port_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
