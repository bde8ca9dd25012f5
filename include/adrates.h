/*
 * adrates.h - C-ABI of the MI355X-native OIS PV / delta / gamma path.
 *
 * The reference (ludcode/ADRates, "Cavour") is pure Python + JAX and has no FFI
 * of its own; the boundary this library replaces is the internal pure-function
 * seam of its valuation engine:
 *
 *   curve cache dict {times, dfs, jac, hess}   cavour/market/position/engine.py:2362-2412
 *   _price_fixed_leg_jax(dfs, times, interp, payment_times, payments, ...)     :2414-2448
 *   _float_leg_jax(dfs, times, interp, payment_times, start_times, end_times,
 *                  pay_alphas, spreads, notionals, ...)                         :2639-2728
 *   grad / hessian + chain rule to the pillar ladders                 :2541-2576, 2899-2934
 *   Portfolio.compute's running sums            cavour/market/portfolio/portfolio.py:39-66
 *
 * Conventions
 *   - every function returns 0 on success or a negative adr_status; the message
 *     for the calling thread's last failure is adr_last_error();
 *   - all arithmetic is IEEE float64; indices are int32/int64;
 *   - plain pointers and sizes only; "host" pointers are ordinary process memory,
 *     "_dev" pointers are HIP device memory of the ctx's GPU (e.g. a torch tensor's
 *     data_ptr);
 *   - a ctx is bound to one GPU and is not thread-safe; use one ctx per GPU/process;
 *   - there is no CPU fallback: adr_init fails when no HIP device is usable.
 */
#ifndef ADRATES_H
#define ADRATES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct adr_ctx adr_ctx;
typedef struct adr_curve adr_curve;
typedef struct adr_trades adr_trades;
typedef struct adr_curve_plan adr_curve_plan;
typedef struct adr_curve_set adr_curve_set;

typedef enum adr_status {
    ADR_OK = 0,
    ADR_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, unsorted times ...) */
    ADR_ERR_UNSUPPORTED = -2, /* interpolation method / pillar count outside what the kernels implement */
    ADR_ERR_HIP = -3,         /* a HIP runtime call failed; text has hipGetErrorString */
    ADR_ERR_NOMEM = -4,
    ADR_ERR_RCCL = -5
} adr_status;

/* Interpolation methods: values of the reference's InterpTypes enum
 * (cavour/utils/global_types.py:76-84) accepted by simple_interpolate
 * (cavour/market/curves/interpolator_ad.py:227-235). */
#define ADR_INTERP_FLAT_FWD_RATES 1
#define ADR_INTERP_LINEAR_FWD_RATES 2
#define ADR_INTERP_LINEAR_ZERO_RATES 4

/* Request mask bits: RequestTypes.VALUE / DELTA / GAMMA (cavour/utils/global_types.py:69-74). */
#define ADR_REQ_VALUE 1u
#define ADR_REQ_DELTA 2u
#define ADR_REQ_GAMMA 4u

/* Largest pillar count of an uploaded curve (the reference has no limit, cavour/market/position/engine.py:2388-2389); larger
 * curves are refused with ADR_ERR_UNSUPPORTED, and so is one whose knot tables do not fit the 160 KiB LDS of a CU next to two
 * 32-pillar tiles of its Jacobian.  Up to 32 pillars: the fast / lite kernels; 33-64: one launch for the whole ladder (wide
 * layout); 65-256: 32-pillar tiles, one launch per tile pair.  Results do not depend on the pillar count's parity or on which
 * kernel family a curve is routed to (DESIGN.md section 5 describes the routes and their measured rates).
 * adr_curve_plan_create (the device curve builder) takes up to ADR_MAX_PLAN_PILLARS. */
#define ADR_MAX_PILLARS 256
#define ADR_MAX_PLAN_PILLARS 64

/* Flags of adr_curve_upload_ex. */
#define ADR_CURVE_PILLAR_TILES 1u   /* curves of 33-64 pillars: price on 32-pillar tiles (one launch per tile pair) even when
                                       the single-launch layout fits the LDS; same results to rounding (diagnostics, A/B) */

int adr_version(void);
const char* adr_last_error(void);

/* One context per GPU: selects the device, creates the library's own stream and scratch. */
int adr_init(int device_ordinal, adr_ctx** out);
void adr_free_ctx(adr_ctx* ctx);
/* Number of HIP devices visible to the process (0 when none; never fails). */
int adr_device_count(void);

/*
 * Curve tables, replacing the reference's per-Engine cache dict
 * (engine.py:2405-2411): knot times [K] (non-decreasing, duplicates allowed and
 * meaningful), knot discount factors [K], jac = d dfs / d par-rates [K*P]
 * row-major, hess = d2 dfs / d par-rates2 [K*P*P] row-major (may be NULL when
 * gamma will never be requested).  The first knot must be the value-time point
 * the reference's grids start with (t = 0, discount factor 1, zero jac row): the
 * reference prices relative to D(0) (engine.py:2426-2435) and the kernels rely on
 * D(0) = 1 instead of dividing.  The library converts the tables to log space,
 * keeps only the knots a query can reference and uploads them.
 */
int adr_curve_upload(adr_ctx* ctx, int interp_method, int K, int P,
                     const double* times, const double* dfs,
                     const double* jac, const double* hess,
                     adr_curve** out);
/* The same with explicit layout flags (ADR_CURVE_*); adr_curve_upload is flags = 0. */
int adr_curve_upload_ex(adr_ctx* ctx, int interp_method, int K, int P,
                        const double* times, const double* dfs,
                        const double* jac, const double* hess,
                        uint32_t flags, adr_curve** out);
void adr_free_curve(adr_curve* curve);
int adr_curve_pillars(const adr_curve* curve);

/*
 * Host-side half of adr_curve_upload, exposed so the table construction can be
 * checked without a GPU: writes, for the Kc knots a query can reach,
 *   knot_index[Kc]  index into the caller's K knots,
 *   log_df[Kc]      ln dfs,
 *   lj[Kc*P]        d ln df_k / d r_p,
 *   lc[Kc*P*P]      d2 ln df_k / d r_p d r_q   (skipped when hess or lc is NULL).
 * Call with all outputs NULL to get Kc.  Returns Kc (>= 0) or a negative status.
 */
int adr_curve_tables_host(int K, int P, const double* times, const double* dfs,
                          const double* jac, const double* hess,
                          int32_t* knot_index, double* log_df, double* lj, double* lc);

/*
 * Diagnostic twin of adr_curve_tables_host: how the fast kernels would lay this curve out in LDS.
 * info[16] = { packed layout usable (0/1), core pillars Pc, core pairs Ec, packed entries Eu,
 *              entries per lane, core-table rows, short-end (mini) knots, LDS bytes of the gamma kernel,
 *              LDS bytes of the general kernel's variant with resident convexity rows (0: none), that variant fits (0/1),
 *              core slots per lane, hub layout found (0/1: the exact kernel variants; 0 = the universal ones),
 *              wide layout (33-64 pillars): 128-entry chunks per row of the packed triangle (7 / 10 / 17; 0: none),
 *              LDS bytes of the wide gamma kernel, the most chunks any knot's convexity row is read in, reserved }.
 * Returns 0 or a negative status.  No GPU needed.
 */
int adr_curve_layout_host(int K, int P, const double* times, const double* dfs,
                          const double* jac, const double* hess, int64_t* info);

/*
 * Book compilers, host side (multi-threaded, no GPU): the coupon schedules of many swap legs at once, and the foreign-leg
 * batches of a cross-currency book.  They replace the per-swap Python the reference runs before its leg functions:
 * Schedule._generate (cavour/utils/schedule.py:163-270) + Calendar.adjust / add_business_days
 * (cavour/utils/calendar.py:139-253) + SwapFloatLeg.generate_payment_dts (cavour/trades/rates/swap_float_leg.py:130-186),
 * and the coupon loop of Engine._compute_xccy (cavour/market/position/engine.py:1486-1578, 1640-1712).
 *
 * Dates are Excel serials (Date.excel_dt(), >= 1-Mar-1900).  Conventions covered: BACKWARD date generation without
 * end-of-month rolling, the WEEKEND calendar (weekend_calendar 0: NONE), bd_type = BusDayAdjustTypes value (1 NONE,
 * 2 FOLLOWING, 3 MODIFIED_FOLLOWING, 4 PRECEDING, 5 MODIFIED_PRECEDING), payment lags in business days, day counts with a
 * fixed denominator.  adr_leg_counts_host: coupons per leg.  adr_leg_times_host: with off = the exclusive prefix sums of
 * those counts ([n + 1]), payment / accrual start / accrual end times as year fractions from value_serial (payment times on
 * payment_denominator when > 0, else on the leg's own), accrual fractions, and plain[i] = 1 when the leg's dates are
 * strictly increasing (0: the reference's schedule de-duplication applies - the entries are then not its schedule and the
 * caller takes the object route for that leg).
 */
int adr_leg_counts_host(int64_t n, const int64_t* eff, const int64_t* term, const int64_t* months_per_period,
                        int64_t* n_coupons);
int adr_leg_times_host(int64_t n, const int64_t* eff, const int64_t* term, const int64_t* months_per_period,
                       const int64_t* payment_lag, int bd_type, int weekend_calendar, const double* denominator,
                       int64_t value_serial, double payment_denominator, const int64_t* off, double* tp, double* ts,
                       double* te, double* alpha, uint8_t* plain);
/*
 * Foreign leg of n cross-currency basis swaps (CSR for_off over the coupons: payment times on the XCCY curve's day count,
 * accrual start / end times, accrual fractions) given the discount factors the device returned for them (adr_curve_df):
 * df_x [m + 1] = D_x at the m payment times, then at the value time; df_f [2 m] = D_f at the m accrual starts, then at the m
 * accrual ends: (1) the rate-ladder batch - every live accruing coupon
 * with its discount factor as notional multiplier (adr_trades_upload_weighted's flt_weight) -, (2) the fixed flows
 * N (fwd + spread) alpha at tp > 0 followed by the notional exchanges (exch_t [n][2] effective / maturity times, exch_on [n])
 * at t > 0, and (3) pv_const [n] (in: the domestic constants; out: + flows dated at the value time, in domestic currency).
 * Output arrays are sized by the caller for every coupon (rates_*: for_off[n]; flows_*: for_off[n] + 2 n); the offsets
 * ([n + 1]) say what was filled.
 */
/* The notional exchanges of n legs (exch_t [n][2]: effective / maturity times; on [n]): flows -N, +N at t > 0 as CSR arrays
 * (off [n + 1]; flow_tp / flow_pay sized 2 n), flows dated AT the value time as pv_const [n] = sum sign * amount / scale. */
int adr_exchange_flows_host(int64_t n, const double* exch_t, const double* notional, const uint8_t* on, const double* sign,
                            double scale, int64_t* off, double* flow_tp, double* flow_pay, double* pv_const);
int adr_xccy_assemble_host(int64_t n, const int64_t* for_off, const double* tp_x, const double* ts, const double* te,
                           const double* alpha, const double* df_x, const double* df_f, const double* for_n,
                           const double* for_spread, const double* for_sign, double spot, const double* exch_t,
                           const uint8_t* exch_on, int64_t* rates_off, double* rates_ts, double* rates_te,
                           double* rates_alpha, double* rates_weight, int64_t* flows_off, double* flows_tp,
                           double* flows_pay, double* pv_const);

/*
 * Curve builder on the device, for batches of par-rate scenarios on one knot grid.  It replaces
 * Engine.build_curve_ad (engine.py:2246-2360: the lax.scan d = (1 - r PV01_prev) / (1 + r acc)) and the
 * jacrev / hessian of Engine._cached_curve (engine.py:2388-2389) for the case the reference handles by
 * rebuilding a Model per shock (Model.scenario, cavour/models/models.py:507-557): the par rates change,
 * the schedules - hence the knot grid - do not.
 *
 * A plan holds what does not depend on the rates: the scan description of the K sorted bootstrap points
 * (engine.py:2283-2334) -
 *   times[K]     knot times (as for adr_curve_upload),
 *   acc[K]       accrual fraction of the coupon period ending at the knot (0 for the t = 0 point),
 *   pillar[K]    index of the calibration swap whose par rate the knot uses,
 *   prev_idx[K]  knot whose PV01 the knot builds on (first sorted point with the previous coupon's
 *                round(t, 2) key), -1 for a swap's first period,
 * and the table layout, taken from the base curve (base_dfs/base_jac/base_hess as for adr_curve_upload;
 * base_hess NULL = no gamma for any curve of the plan).  Curves built from a plan point into it: free
 * the sets before the plan.
 */
int adr_curve_plan_create(adr_ctx* ctx, int interp_method, int K, int P,
                          const double* times, const double* acc,
                          const int32_t* pillar, const int32_t* prev_idx,
                          const double* base_dfs, const double* base_jac, const double* base_hess,
                          adr_curve_plan** out);
void adr_free_curve_plan(adr_curve_plan* plan);

/*
 * Bootstrap n_scen curves (rates: host array [n_scen * P] of par rates, decimal) with their first and -
 * when the plan has hess - second par-rate derivatives on the GPU and convert them to the kernels' tables.
 * Blocks until the curves are ready.  adr_curve_set_get returns a curve owned by the set (do not free it)
 * that adr_price / adr_price_dev accept like an uploaded one.
 */
int adr_curve_set_build(adr_ctx* ctx, const adr_curve_plan* plan, int n_scen, const double* rates,
                        adr_curve_set** out);
int adr_curve_set_size(const adr_curve_set* set);
const adr_curve* adr_curve_set_get(const adr_curve_set* set, int i);
/* Copy scenario i's dense arrays back (any pointer may be NULL): dfs[K], jac[K*P], hess[K*P*P] - the
 * contents of the reference's cache dict for that scenario. */
int adr_curve_set_download(const adr_curve_set* set, int i, double* dfs, double* jac, double* hess);
void adr_free_curve_set(adr_curve_set* set);

/*
 * A batch of OIS trades in CSR form - the per-trade arrays the reference engine
 * extracts from the leg objects (engine.py:2519-2527 fixed, :2858-2877 float):
 *   fix_off/flt_off [n+1]  offsets into the cash-flow arrays (fix_off[0] = flt_off[0] = 0)
 *   fix_tp, fix_pay        fixed payment times (years from the value date) and amounts
 *   flt_tp/ts/te/alpha     float payment, accrual-start, accrual-end times and accrual fractions
 *   notional, spread       per trade (float-leg notional and spread)
 *   fix_sign, flt_sign     +1 receive / -1 pay, per trade
 * Value time is 0 and both principals are 0, as for every OIS the reference builds
 * (cavour/trades/rates/ois.py:149, swap_float_leg.py:106).
 * The arrays are validated and classified on the host (a thread per contiguous trade range), copied to the
 * device once, and the kernels' padded row tables are gathered from them ON THE DEVICE (trades_build.hip):
 * about 40 ms per million benchmark trades.  Blocks until the batch is usable; may be called from several
 * host threads on one ctx (it touches no shared state of the ctx but its stream).
 */
int adr_trades_upload(adr_ctx* ctx, int64_t n_trades,
                      const int64_t* fix_off, const int64_t* flt_off,
                      const double* fix_tp, const double* fix_pay,
                      const double* flt_tp, const double* flt_ts,
                      const double* flt_te, const double* flt_alpha,
                      const double* notional, const double* spread,
                      const double* fix_sign, const double* flt_sign,
                      adr_trades** out);
/*
 * The same with a weight per float coupon (flt_weight[sum n_flt], NULL = all 1) that multiplies the trade's
 * notional for that coupon.  It carries the discount factor of the *other* curve when a float leg is
 * discounted on one curve and projected on another - the foreign leg of a cross-currency swap, where the
 * reference calls _float_leg_jax with disc != index curve (engine.py:1640-1712): holding the XCCY curve fixed,
 * the sensitivities to the foreign OIS rates are those of sum_j w_j N D(ts_j)/D(te_j) with w_j = D_x(tp_j).
 * Trades with a weight != 1 are priced like payment-lag trades: the payment-lag rows of the lite kernel (PV / DELTA),
 * the payment-lag variant of the fast kernel (GAMMA, curves with the packed layout), the general kernel otherwise.
 */
int adr_trades_upload_weighted(adr_ctx* ctx, int64_t n_trades,
                               const int64_t* fix_off, const int64_t* flt_off,
                               const double* fix_tp, const double* fix_pay,
                               const double* flt_tp, const double* flt_ts,
                               const double* flt_te, const double* flt_alpha,
                               const double* flt_weight,
                               const double* notional, const double* spread,
                               const double* fix_sign, const double* flt_sign,
                               adr_trades** out);
void adr_free_trades(adr_trades* trades);
int64_t adr_trades_count(const adr_trades* trades);
/* Bytes of trade input one pricing pass has to read (SURVEY.md section 8(d):
 * 16 per fixed flow + 32 per float flow + 40 per trade). */
int64_t adr_trades_input_bytes(const adr_trades* trades);

/*
 * Price the batch: per-trade PV [n], delta ladder [n*P] and gamma [n*P*P]
 * (row-major, full symmetric matrix), units as the reference: delta per 1 bp
 * (x1e-4), gamma per bp^2 (x1e-8).  Any output may be NULL; req_mask says what
 * to compute.  agg (optional) receives the portfolio sums laid out as
 * [pv, delta[P], gamma[P*P]] = 1 + P + P*P doubles - what Portfolio.compute
 * returns.  Blocks until the results are in the (host) buffers.
 */
int adr_price(adr_ctx* ctx, const adr_curve* curve, const adr_trades* trades,
              uint32_t req_mask,
              double* pv, double* delta, double* gamma, double* agg);

/*
 * Same, with device-resident outputs and no host synchronisation: the kernels
 * are enqueued on `stream` (a hipStream_t; NULL = the ctx's own stream) and the
 * call returns immediately.  Output pointers are device memory owned by the
 * caller.  This is the entry the throughput benchmark times.  The call neither
 * allocates nor synchronises, so a sequence of them (a scenario ladder, the pieces
 * of a cross-currency book) can be captured on `stream` into a HIP graph and replayed.
 * Stream rules: (1) a call with agg_dev != NULL stages its per-block partial sums in scratch
 * owned by the ctx, so all aggregate-producing calls of one ctx must be ordered on ONE
 * stream (or separated by a synchronisation); use one ctx per stream for concurrent aggregates.
 * (2) A batch that holds payment-lag or weighted coupons owns a per-wave scratch used by GAMMA requests:
 * calls with GAMMA on the SAME adr_trades must be stream-ordered.  Everything else - different batches on
 * different streams, calls without agg_dev - may run concurrently on one ctx.
 */
int adr_price_dev(adr_ctx* ctx, const adr_curve* curve, const adr_trades* trades,
                  uint32_t req_mask,
                  double* pv_dev, double* delta_dev, double* gamma_dev, double* agg_dev,
                  void* stream);

/*
 * The foreign leg of a book of cross-currency swaps on TWO curves, one launch: Engine._compute_xccy's second leg call
 * (cavour/market/position/engine.py:1640-1733 - _float_leg_jax with the foreign OIS curve as index curve and the XCCY curve
 * as discount curve) with its two first-order ladders.  `legs` is an ordinary batch (adr_trades_upload): per swap the
 * foreign float coupons with flt_tp = payment times in the XCCY curve's day count, flt_ts / flt_te / flt_alpha in the
 * leg's own, the notional exchanges as fixed flows (times in the XCCY curve's day count), notional, spread and signs in
 * FOREIGN currency (the caller converts the results with 1 / spot, as the reference does at :1713, :1733).  A coupon is
 * N ((D_f(ts) / D_f(te) - 1) + spread alpha) D_x(tp): pv [n]; delta_foreign [n * P_f] = d pv / d (foreign par rates) with the
 * XCCY curve held fixed (:1702-1712); delta_basis [n * P_x] = d pv / d (basis spreads); per bp.  agg_foreign / agg_basis: the
 * book sums in adr_price's layout ([pv, delta[P], zeros]; the PV total sits in agg_foreign[0]).  VALUE / DELTA only
 * (ADR_ERR_UNSUPPORTED with GAMMA: use three batches - adr_trades_upload_weighted - and adr_price); both curves up to 32
 * pillars, both on LINEAR_FWD_RATES or both on a log-linear scheme; every leg at most 390 coupons with some accrual end != payment time.
 * The _dev form enqueues on `stream` and neither allocates nor synchronises.
 */
int adr_price_xccy_foreign(adr_ctx* ctx, const adr_curve* foreign_curve, const adr_curve* xccy_curve, const adr_trades* legs,
                           uint32_t req_mask, double* pv, double* delta_foreign, double* delta_basis, double* agg_foreign,
                           double* agg_basis);
int adr_price_xccy_foreign_dev(adr_ctx* ctx, const adr_curve* foreign_curve, const adr_curve* xccy_curve, const adr_trades* legs,
                               uint32_t req_mask, double* pv_dev, double* delta_foreign_dev, double* delta_basis_dev,
                               double* agg_foreign_dev, double* agg_basis_dev, void* stream);

/*
 * Host-side half of adr_price_dev's routing, exposed so that it can be checked without a GPU (like adr_curve_layout_host):
 * the launch plan for a curve (arguments as adr_curve_upload_ex) and a batch (the arrays of adr_trades_upload_weighted the
 * classification reads) under a request - req_mask, per_trade != 0: some per-trade output is wanted, aggregate != 0: agg is
 * wanted - on a device of n_cu compute units.  launches [max_launches][4] receives {kernel family, trade set, items,
 * blocks} per launch (enums of adrates_amd/csrc/route.hpp), cover [n] how many launches price each trade (the tile
 * launches of one pass count once): the library's contract is cover[i] == 1 for every trade.  Returns the number of
 * launches (possibly > max_launches) or a negative status.
 */
int adr_route_host(int interp_method, int K, int P, const double* times, const double* dfs, const double* jac, const double* hess,
                   uint32_t curve_flags, int64_t n, const int64_t* fix_off, const int64_t* flt_off, const double* flt_tp,
                   const double* flt_te, const double* flt_alpha, const double* flt_weight, uint32_t req_mask, int per_trade,
                   int aggregate, int n_cu, int32_t* cover, int32_t* launches, int max_launches);

/*
 * Discount factors at n query times off an uploaded curve: InterpolatorAd.simple_interpolate evaluated on the GPU
 * (cavour/market/curves/interpolator_ad.py:186-249; same snap / + 1e-12 / duplicate-knot semantics as the pricing
 * kernels, all three schemes).  Replaces the reference's df lookups inside the cross-currency leg function
 * (cavour/market/position/engine.py:1640-1712: D_x(tp_j), D_f(ts_j), D_f(te_j)).  adr_curve_df takes and fills host
 * arrays and blocks; adr_curve_df_dev takes device arrays and enqueues on `stream` (NULL = the ctx's own).
 */
int adr_curve_df(adr_ctx* ctx, const adr_curve* curve, int64_t n, const double* t, double* df);
int adr_curve_df_dev(adr_ctx* ctx, const adr_curve* curve, int64_t n, const double* t_dev, double* df_dev, void* stream);

/* Wait for everything enqueued on the ctx's own stream. */
int adr_sync(adr_ctx* ctx);

/*
 * Sum the aggregate ladder over the ranks of an RCCL communicator (one rank per
 * GPU): in-place ncclAllReduce(sum, double) of `count` doubles at agg_dev on
 * `stream`.  `rccl_comm` is an ncclComm_t.  This is the only exchange step of
 * the multi-GPU path; per-trade results never leave their GPU.
 */
int adr_allreduce_agg(adr_ctx* ctx, void* rccl_comm, double* agg_dev, int count, void* stream);

/*
 * The communicator for adr_allreduce_agg, for hosts that have no RCCL binding of their own (one rank per GPU / process):
 * rank 0 draws a unique id (adr_rccl_unique_id: ADR_RCCL_ID_BYTES bytes, = ncclGetUniqueId), hands it to the other ranks
 * by whatever channel the host has (a file, a socket, torch.distributed's store), and every rank calls adr_rccl_comm_init
 * with the same id (= ncclCommInitRank on the ctx's GPU; blocks until all n_ranks have called it).  The handle is an
 * ncclComm_t; free it with adr_rccl_comm_destroy before the ctx.
 */
#define ADR_RCCL_ID_BYTES 128
int adr_rccl_unique_id(void* id_out);
int adr_rccl_comm_init(adr_ctx* ctx, const void* id, int n_ranks, int rank, void** comm_out);
void adr_rccl_comm_destroy(void* rccl_comm);

#ifdef __cplusplus
}
#endif
#endif /* ADRATES_H */
