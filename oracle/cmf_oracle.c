/*
 * cmf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Literal fp64 CPU restatement of the convolutive-NMF multiplicative-update
 * (MU) path of degleris1/CMF.jl.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this file; the product path
 * (cmf.jl_amd/) never does.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference has no assertions, golden
 * vectors or fixtures for this path (test/test.jl has the MU line commented
 * out), and Julia is not available in the build container, so this oracle
 * is pinned only by (i) an independently written numpy restatement
 * (oracle/cmf_oracle.py), (ii) an index-level brute-force restatement and
 * algebraic property tests in tests/, and (iii) the committed fixtures in
 * tests/golden/ generated from it.
 *
 * Layouts are Julia's (column-major, first index fastest):
 *   data[n + N*t]            N x T
 *   W[k + K*(n + N*l)]       K x N x L
 *   H[k + K*t]               K x T
 *
 * Each function cites the reference lines it restates
 * (paths relative to the reference checkout).
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define CMF_EPS 2.220446049250313e-16 /* eps(Float64): src/CMF.jl:20, mult.jl:37-38 */

/* ------------------------------------------------------------------ */
/* Portable counter-based RNG (spec shared with the product's own     */
/* implementation in cmf.jl_amd/csrc/cmf_rng.h; written separately).  */
/* ------------------------------------------------------------------ */
static inline uint64_t rng_f(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
static inline uint64_t rng_base(uint64_t seed, uint64_t stream)
{
    return rng_f(seed + 0x632BE59BD9B4E019ULL * (stream + 1));
}
static inline uint64_t rng_bits(uint64_t base, uint64_t i)
{
    return rng_f(base + (i + 1) * 0x9E3779B97F4A7C15ULL);
}
/* uniform in [0,1) */
static inline double rng_u01(uint64_t base, uint64_t i)
{
    return (double)(rng_bits(base, i) >> 11) * 0x1.0p-53;
}
/* uniform in (0,1] */
static inline double rng_u01_open0(uint64_t base, uint64_t i)
{
    return (double)((rng_bits(base, i) >> 11) + 1) * 0x1.0p-53;
}
/* standard normal: Box-Muller, cosine branch, two sub-streams */
static inline double rng_normal(uint64_t base_a, uint64_t base_b, uint64_t i)
{
    double u1 = rng_u01_open0(base_a, i);
    double u2 = rng_u01(base_b, i);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

double oracle_rng_u01(uint64_t seed, uint64_t stream, uint64_t i)
{
    return rng_u01(rng_base(seed, stream), i);
}
double oracle_rng_normal(uint64_t seed, uint64_t stream, uint64_t i)
{
    return rng_normal(rng_base(seed, stream), rng_base(seed, stream + 1), i);
}

/* Gamma(shape a, scale 1), Marsaglia-Tsang with the a<1 boost.  Sample j,
 * attempt t uses counters 4*(64*j + t) + {0,1,2}; counter 4*64*j+3 of the
 * first attempt slot holds the boost uniform. */
static double rng_gamma(uint64_t bn_a, uint64_t bn_b, uint64_t bu, uint64_t j, double a)
{
    double boost = 1.0;
    if (a < 1.0) {
        double u = rng_u01_open0(bu, 4 * (64 * j) + 3);
        boost = pow(u, 1.0 / a);
        a += 1.0;
    }
    double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint64_t t = 0; t < 64; ++t) {
        uint64_t ctr = 4 * (64 * j + t);
        double x = rng_normal(bn_a, bn_b, ctr);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        double u = rng_u01_open0(bu, ctr + 1);
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return boost * d * v;
    }
    return boost * d; /* unreachable in practice */
}

/* ------------------------------------------------------------------ */
/* Convolution primitives                                             */
/* ------------------------------------------------------------------ */

/* tensor_conv!(est, W, H): src/common.jl:24-34 (+ s_dot! :108-118).
 * est = 0; for lag: est[:, lag+1:T] += W[:,:,lag+1]' * H[:, 1:T-lag].
 * Kept per-lag (lag outermost) like the reference so the accumulation order
 * over lags is the reference's. */
void oracle_tensor_conv(int64_t N, int64_t T, int64_t K, int64_t L,
                        const double *W, const double *H, double *est)
{
    memset(est, 0, sizeof(double) * (size_t)N * (size_t)T);
    for (int64_t lag = 0; lag < L && lag < T; ++lag) {
        const double *Wl = W + (size_t)K * N * lag;
#pragma omp parallel for schedule(static) if (N * K * (T - lag) > 2000000)
        for (int64_t t = lag; t < T; ++t) {
            const double *h = H + (size_t)K * (t - lag);
            double *e = est + (size_t)N * t;
            for (int64_t n = 0; n < N; ++n) {
                const double *w = Wl + (size_t)K * n;
                double s = 0.0;
                for (int64_t k = 0; k < K; ++k) s += w[k] * h[k];
                e[n] += s;
            }
        }
    }
}

/* tensor_transconv!(out, W, X): src/common.jl:71-81.
 * out = 0; for lag: out[:, 1:T-lag] += W[:,:,lag+1] * X[:, 1+lag:T]. */
void oracle_tensor_transconv(int64_t N, int64_t T, int64_t K, int64_t L,
                             const double *W, const double *X, double *out)
{
    memset(out, 0, sizeof(double) * (size_t)K * (size_t)T);
    for (int64_t lag = 0; lag < L && lag < T; ++lag) {
        const double *Wl = W + (size_t)K * N * lag;
#pragma omp parallel for schedule(static) if (N * K * (T - lag) > 2000000)
        for (int64_t t = 0; t < T - lag; ++t) {
            const double *x = X + (size_t)N * (t + lag);
            double *o = out + (size_t)K * t;
            for (int64_t n = 0; n < N; ++n) {
                const double *w = Wl + (size_t)K * n;
                double xv = x[n];
                for (int64_t k = 0; k < K; ++k) o[k] += w[k] * xv;
            }
        }
    }
}

static double frob_norm(const double *x, size_t n)
{
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static) if (n > 2000000)
    for (int64_t i = 0; i < (int64_t)n; ++i) s += x[i] * x[i];
    return sqrt(s);
}

double oracle_norm(const double *x, int64_t n) { return frob_norm(x, (size_t)n); }

/* compute_loss / compute_resids: src/common.jl:54-59.
 * scratch: N*T doubles. */
double oracle_compute_loss(int64_t N, int64_t T, int64_t K, int64_t L,
                           const double *data, const double *W, const double *H,
                           double *scratch)
{
    size_t NT = (size_t)N * T;
    oracle_tensor_conv(N, T, K, L, W, H, scratch);
    for (size_t i = 0; i < NT; ++i) scratch[i] -= data[i];
    return frob_norm(scratch, NT) / frob_norm(data, NT);
}

/* The per-lag H_shift * X' contraction of update_motifs!: src/algs/mult.jl:31-34.
 * out[:, :, lag+1] = shift_cols(H, lag) * X[:, 1+lag:T]'
 *   out[k,n,lag] = sum_{t=0}^{T-lag-1} H[k,t] * X[n,t+lag]. */
void oracle_hxt(int64_t N, int64_t T, int64_t K, int64_t L,
                const double *H, const double *X, double *out)
{
    memset(out, 0, sizeof(double) * (size_t)K * N * L);
#pragma omp parallel for schedule(dynamic) if (N * K * T > 2000000)
    for (int64_t lag = 0; lag < L; ++lag) {
        double *o = out + (size_t)K * N * lag;
        for (int64_t t = 0; t < T - lag; ++t) {
            const double *h = H + (size_t)K * t;
            const double *x = X + (size_t)N * (t + lag);
            for (int64_t n = 0; n < N; ++n) {
                double xv = x[n];
                double *on = o + (size_t)K * n;
                for (int64_t k = 0; k < K; ++k) on[k] += h[k] * xv;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* MU rule: src/algs/mult.jl                                           */
/* ------------------------------------------------------------------ */

/* update_motifs!(rule::MultUpdate, data, W, H; l1W, l2W): mult.jl:23-39.
 * est (N*T), numW, denomW (K*N*L) are the rule's scratch fields. */
void oracle_update_motifs(int64_t N, int64_t T, int64_t K, int64_t L,
                          const double *data, double *W, const double *H,
                          double *est, double *numW, double *denomW,
                          double l1W, double l2W)
{
    oracle_tensor_conv(N, T, K, L, W, H, est);       /* :28 */
    oracle_hxt(N, T, K, L, H, data, numW);            /* :32 */
    oracle_hxt(N, T, K, L, H, est, denomW);           /* :33 */
    size_t n = (size_t)K * N * L;
    for (size_t i = 0; i < n; ++i) {
        /* :37  W *= numW / (denomW + l1W + 2*l2W*W + eps())
         * Julia parses a+b+c+d as +(a,b,c,d) -> left fold; 2*l2W*W -> (2*l2W)*W */
        double den = ((denomW[i] + l1W) + (2.0 * l2W) * W[i]) + CMF_EPS;
        double w = W[i] * (numW[i] / den);
        W[i] = w > CMF_EPS ? w : CMF_EPS;             /* :38 max(eps(), W) */
        if (w != w) W[i] = w;                         /* Julia max propagates NaN */
    }
}

/* update_feature_maps!(rule::MultUpdate, data, W, H; l1H, l2H): mult.jl:42-58.
 * returns norm(est - data) / data_norm. */
double oracle_update_feature_maps(int64_t N, int64_t T, int64_t K, int64_t L,
                                  const double *data, const double *W, double *H,
                                  double *est, double *numH, double *denomH,
                                  double data_norm, double l1H, double l2H)
{
    oracle_tensor_conv(N, T, K, L, W, H, est);            /* :44 */
    oracle_tensor_transconv(N, T, K, L, W, data, numH);   /* :47 */
    oracle_tensor_transconv(N, T, K, L, W, est, denomH);  /* :48 */
    size_t n = (size_t)K * T;
    for (size_t i = 0; i < n; ++i) {
        double den = ((denomH[i] + l1H) + (2.0 * l2H) * H[i]) + CMF_EPS; /* :51 */
        double h = H[i] * (numH[i] / den);
        H[i] = h > CMF_EPS ? h : CMF_EPS;                                /* :52 */
        if (h != h) H[i] = h;
    }
    oracle_tensor_conv(N, T, K, L, W, H, est);            /* :55 */
    size_t NT = (size_t)N * T;
    for (size_t i = 0; i < NT; ++i) est[i] -= data[i];    /* :56 resids */
    return frob_norm(est, NT) / data_norm;                /* :57 */
}

/* converged(loss_hist, patience, tol): src/model.jl:91-107. */
int oracle_converged(const double *loss_hist, int64_t len, int64_t patience, double tol)
{
    if (len <= patience) return 0;
    /* d_loss = diff(loss_hist[end-patience:end]) -> `patience` differences */
    for (int64_t i = len - patience; i < len; ++i)
        if (!(fabs(loss_hist[i] - loss_hist[i - 1]) < tol)) return 0;
    return 1;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* fit(alg::AlternatingOptimizer, data, L, K, W_init, H_init; kwargs...):
 * src/algs/alternating.jl:16-71 with update_rule = MultUpdate
 * (src/algs/mult.jl:11-20 ctor).  W, H are updated in place (the caller
 * passes copies: alternating.jl:33-34 deepcopy).  loss_hist/time_hist need
 * max_itr+1 entries; *n_hist receives the number written.
 * Returns 1 if it stopped on the convergence test (:63-66), else 0. */
int oracle_fit_mult(int64_t N, int64_t T, int64_t K, int64_t L,
                    const double *data, double *W, double *H,
                    int64_t max_itr, double max_time,
                    int check_convergence, int64_t patience, double tol, int eval_mode,
                    double l1W, double l2W, double l1H, double l2H,
                    double *loss_hist, double *time_hist, int64_t *n_hist)
{
    size_t NT = (size_t)N * T, KNL = (size_t)K * N * L, KT = (size_t)K * T;
    double *est = (double *)malloc(sizeof(double) * NT);
    double *numW = (double *)calloc(KNL, sizeof(double));
    double *denomW = (double *)calloc(KNL, sizeof(double));
    double *numH = (double *)calloc(KT, sizeof(double));
    double *denomH = (double *)calloc(KT, sizeof(double));
    double data_norm = frob_norm(data, NT);           /* mult.jl:13 */
    int stopped = 0;
    int64_t len = 0;

    loss_hist[len] = oracle_compute_loss(N, T, K, L, data, W, H, est); /* :37 */
    time_hist[len] = 0.0;                                              /* :38 */
    ++len;

    int64_t itr = 1;
    while (itr <= max_itr && time_hist[len - 1] <= max_time) {         /* :45 */
        itr += 1;
        double t0 = now_s();
        if (!eval_mode)                                                /* :51-53 */
            oracle_update_motifs(N, T, K, L, data, W, H, est, numW, denomW, l1W, l2W);
        double loss = oracle_update_feature_maps(N, T, K, L, data, W, H, est,
                                                 numH, denomH, data_norm, l1H, l2H);
        double dur = now_s() - t0;
        time_hist[len] = time_hist[len - 1] + dur;                     /* :58 */
        loss_hist[len] = loss;                                         /* :59 */
        ++len;
        if (check_convergence && oracle_converged(loss_hist, len, patience, tol)) { /* :63 */
            stopped = 1;
            break;
        }
    }
    *n_hist = len;
    free(est); free(numW); free(denomW); free(numH); free(denomH);
    return stopped;
}

/* ------------------------------------------------------------------ */
/* HALS rule (BASELINE config 5): src/algs/hals.jl                      */
/* resids (N*T) is the rule's state, carried across iterations.        */
/* ------------------------------------------------------------------ */

/* update_motifs!(::HALSUpdate): hals.jl:31-34, :53-61, :90-112.
 * Row ind = K*lag + k of H_unfold (common.jl:133-142) is H[k, :] shifted right by lag. */
void oracle_hals_update_motifs(int64_t N, int64_t T, int64_t K, int64_t L,
                               double *W, const double *H, double *resids, double l1W, double l2W)
{
    for (int64_t k = 0; k < K; ++k)          /* :92 k outer */
        for (int64_t l = 0; l < L; ++l) {    /* :93 lag inner */
            double nrm2 = 0.0;               /* H_norms[ind]^2, :57-60 */
            for (int64_t t = l; t < T; ++t) { double h = H[k + K * (t - l)]; nrm2 += h * h; }
            double nrm = sqrt(nrm2);
            double *w = W + k + K * N * l;   /* W[k, n, l] at stride K over n */
            double den = nrm * nrm + CMF_EPS + l2W;
#pragma omp parallel for schedule(static) if (N * (T - l) > 2000000)
            for (int64_t n = 0; n < N; ++n) {
                double wo = w[K * n], dot = 0.0;
                /* :104 resids -= w_old * h' ; then :110 dot = (resids * h)[n] */
                for (int64_t t = l; t < T; ++t) {
                    double h = H[k + K * (t - l)];
                    double r = resids[n + N * t] - wo * h;
                    resids[n + N * t] = r;
                    dot += r * h;
                }
                double wn = (-dot - l1W) / den;
                wn = wn > 0.0 ? wn : 0.0;
                w[K * n] = wn;
                for (int64_t t = l; t < T; ++t)   /* :106 */
                    resids[n + N * t] += wn * H[k + K * (t - l)];
            }
        }
}

/* update_feature_maps!(::HALSUpdate): hals.jl:37-42, :64-80, :121-154.  Returns the loss. */
double oracle_hals_update_feature_maps(int64_t N, int64_t T, int64_t K, int64_t L,
                                       const double *W, double *H, double *resids,
                                       double data_norm, double l1H, double l2H)
{
    double *wn2 = (double *)calloc((size_t)K * L, sizeof(double)); /* W_norms[k,l]^2, :67-72 */
    for (int64_t l = 0; l < L; ++l)
        for (int64_t n = 0; n < N; ++n)
            for (int64_t k = 0; k < K; ++k) {
                double v = W[k + K * (n + N * l)];
                wn2[k + K * l] += v * v;
            }
    for (int64_t k = 0; k < K; ++k)          /* :124 */
        for (int64_t t = 0; t < T; ++t) {    /* :125 */
            int64_t Lt = (T - t < L) ? (T - t) : L;   /* :136 */
            double nrm2 = 0.0;
            for (int64_t l = 0; l < Lt; ++l) nrm2 += wn2[k + K * l];
            double nrm = sqrt(nrm2);
            double ho = H[k + K * t], trace = 0.0;
            for (int64_t l = 0; l < Lt; ++l) {
                const double *w = W + k + K * N * l;
                double *r = resids + N * (t + l);
                for (int64_t n = 0; n < N; ++n) {
                    double rv = r[n] - ho * w[K * n];   /* :139-140 remove factor */
                    r[n] = rv;
                    trace -= w[K * n] * rv;             /* :152 dot(Wkt, -remainder) */
                }
            }
            double hn = (trace - l1H) / (nrm * nrm + CMF_EPS + l2H);  /* :153 */
            hn = hn > 0.0 ? hn : 0.0;
            H[k + K * t] = hn;
            for (int64_t l = 0; l < Lt; ++l) {           /* :146 add back */
                const double *w = W + k + K * N * l;
                double *r = resids + N * (t + l);
                for (int64_t n = 0; n < N; ++n) r[n] += hn * w[K * n];
            }
        }
    free(wn2);
    return frob_norm(resids, (size_t)N * T) / data_norm;   /* :41 */
}

int oracle_fit_hals(int64_t N, int64_t T, int64_t K, int64_t L,
                    const double *data, double *W, double *H,
                    int64_t max_itr, double max_time,
                    int check_convergence, int64_t patience, double tol, int eval_mode,
                    double l1W, double l2W, double l1H, double l2H,
                    double *loss_hist, double *time_hist, int64_t *n_hist)
{
    size_t NT = (size_t)N * T;
    double *resids = (double *)malloc(sizeof(double) * NT);
    double data_norm = frob_norm(data, NT);                    /* hals.jl:23 */
    oracle_tensor_conv(N, T, K, L, W, H, resids);              /* hals.jl:22 */
    for (size_t i = 0; i < NT; ++i) resids[i] -= data[i];
    int stopped = 0;
    int64_t len = 0;
    loss_hist[len] = frob_norm(resids, NT) / data_norm;        /* alternating.jl:37 (same value) */
    time_hist[len] = 0.0;
    ++len;
    int64_t itr = 1;
    while (itr <= max_itr && time_hist[len - 1] <= max_time) {
        itr += 1;
        double t0 = now_s();
        if (!eval_mode) oracle_hals_update_motifs(N, T, K, L, W, H, resids, l1W, l2W);
        double loss = oracle_hals_update_feature_maps(N, T, K, L, W, H, resids, data_norm, l1H, l2H);
        time_hist[len] = time_hist[len - 1] + (now_s() - t0);
        loss_hist[len] = loss;
        ++len;
        if (check_convergence && oracle_converged(loss_hist, len, patience, tol)) { stopped = 1; break; }
    }
    *n_hist = len;
    free(resids);
    return stopped;
}

/* ------------------------------------------------------------------ */
/* init_rand(data, L, K): src/model.jl:113-125                          */
/* W = rand(K,N,L); H = rand(K,T) with the portable RNG (streams 0, 1)  */
/* in Julia memory order; alpha = <data,est>/||est||^2; both *= sqrt|a|. */
/* ------------------------------------------------------------------ */
/* PGD rule: src/algs/pgd.jl (second, independent restatement; the    */
/* first is the numpy one in cmf_oracle.py)                           */
/* ------------------------------------------------------------------ */
static double sgn(double x) { return (x > 0.0) - (x < 0.0); }

/* projection!(::UnitNormConstraint, x): pgd.jl:100-110.  x has K slices along its first (fastest) dimension:
 * element (k, j) at x[k + K*j], j < M. */
static void unit_norm_projection(double *x, int64_t K, int64_t M)
{
    for (int64_t k = 0; k < K; ++k) {
        double ss = 0.0;
        for (int64_t j = 0; j < M; ++j) ss += x[k + K * j] * x[k + K * j];
        const double mag = sqrt(ss);
        if (mag > 1.0)
            for (int64_t j = 0; j < M; ++j) x[k + K * j] /= mag;
    }
}

/* pgd!: pgd.jl:224-255.  x (nx entries, K slices of M) is W or H; est holds tensor_conv(W, H) on entry and on exit.
 * is_W selects compute_gradW! (:206-214) or compute_gradH! (:218-221).  Returns the new step. */
static double pgd_step(int64_t N, int64_t T, int64_t K, int64_t L, const double *data, double *W, double *H, double *est,
                       double *x, double *gradx, int is_W, double step, double pen_sq, double pen_abs, int constr,
                       int loss_abs, const double *mask, double *cur_loss)
{
    const size_t NT = (size_t)N * T;
    const size_t nx = is_W ? (size_t)K * N * L : (size_t)K * T;
    /* :230  grad!(loss_func, est, est, data): 2 (est - data) (:31-33) or sign(est - data) (:42-44), times the mask (:64-67) */
    for (size_t i = 0; i < NT; ++i) {
        double g = loss_abs ? sgn(est[i] - data[i]) : 2.0 * (est[i] - data[i]);
        est[i] = mask ? g * mask[i] : g;
    }
    if (is_W) oracle_hxt(N, T, K, L, H, est, gradx);       /* :231, :206-214 */
    else oracle_tensor_transconv(N, T, K, L, W, est, gradx); /* :231, :218-221 */
    for (size_t i = 0; i < nx; ++i) gradx[i] += 2.0 * pen_sq * x[i] + pen_abs * sgn(x[i]); /* :232-234, :78-80, :87-89 */
    const double alpha = step / (frob_norm(gradx, nx) + 2.220446049250313e-16);            /* :237 */
    for (size_t i = 0; i < nx; ++i) x[i] -= alpha * gradx[i];                              /* :240 */
    if (constr == 1) {
        for (size_t i = 0; i < nx; ++i) x[i] = x[i] > 2.220446049250313e-16 ? x[i] : 2.220446049250313e-16; /* :94-96 */
    } else if (constr == 2) {
        unit_norm_projection(x, K, is_W ? N * L : T);                                      /* :100-110 */
    }
    oracle_tensor_conv(N, T, K, L, W, H, est);                                             /* :245 */
    double loss = 0.0;                                                                     /* :246 eval(loss_func, data, est) */
    for (size_t i = 0; i < NT; ++i) {
        const double m = mask ? mask[i] : 1.0;
        const double d = m * data[i] - m * est[i];                                         /* :68-70 */
        loss += loss_abs ? fabs(d) : d * d;                                                /* :45-47 / :34-36 */
    }
    step *= (loss < *cur_loss) ? 1.05 : 0.70;                                              /* :248-252 */
    *cur_loss = loss;                                                                      /* :253 */
    return step;
}

/* fit(::AlternatingOptimizer{PGDUpdate}) for exactly max_itr iterations (alternating.jl:44-60 without the stop tests):
 * loss_hist[0] = compute_loss (:37), then update_motifs! / update_feature_maps! (pgd.jl:158-202).
 * constr: 0 none, 1 NonnegConstraint, 2 UnitNormConstraint; loss_abs: AbsoluteLoss instead of SquareLoss; mask may be NULL. */
void oracle_fit_pgd(int64_t N, int64_t T, int64_t K, int64_t L, const double *data, double *W, double *H, int64_t max_itr,
                    double penW_sq, double penW_abs, double penH_sq, double penH_abs, int constrW, int constrH,
                    int loss_abs, const double *mask, double *loss_hist, double *steps_out)
{
    const size_t NT = (size_t)N * T;
    double *est = (double *)malloc(sizeof(double) * NT);
    double *gradW = (double *)malloc(sizeof(double) * (size_t)K * N * L);
    double *gradH = (double *)malloc(sizeof(double) * (size_t)K * T);
    const double datanorm = frob_norm(data, NT);       /* pgd.jl:135 */
    double stepW = 5.0, stepH = 5.0, cur = datanorm;    /* :148-151 (cur_loss starts as the norm, not its square) */
    loss_hist[0] = oracle_compute_loss(N, T, K, L, data, W, H, est);
    oracle_tensor_conv(N, T, K, L, W, H, est);          /* :136 */
    for (int64_t it = 0; it < max_itr; ++it) {
        stepW = pgd_step(N, T, K, L, data, W, H, est, W, gradW, 1, stepW, penW_sq, penW_abs, constrW, loss_abs, mask, &cur);
        stepH = pgd_step(N, T, K, L, data, W, H, est, H, gradH, 0, stepH, penH_sq, penH_abs, constrH, loss_abs, mask, &cur);
        loss_hist[it + 1] = sqrt(cur / (datanorm * datanorm)); /* :201 */
    }
    steps_out[0] = stepW;
    steps_out[1] = stepH;
    free(est); free(gradW); free(gradH);
}

/* ------------------------------------------------------------------ */
void oracle_init_rand(int64_t N, int64_t T, int64_t K, int64_t L, uint64_t seed,
                      const double *data, double *W, double *H)
{
    size_t KNL = (size_t)K * N * L, KT = (size_t)K * T, NT = (size_t)N * T;
    uint64_t bW = rng_base(seed, 0), bH = rng_base(seed, 1);
    for (size_t i = 0; i < KNL; ++i) W[i] = rng_u01(bW, i);
    for (size_t i = 0; i < KT; ++i) H[i] = rng_u01(bH, i);
    double *est = (double *)malloc(sizeof(double) * NT);
    oracle_tensor_conv(N, T, K, L, W, H, est);
    double dot = 0.0, nn = 0.0;
    for (size_t i = 0; i < NT; ++i) { dot += data[i] * est[i]; nn += est[i] * est[i]; }
    double alpha = dot / nn;
    double s = sqrt(fabs(alpha));
    for (size_t i = 0; i < KNL; ++i) W[i] *= s;
    for (size_t i = 0; i < KT; ++i) H[i] *= s;
    free(est);
}

/* ------------------------------------------------------------------ */
/* gen_synthetic: follows synthetic_sequences, datasets/synthetic.jl:29-61 */
/* (README.md:14 promises gen_synthetic(N=,T=) returning the data).     */
/*  streams: 10,11 gamma normals; 12 gamma uniforms; 13 centres;        */
/*           14 exponential; 15 bernoulli; 16,17 noise normals.         */
/* ------------------------------------------------------------------ */
void oracle_gen_synthetic(int64_t N, int64_t T, int64_t K, int64_t L,
                          double alpha, double p_h, double sigma, double noise_scale,
                          uint64_t seed, double *data, double *W, double *H)
{
    uint64_t bga = rng_base(seed, 10), bgb = rng_base(seed, 11), bgu = rng_base(seed, 12);
    uint64_t bc = rng_base(seed, 13), be = rng_base(seed, 14), bb = rng_base(seed, 15);
    uint64_t bna = rng_base(seed, 16), bnb = rng_base(seed, 17);
    /* :42 mW[n,:] ~ Dirichlet(alpha) = normalised Gamma(alpha,1) */
    double *mW = (double *)malloc(sizeof(double) * (size_t)N * K);
    for (int64_t n = 0; n < N; ++n) {
        double s = 0.0;
        for (int64_t k = 0; k < K; ++k) {
            double g = rng_gamma(bga, bgb, bgu, (uint64_t)(n * K + k), alpha);
            mW[n * K + k] = g; s += g;
        }
        if (!(s > 0.0)) { /* all underflowed: put the mass on one component */
            for (int64_t k = 0; k < K; ++k) mW[n * K + k] = (k == n % K) ? 1.0 : 0.0;
            s = 1.0;
        }
        for (int64_t k = 0; k < K; ++k) mW[n * K + k] /= s;
    }
    /* :47-51 W[k,n,:] = mW[n,k] * pdf(Normal(cent,sigma), range(-1,1,L)) */
    const double inv_s2pi = 0.39894228040143267793994605993438;
    for (int64_t k = 0; k < K; ++k)
        for (int64_t n = 0; n < N; ++n) {
            /* Iterators.product(1:K,1:N): k fastest */
            double cent = -1.0 + 2.0 * rng_u01(bc, (uint64_t)(k + K * n));
            for (int64_t l = 0; l < L; ++l) {
                double x = (L > 1) ? (-1.0 + 2.0 * (double)l / (double)(L - 1)) : -1.0;
                double z = (x - cent) / sigma;
                W[k + K * (n + N * l)] = mW[n * K + k] * (inv_s2pi / sigma) * exp(-0.5 * z * z);
            }
        }
    /* :54 H = Exponential(1) .* Bernoulli(p_h) */
    for (size_t i = 0; i < (size_t)K * T; ++i) {
        double e = -log(rng_u01_open0(be, i));
        H[i] = (rng_u01(bb, i) < p_h) ? e : 0.0;
    }
    /* :57-58 data = max.(0, tensor_conv(W,H) + noise) */
    oracle_tensor_conv(N, T, K, L, W, H, data);
    for (size_t i = 0; i < (size_t)N * T; ++i) {
        double v = data[i] + noise_scale * rng_normal(bna, bnb, i);
        data[i] = v > 0.0 ? v : 0.0;
    }
    free(mW);
}
