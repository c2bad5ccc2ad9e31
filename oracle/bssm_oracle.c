/*
 * bssm_oracle.c -- CPU restatement of the helske/bssm particle-filter hot path.
 * TEST INFRASTRUCTURE ONLY; see bssm_oracle.h for the parity status (deterministic
 * half pinned by the reference tests' stated values; particle systems "parity
 * unpinned" at bit level, pinned statistically and analytically).
 *
 * Citations are file:line under the reference tree (src/ unless stated).
 */
#include "bssm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846
#define ORC_ZERO_TOL 1e-12 /* model_ssm_ung.h:21, model_ssm_ulg.h:19 */
#define ORC_EPS 2.220446049250313e-16

/* ------------------------------------------------------------------ */
/* Random stream                                                       */
/* ------------------------------------------------------------------ */

/* Philox4x32-10 (Salmon et al., SC'11), Random123 constants. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline uint64_t bits53(uint32_t a, uint32_t b) {
  return ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6);
}

static void draw(const orc_rng *r, uint32_t purpose, uint32_t t, uint32_t i, uint32_t d, uint32_t w[4]) {
  uint32_t ctr[4] = { i, t, d, purpose };
  uint32_t key[2] = { r->seed, r->stream };
  orc_philox4x32_10(ctr, key, w);
}

/* purpose: 0 initial-state normals, 1 resampling uniform, 2 transition normals,
 * 3 trajectory pick, 4 SDE increments.
 * Two block formats, chosen by the number of normals a draw needs (the kernels' pf_device.h states the same rule):
 *   WIDE (a draw of <= 2 normals; every Milstein sub-step pair): one block per pair -- words 0-1 a 53-bit radius uniform,
 *     word 2 a 32-bit angle, word 3 left over.  The resampling uniform of slot i at step t of a filter whose step draw is
 *     WIDE is that spare word of the "step block" (i, t, draw 0, purpose 2), the block that also carries the slot's
 *     transition normals.
 *   PACKED (a draw of > 2 normals): pair p comes from block (i, t, p / 2, purpose), words (0,1) for even p and (2,3) for
 *     odd p: 40-bit radius uniform (first word, then the top 8 bits of the second) and 24-bit angle (the low 24 bits of the
 *     second).  Nothing is left over, so the resampling uniform of slot i at step t of such a filter is word i % 4 of block
 *     (i / 4, t, 0, purpose 1), shared by four consecutive slots.
 * PACKED applies to the linear-state filters (ssm_ung / ssm_mng / ssm_ulg / ssm_mlg: bootstrap, psi-APF, simulation smoother);
 * the functor-model filters (ssm_nlg, ssm_sde, EKF particle filter) draw WIDE blocks whatever the count. */
double orc_rng_uniform(const orc_rng *r, uint32_t purpose, uint32_t t, uint32_t i, uint32_t d) {
  if (r->mode == 1 && purpose == 1) return r->unif[(size_t)t * r->N + i];
  uint32_t w[4];
  if (purpose == 1) { /* d = number of normals of the filter's step draw */
    if (d <= 2) {
      draw(r, 2, t, i, 0, w);
      return (double)w[3] * 0x1.0p-32;
    }
    draw(r, 1, t, i >> 2, 0, w);
    return (double)w[i & 3u] * 0x1.0p-32;
  }
  draw(r, purpose, t, i, d, w);
  return (double)bits53(w[0], w[1]) * 0x1.0p-53;
}

void orc_rng_normal_pair(const orc_rng *r, uint32_t purpose, uint32_t t, uint32_t i, uint32_t d,
                         double *z0, double *z1) {
  uint32_t w[4];
  draw(r, purpose, t, i, d, w);
  double u1 = (double)(bits53(w[0], w[1]) + 1) * 0x1.0p-53; /* (0,1] */
  double u2 = (double)w[2] * 0x1.0p-32;                      /* [0,1) */
  double rad = sqrt(-2.0 * log(u1));
  double ang = 2.0 * ORC_PI * u2;
  *z0 = rad * cos(ang);
  *z1 = rad * sin(ang);
}

/* pair p of a PACKED draw */
void orc_rng_normal_pair_packed(const orc_rng *r, uint32_t purpose, uint32_t t, uint32_t i, uint32_t p,
                                double *z0, double *z1) {
  uint32_t w[4];
  draw(r, purpose, t, i, p >> 1, w);
  const uint32_t a = w[2 * (p & 1u)], b = w[2 * (p & 1u) + 1];
  double u1 = (double)((((uint64_t)a << 8) | (uint64_t)(b >> 24)) + 1) * 0x1.0p-40; /* (0,1] */
  double u2 = (double)(b & 0xFFFFFFu) * 0x1.0p-24;                                  /* [0,1) */
  double rad = sqrt(-2.0 * log(u1));
  double ang = 2.0 * ORC_PI * u2;
  *z0 = rad * cos(ang);
  *z1 = rad * sin(ang);
}

/* cnt standard normals for particle i at step t */
void orc_rng_normals(const orc_rng *r, uint32_t purpose, uint32_t t, uint32_t i, int cnt, double *out) {
  for (int j = 0; j < cnt; j += 2) {
    double z0, z1;
    if (cnt <= 2) orc_rng_normal_pair(r, purpose, t, i, 0, &z0, &z1);
    else orc_rng_normal_pair_packed(r, purpose, t, i, (uint32_t)(j / 2), &z0, &z1);
    out[j] = z0;
    if (j + 1 < cnt) out[j + 1] = z1;
  }
}
/* test helper: the draws of particles i0 ... i0 + count - 1 at step t, cnt normals each (out: count x cnt, row-major) */
void orc_rng_normals_bulk(const orc_rng *r, uint32_t purpose, uint32_t t, uint32_t i0, uint32_t count, int cnt, double *out) {
#pragma omp parallel for schedule(static)
  for (uint32_t q = 0; q < count; q++) orc_rng_normals(r, purpose, t, i0 + q, cnt, out + (size_t)q * cnt);
}
static void rng_normals_fmt(const orc_rng *r, uint32_t purpose, uint32_t t, uint32_t i, int cnt, double *out, int wide_only) {
  if (r->mode == 1) {
    if (purpose == 0) {
      for (int j = 0; j < cnt; j++) out[j] = r->init_normals[(size_t)i * cnt + j];
    } else {
      for (int j = 0; j < cnt; j++) out[j] = r->normals[((size_t)t * r->N + i) * r->kk + j];
    }
    return;
  }
  if (!wide_only) { orc_rng_normals(r, purpose, t, i, cnt, out); return; }
  for (int j = 0; j < cnt; j += 2) {
    double z0, z1;
    orc_rng_normal_pair(r, purpose, t, i, (uint32_t)(j / 2), &z0, &z1);
    out[j] = z0;
    if (j + 1 < cnt) out[j + 1] = z1;
  }
}
static void rng_normals(const orc_rng *r, uint32_t purpose, uint32_t t, uint32_t i, int cnt, double *out) {
  rng_normals_fmt(r, purpose, t, i, cnt, out, 0);
}
/* the functor-model filters (ssm_nlg, ssm_sde, EKF particle filter) keep WIDE blocks whatever the count: one block per pair,
 * the resampling uniform in the spare word of the step block (their kernels are instantiated for the padded dimensions) */
static void rng_normals_wide(const orc_rng *r, uint32_t purpose, uint32_t t, uint32_t i, int cnt, double *out) {
  rng_normals_fmt(r, purpose, t, i, cnt, out, 1);
}

/* ------------------------------------------------------------------ */
/* Small dense helpers (m <= ORC_MAXM)                                 */
/* ------------------------------------------------------------------ */
#define ORC_MAXM 64

/* lower Cholesky of the sub-matrix selected by idx (count entries); arma::chol(.,"lower") */
static int chol_lower_sub(const double *P, int m, const int *idx, int cnt, double *L /* m x m, zeroed */) {
  for (int a = 0; a < cnt; a++) {
    for (int b = 0; b <= a; b++) {
      double s = P[idx[a] + m * idx[b]];
      for (int c = 0; c < b; c++) s -= L[idx[a] + m * idx[c]] * L[idx[b] + m * idx[c]];
      if (a == b) {
        if (!(s > 0.0)) return -1;
        L[idx[a] + m * idx[a]] = sqrt(s);
      } else {
        L[idx[a] + m * idx[b]] = s / L[idx[b] + m * idx[b]];
      }
    }
  }
  return 0;
}

/* model_ssm_ung.cpp:614-619 */
static void ung_initial_chol(const double *P1, int m, double *L) {
  int idx[ORC_MAXM], cnt = 0;
  memset(L, 0, sizeof(double) * m * m);
  for (int j = 0; j < m; j++) if (P1[j + m * j] > 0) idx[cnt++] = j;
  if (cnt > 0) chol_lower_sub(P1, m, idx, cnt, L);
}

/* psd_chol.cpp:5-18 */
void orc_psd_chol(const double *x, int m, double *L) {
  int idx[ORC_MAXM], cnt = 0;
  double dmax = x[0];
  for (int j = 1; j < m; j++) if (x[j + m * j] > dmax) dmax = x[j + m * j];
  double thr = fmax(ORC_EPS, ORC_EPS * m * dmax);
  memset(L, 0, sizeof(double) * m * m);
  for (int j = 0; j < m; j++) if (x[j + m * j] > thr) idx[cnt++] = j;
  if (cnt > 0) chol_lower_sub(x, m, idx, cnt, L);
}

/* ------------------------------------------------------------------ */
/* distribution constants: distr_consts.cpp:7-31                       */
/* ------------------------------------------------------------------ */
static double orc_lchoose(double n, double k) { /* distr_consts.cpp:7-9 */
  return -log(n + 1) - lgamma(n - k + 1) - lgamma(k + 1) + lgamma(n + 2);
}
static double norm_log_const(double sd) { return -0.5 * log(2.0 * ORC_PI) - log(sd); }

/* model_ssm_ung.cpp:690-710 (bsf) and :351-368 (first part of compute_const_term) */
double orc_ung_const_term(const orc_ung_model *mod) {
  double res = 0.0;
  int nfin = 0;
  for (int t = 0; t < mod->n; t++) {
    double y = mod->y[t];
    if (!isfinite(y)) continue;
    nfin++;
    double u = mod->u ? mod->u[t] : 1.0;
    switch (mod->distribution) {
    case ORC_POISSON: res += y * log(u) - lgamma(y + 1); break;                 /* distr_consts.cpp:38-44 */
    case ORC_BINOMIAL: res += orc_lchoose(u, y); break;                          /* :46-52 */
    case ORC_NEGBIN: res += orc_lchoose(y + mod->phi - 1, y) + mod->phi * log(mod->phi) + y * log(u); break; /* :54-60 */
    case ORC_GAMMA: res += mod->phi * log(mod->phi) - lgamma(mod->phi) + (mod->phi - 1) * log(y) - mod->phi * log(u); break; /* :62-67 */
    default: break;
    }
  }
  if (mod->distribution == ORC_SVM) res = nfin * norm_log_const(mod->phi); /* model_ssm_ung.cpp:692 */
  return res;
}

/* ------------------------------------------------------------------ */
/* observation densities                                               */
/* ------------------------------------------------------------------ */
static inline double ung_signal(const orc_ung_model *mod, int t, const double *a) {
  /* model_ssm_ung.cpp:462-463: D(t*Dtv) + Z.col(t*Ztv).t() * alpha + xbeta(t) */
  const double *Zt = mod->Z + (size_t)mod->m * (t * mod->Ztv);
  double za = 0.0;
  for (int j = 0; j < mod->m; j++) za += Zt[j] * a[j];
  return mod->D[t * mod->Dtv] + za + mod->xbeta[t];
}

/* model_ssm_ung.cpp:447-493, one particle; gaussian: model_ssm_ulg.cpp:582-586 */
static inline double ung_log_obs_density(const orc_ung_model *mod, int t, const double *a) {
  double y = mod->y[t], phi = mod->phi;
  switch (mod->distribution) {
  case ORC_SVM: {
    double s = a[0]; /* :456 */
    return -0.5 * (s + pow(y / phi, 2.0) * exp(-s));
  }
  case ORC_POISSON: { double s = ung_signal(mod, t, a); return y * s - mod->u[t] * exp(s); }
  case ORC_BINOMIAL: { double s = ung_signal(mod, t, a); return y * s - mod->u[t] * log1p(exp(s)); }
  case ORC_NEGBIN: { double s = ung_signal(mod, t, a); return y * s - (y + phi) * log(phi + mod->u[t] * exp(s)); }
  case ORC_GAMMA: { double s = ung_signal(mod, t, a); return -phi * (s + (y * exp(-s) / mod->u[t])); }
  case ORC_GAUSSIAN: {
    const double *Zt = mod->Z + (size_t)mod->m * (t * mod->Ztv);
    double za = 0.0;
    for (int j = 0; j < mod->m; j++) za += Zt[j] * a[j];
    double mu = mod->D[t * mod->Dtv] + za; /* ulg bsf ignores xbeta: model_ssm_ulg.cpp:583-584 */
    double H = mod->H[t * mod->Htv];
    return -0.5 * pow(y - mu, 2.0) / (H * H);
  }
  }
  return 0.0;
}

/* model_ssm_ung.cpp:385-439, one particle */
static inline double ung_log_weight(const orc_ung_model *mod, const orc_psi_setup *su, int t, const double *a) {
  double s = (mod->distribution == ORC_SVM) ? a[0] : ung_signal(mod, t, a);
  return ung_log_obs_density(mod, t, a) + 0.5 * pow((su->approx_y[t] - s) / su->approx_H[t], 2.0);
}

/* ------------------------------------------------------------------ */
/* stratified resampling: stratified_sample.cpp:9-28                   */
/* ------------------------------------------------------------------ */
/* general form: p has n_elem entries, N samples are drawn (the filters always call it with n_elem == N) */
void orc_stratified_sample_n(double *p, unsigned n_elem, const double *r, unsigned N, unsigned *out) {
  for (unsigned k = 1; k < n_elem; k++) p[k] += p[k - 1]; /* arma::cumsum, :12 */
  p[n_elem - 1] = 1;                                      /* :13 */
  unsigned j = 0;
  double alpha = 1.0 / N;                                 /* :17 */
  for (unsigned k = 0; k < n_elem && j < N; k++) {        /* :18-23 */
    while (j < N && (r[j] + j) * alpha <= p[k]) { out[j] = k; j++; }
  }
  while (j < N) { out[j] = n_elem - 1; j++; }             /* :24-27 */
}
void orc_stratified_sample(double *p, const double *r, unsigned N, unsigned *out) {
  orc_stratified_sample_n(p, N, r, N, out);
}

/* ---- tie-following (test instrument for full-size parity; NOT part of the reference's algorithm) ----
 * At N = 65 536 a slot's target (r_j + j) / N lands within rounding distance of a cumulative-sum entry about once per
 * 10^9 particle-steps; the other implementation under test (different summation order, different last bits in exp / log)
 * may then pick the neighbouring ancestor, after which two free-running systems are different, equally valid particle
 * systems.  With a follow table installed, every filter of this file still resamples exactly as the reference does
 * (stratified_sample.cpp:9-28) and compares its ancestors with the table's: an entry that differs is ADOPTED iff it is the
 * adjacent particle AND the slot's target lies within `tol` strata of the cumulative-sum entry that separates the two
 * (an "explained tie": either answer is a correctly rounded evaluation of the same comparison); anything else is counted
 * as unexplained and NOT adopted, so the run diverges and the caller's comparison fails.  The caller then holds the
 * reference recursion along the very path the device took and can compare everything strictly. */
static _Thread_local orc_follow *g_follow = 0;
void orc_follow_begin(orc_follow *f) { f->ties = 0; f->unexplained = 0; f->max_margin = 0.0; f->first_tie_t = -1; f->first_tie_slot = -1; g_follow = f; }
void orc_follow_end(void) { g_follow = 0; }

static void filter_resample(double *nw, const double *r, unsigned N, unsigned *out, int t) {
  orc_stratified_sample(nw, r, N, out); /* nw now holds its cumulative sum */
  orc_follow *f = g_follow;
  if (!f) return;
  const unsigned *want = f->table + (size_t)N * t;
  for (unsigned j = 0; j < N; j++) {
    if (out[j] == want[j]) continue;
    unsigned a = out[j], b = want[j];
    unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    double margin = fabs((r[j] + j) / N - nw[lo]) * N; /* in strata (1 / N) */
    if (hi - lo == 1 && margin <= f->tol) {
      if (f->ties == 0) { f->first_tie_t = t; f->first_tie_slot = (int)j; }
      f->ties++;
      if (margin > f->max_margin) f->max_margin = margin;
      out[j] = b;
    } else {
      f->unexplained++;
    }
  }
}

/* shared tail of every filter step: model_ssm_ung.cpp:634-647 / 672-687.
 * logw holds N log-weights; on return wcol = exp(logw - max), nw = normalised.
 * returns 0 on degeneracy. */
static int normalise_step(const double *logw, unsigned N, double *wcol, size_t wstride, double *nw, double *ll_inc) {
  double maxw = logw[0];
  for (unsigned i = 1; i < N; i++) if (logw[i] > maxw) maxw = logw[i];
  double sumw = 0.0;
  for (unsigned i = 0; i < N; i++) { wcol[i * wstride] = exp(logw[i] - maxw); }
  for (unsigned i = 0; i < N; i++) sumw += wcol[i * wstride];
  if (!(sumw > 0.0)) return 0;
  for (unsigned i = 0; i < N; i++) nw[i] = wcol[i * wstride] / sumw;
  *ll_inc = maxw + log(sumw / N);
  return 1;
}

#define A_(j, t, i) alpha[(size_t)(j) + (size_t)m * ((size_t)(t) + (size_t)(n + 1) * (size_t)(i))]

/* ------------------------------------------------------------------ */
/* bootstrap filter: model_ssm_ung.cpp:611-712 (+ model_ssm_ulg.cpp:557-643) */
/* ------------------------------------------------------------------ */
double orc_ung_bsf_filter(const orc_ung_model *mod, unsigned N, const orc_rng *rng,
                          double *alpha, double *weights, unsigned *indices) {
  const int n = mod->n, m = mod->m, k = mod->k;
  const int gauss = mod->distribution == ORC_GAUSSIAN;
  double L[ORC_MAXM * ORC_MAXM];
  if (gauss) orc_psd_chol(mod->P1, m, L); else ung_initial_chol(mod->P1, m, L);

  double *nw = (double *)malloc(sizeof(double) * N);
  double *lw = (double *)malloc(sizeof(double) * N);
  double *r = (double *)malloc(sizeof(double) * N);
  double *atmp = (double *)malloc(sizeof(double) * N * m);
  double um[ORC_MAXM];

  for (unsigned i = 0; i < N; i++) { /* :621-627 */
    rng_normals(rng, 0, 0, i, m, um);
    for (int j = 0; j < m; j++) {
      double s = 0.0;
      for (int c = 0; c < m; c++) s += L[j + m * c] * um[c];
      A_(j, 0, i) = mod->a1[j] + s;
    }
  }
  double loglik = 0.0, inc;
  if (isfinite(mod->y[0])) { /* :633-643 */
    for (unsigned i = 0; i < N; i++) lw[i] = ung_log_obs_density(mod, 0, &A_(0, 0, i));
    if (!normalise_step(lw, N, weights, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
    loglik = inc + (gauss ? norm_log_const(mod->H[0]) : 0.0);
  } else {
    for (unsigned i = 0; i < N; i++) { weights[i] = 1.0; nw[i] = 1.0 / N; }
  }
  for (int t = 0; t < n; t++) {
    for (unsigned i = 0; i < N; i++) r[i] = orc_rng_uniform(rng, 1, t, i, (uint32_t)k); /* :650-653 */
    filter_resample(nw, r, N, indices + (size_t)N * t, t);                 /* :655 */
    for (unsigned i = 0; i < N; i++)                                          /* :659-661 */
      for (int j = 0; j < m; j++) atmp[j + (size_t)m * i] = A_(j, t, indices[i + (size_t)N * t]);
    const double *Tt = mod->T + (size_t)m * m * (t * mod->Ttv);
    const double *Rt = mod->R + (size_t)m * k * (t * mod->Rtv);
    const double *Ct = mod->C + (size_t)m * (t * mod->Ctv);
    for (unsigned i = 0; i < N; i++) { /* :663-670 */
      rng_normals(rng, 2, t, i, k, um);
      for (int j = 0; j < m; j++) {
        double s = 0.0, q = 0.0;
        for (int c = 0; c < m; c++) s += Tt[j + m * c] * atmp[c + (size_t)m * i];
        for (int c = 0; c < k; c++) q += Rt[j + m * c] * um[c];
        A_(j, t + 1, i) = Ct[j] + s + q;
      }
    }
    double *wcol = weights + (size_t)N * (t + 1);
    if ((t < n - 1) && isfinite(mod->y[t + 1])) { /* :672-683 */
      for (unsigned i = 0; i < N; i++) lw[i] = ung_log_obs_density(mod, t + 1, &A_(0, t + 1, i));
      if (!normalise_step(lw, N, wcol, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
      loglik += inc + (gauss ? norm_log_const(mod->H[(t + 1) * mod->Htv]) : 0.0);
    } else {
      for (unsigned i = 0; i < N; i++) { wcol[i] = 1.0; nw[i] = 1.0 / N; }
    }
  }
  if (!gauss) loglik += orc_ung_const_term(mod); /* :690-710 */
done:
  free(nw); free(lw); free(r); free(atmp);
  return loglik;
}

/* ------------------------------------------------------------------ */
/* Linear-Gaussian machinery (ssm_ulg)                                 */
/* ------------------------------------------------------------------ */
static void mat_mul(const double *A, const double *B, int r, int inner, int c, double *out) {
  for (int j = 0; j < c; j++)
    for (int i = 0; i < r; i++) {
      double s = 0.0;
      for (int q = 0; q < inner; q++) s += A[i + r * q] * B[q + inner * j];
      out[i + r * j] = s;
    }
}
static void mat_mul_bt(const double *A, const double *B, int r, int inner, int c, double *out) { /* A * B' ; B is c x inner */
  for (int j = 0; j < c; j++)
    for (int i = 0; i < r; i++) {
      double s = 0.0;
      for (int q = 0; q < inner; q++) s += A[i + r * q] * B[j + c * q];
      out[i + r * j] = s;
    }
}
static void mat_mul_at(const double *A, const double *B, int r, int inner, int c, double *out) { /* A' * B ; A is inner x r */
  for (int j = 0; j < c; j++)
    for (int i = 0; i < r; i++) {
      double s = 0.0;
      for (int q = 0; q < inner; q++) s += A[q + inner * i] * B[q + inner * j];
      out[i + r * j] = s;
    }
}
static void symmatu(double *A, int m) { /* arma::symmatu: copy upper to lower */
  for (int j = 0; j < m; j++) for (int i = j + 1; i < m; i++) A[i + m * j] = A[j + m * i];
}
static void compute_RR(const double *R, int m, int k, double *RR) { mat_mul_bt(R, R, m, k, m, RR); }

#define G_Z(t) (g->Z + (size_t)m * ((t) * g->Ztv))
#define G_T(t) (g->T + (size_t)m * m * ((t) * g->Ttv))
#define G_R(t) (g->R + (size_t)m * g->k * ((t) * g->Rtv))
#define G_C(t) (g->C + (size_t)m * ((t) * g->Ctv))
#define G_HH(t) (g->H[(t) * g->Htv] * g->H[(t) * g->Htv])

static double dotv(const double *a, const double *b, int m) { double s = 0; for (int j = 0; j < m; j++) s += a[j] * b[j]; return s; }
static double quad(const double *z, const double *P, int m) { /* z' P z */
  double s = 0; for (int j = 0; j < m; j++) { double q = 0; for (int i = 0; i < m; i++) q += z[i] * P[i + m * j]; s += q * z[j]; } return s;
}

/* one prediction step shared by the three Kalman routines. joseph=0: Pt - K K' F form
 * (model_ssm_ulg.cpp:129), joseph=1: (I-KZ')P(I-KZ')' + K HH K' (:276-277, :442-443) */
static void kf_step(const orc_ulg_model *g, int t, const double *at, const double *Pt, int joseph,
                    double *at1, double *Pt1, double *F_out, double *v_out, double *K_out, int *observed) {
  const int m = g->m;
  const double *Zt = G_Z(t), *Tt = G_T(t);
  double RR[ORC_MAXM * ORC_MAXM], tmp[ORC_MAXM * ORC_MAXM], tmp2[ORC_MAXM * ORC_MAXM], M[ORC_MAXM * ORC_MAXM];
  compute_RR(G_R(t), m, g->k, RR);
  double HH = G_HH(t);
  double F = quad(Zt, Pt, m) + HH;
  double ytmp = g->y[t] - g->xbeta[t];
  *F_out = F;
  if (isfinite(ytmp) && F > ORC_ZERO_TOL) {
    *observed = 1;
    double v = ytmp - g->D[t * g->Dtv] - dotv(Zt, at, m);
    *v_out = v;
    double K[ORC_MAXM], af[ORC_MAXM];
    for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += Pt[i + m * j] * Zt[j]; K[i] = s / F; }
    if (K_out) memcpy(K_out, K, sizeof(double) * m);
    for (int i = 0; i < m; i++) af[i] = at[i] + K[i] * v;
    for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += Tt[i + m * j] * af[j]; at1[i] = G_C(t)[i] + s; }
    if (!joseph) {
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) M[i + m * j] = Pt[i + m * j] - K[i] * K[j] * F;
    } else {
      double IKZ[ORC_MAXM * ORC_MAXM];
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) IKZ[i + m * j] = (i == j ? 1.0 : 0.0) - K[i] * Zt[j];
      mat_mul(IKZ, Pt, m, m, m, tmp);
      mat_mul_bt(tmp, IKZ, m, m, m, M);
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) M[i + m * j] += K[i] * HH * K[j];
    }
    mat_mul(Tt, M, m, m, m, tmp);
    mat_mul_bt(tmp, Tt, m, m, m, tmp2);
  } else {
    *observed = 0;
    *v_out = 0.0;
    if (K_out) memset(K_out, 0, sizeof(double) * m);
    for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += Tt[i + m * j] * at[j]; at1[i] = G_C(t)[i] + s; }
    mat_mul(Tt, Pt, m, m, m, tmp);
    mat_mul_bt(tmp, Tt, m, m, m, tmp2);
  }
  for (int q = 0; q < m * m; q++) Pt1[q] = tmp2[q] + RR[q];
  symmatu(Pt1, m);
}

/* model_ssm_ulg.cpp:106-138 */
double orc_ulg_loglik(const orc_ulg_model *g) {
  const int n = g->n, m = g->m;
  double accH = 0, accR = 0;
  for (int t = 0; t < (g->Htv ? n : 1); t++) accH += g->H[t];
  for (int q = 0; q < m * g->k * (g->Rtv ? n : 1); q++) accR += g->R[q];
  if (accH + accR < ORC_ZERO_TOL) return -INFINITY;
  double at[ORC_MAXM], Pt[ORC_MAXM * ORC_MAXM], at1[ORC_MAXM], Pt1[ORC_MAXM * ORC_MAXM];
  memcpy(at, g->a1, sizeof(double) * m);
  memcpy(Pt, g->P1, sizeof(double) * m * m);
  const double LOG2PI = log(2.0 * ORC_PI);
  double ll = 0.0;
  for (int t = 0; t < n; t++) {
    double F, v; int obs;
    kf_step(g, t, at, Pt, 0, at1, Pt1, &F, &v, NULL, &obs);
    if (obs) ll -= 0.5 * (LOG2PI + log(F) + v * v / F);
    memcpy(at, at1, sizeof(double) * m);
    memcpy(Pt, Pt1, sizeof(double) * m * m);
  }
  return ll;
}

/* model_ssm_ulg.cpp:252-305 */
void orc_ulg_fast_smoother(const orc_ulg_model *g, double *at) {
  const int n = g->n, m = g->m;
  double *vt = (double *)calloc(n, sizeof(double)), *Ft = (double *)calloc(n, sizeof(double));
  double *Kt = (double *)calloc((size_t)m * n, sizeof(double));
  int *ob = (int *)calloc(n, sizeof(int));
  double *rt = (double *)calloc((size_t)m * n, sizeof(double));
  double Pt[ORC_MAXM * ORC_MAXM], Pt1[ORC_MAXM * ORC_MAXM];
  memcpy(at, g->a1, sizeof(double) * m);
  memcpy(Pt, g->P1, sizeof(double) * m * m);
  for (int t = 0; t < n; t++) {
    kf_step(g, t, at + (size_t)m * t, Pt, 1, at + (size_t)m * (t + 1), Pt1, &Ft[t], &vt[t], Kt + (size_t)m * t, &ob[t]);
    memcpy(Pt, Pt1, sizeof(double) * m * m);
  }
  double L[ORC_MAXM * ORC_MAXM];
  for (int t = n - 1; t > 0; t--) { /* :284-291 */
    const double *Zt = G_Z(t), *Tt = G_T(t), *r1 = rt + (size_t)m * t;
    double *r0 = rt + (size_t)m * (t - 1);
    if (ob[t]) {
      double IKZ[ORC_MAXM * ORC_MAXM];
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) IKZ[i + m * j] = (i == j ? 1.0 : 0.0) - Kt[i + (size_t)m * t] * Zt[j];
      mat_mul(Tt, IKZ, m, m, m, L);
      for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += L[j + m * i] * r1[j]; r0[i] = Zt[i] / Ft[t] * vt[t] + s; }
    } else {
      for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += Tt[j + m * i] * r1[j]; r0[i] = s; }
    }
  }
  { /* :292-297 */
    const double *Z0 = g->Z, *T0 = g->T, *r0 = rt;
    double tmpv[ORC_MAXM];
    if (isfinite(g->y[0]) && Ft[0] > ORC_ZERO_TOL) {
      double IKZ[ORC_MAXM * ORC_MAXM];
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) IKZ[i + m * j] = (i == j ? 1.0 : 0.0) - Kt[i] * Z0[j];
      mat_mul(T0, IKZ, m, m, m, L);
      for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += L[j + m * i] * r0[j]; tmpv[i] = Z0[i] / Ft[0] * vt[0] + s; }
    } else {
      for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += T0[j + m * i] * r0[j]; tmpv[i] = s; }
    }
    for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += g->P1[i + m * j] * tmpv[j]; at[i] = g->a1[i] + s; }
  }
  for (int t = 0; t < n - 1; t++) { /* :299-301 */
    double RR[ORC_MAXM * ORC_MAXM];
    compute_RR(G_R(t), m, g->k, RR);
    const double *Tt = G_T(t), *a0 = at + (size_t)m * t, *r0 = rt + (size_t)m * t;
    double *a1v = at + (size_t)m * (t + 1);
    for (int i = 0; i < m; i++) {
      double s = 0, q = 0;
      for (int j = 0; j < m; j++) { s += Tt[i + m * j] * a0[j]; q += RR[i + m * j] * r0[j]; }
      a1v[i] = G_C(t)[i] + s + q;
    }
  }
  free(vt); free(Ft); free(Kt); free(ob); free(rt);
}

/* model_ssm_ulg.cpp:419-473 */
void orc_ulg_smoother_ccov(const orc_ulg_model *g, double *at, double *Pt, double *ccov) {
  const int n = g->n, m = g->m, mm = m * m;
  double *vt = (double *)calloc(n, sizeof(double)), *Ft = (double *)calloc(n, sizeof(double));
  double *Kt = (double *)calloc((size_t)m * n, sizeof(double));
  int *ob = (int *)calloc(n, sizeof(int));
  memcpy(at, g->a1, sizeof(double) * m);
  memcpy(Pt, g->P1, sizeof(double) * mm);
  for (int t = 0; t < n; t++) {
    kf_step(g, t, at + (size_t)m * t, Pt + (size_t)mm * t, 1, at + (size_t)m * (t + 1), Pt + (size_t)mm * (t + 1),
            &Ft[t], &vt[t], Kt + (size_t)m * t, &ob[t]);
    memcpy(ccov + (size_t)mm * t, Pt + (size_t)mm * (t + 1), sizeof(double) * mm); /* :449 */
  }
  double rt[ORC_MAXM] = {0}, Nt[ORC_MAXM * ORC_MAXM] = {0};
  double L[ORC_MAXM * ORC_MAXM], A[ORC_MAXM * ORC_MAXM], B[ORC_MAXM * ORC_MAXM], Cm[ORC_MAXM * ORC_MAXM], rn[ORC_MAXM];
  for (int t = n - 1; t >= 0; t--) {
    const double *Zt = G_Z(t), *Tt = G_T(t);
    double *Ptt = Pt + (size_t)mm * t, *cc = ccov + (size_t)mm * t;
    if (ob[t]) {
      double IKZ[ORC_MAXM * ORC_MAXM];
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) IKZ[i + m * j] = (i == j ? 1.0 : 0.0) - Kt[i + (size_t)m * t] * Zt[j];
      mat_mul(Tt, IKZ, m, m, m, L);
      /* ccov = Pt * L' * (I - Nt * ccov) */
      mat_mul(Nt, cc, m, m, m, A);
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) A[i + m * j] = (i == j ? 1.0 : 0.0) - A[i + m * j];
      mat_mul_bt(Ptt, L, m, m, m, B);
      mat_mul(B, A, m, m, m, Cm);
      memcpy(cc, Cm, sizeof(double) * mm);
      for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += L[j + m * i] * rt[j]; rn[i] = Zt[i] / Ft[t] * vt[t] + s; }
      memcpy(rt, rn, sizeof(double) * m);
      mat_mul_at(L, Nt, m, m, m, A); /* L' Nt */
      mat_mul(A, L, m, m, m, B);     /* L' Nt L */
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) Nt[i + m * j] = Zt[i] * Zt[j] / Ft[t] + B[i + m * j];
      symmatu(Nt, m);
    } else {
      mat_mul(Nt, cc, m, m, m, A);
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) A[i + m * j] = (i == j ? 1.0 : 0.0) - A[i + m * j];
      mat_mul_bt(Ptt, Tt, m, m, m, B);
      mat_mul(B, A, m, m, m, Cm);
      memcpy(cc, Cm, sizeof(double) * mm);
      for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += Tt[j + m * i] * rt[j]; rn[i] = s; }
      memcpy(rt, rn, sizeof(double) * m);
      mat_mul_at(Tt, Nt, m, m, m, A);
      mat_mul(A, Tt, m, m, m, B);
      memcpy(Nt, B, sizeof(double) * mm);
      symmatu(Nt, m);
    }
    /* :469-470 */
    for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += Ptt[i + m * j] * rt[j]; at[i + (size_t)m * t] += s; }
    mat_mul(Ptt, Nt, m, m, m, A);
    mat_mul(A, Ptt, m, m, m, B);
    symmatu(B, m);
    for (int q = 0; q < mm; q++) Ptt[q] -= B[q];
  }
  memset(ccov + (size_t)mm * n, 0, sizeof(double) * mm); /* :472 */
  free(vt); free(Ft); free(Kt); free(ob);
}

/* Left singular vectors / singular values of a (numerically) symmetric PSD
 * matrix, standing in for arma::svd_econ(U, s, V, A, "left") in
 * conditional_dist.cpp:14,21,29.  Cyclic Jacobi on the symmetrised matrix;
 * s sorted descending; column sign fixed so the largest-magnitude entry of each
 * column is positive (LAPACK's sign is unspecified; any sign gives a valid
 * square-root factor U diag(sqrt(s))). */
static void sym_svd(const double *Ain, int m, double *U, double *s) {
  double A[ORC_MAXM * ORC_MAXM];
  for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) A[i + m * j] = 0.5 * (Ain[i + m * j] + Ain[j + m * i]);
  for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) U[i + m * j] = (i == j);
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0;
    for (int j = 0; j < m; j++) for (int i = 0; i < j; i++) off += A[i + m * j] * A[i + m * j];
    if (off < 1e-300) break;
    for (int p = 0; p < m - 1; p++) for (int q = p + 1; q < m; q++) {
      double apq = A[p + m * q];
      if (apq == 0.0) continue;
      double app = A[p + m * p], aqq = A[q + m * q];
      double tau = (aqq - app) / (2.0 * apq);
      double tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
      double c = 1.0 / sqrt(1.0 + tt * tt), sn = tt * c;
      for (int r = 0; r < m; r++) { double arp = A[r + m * p], arq = A[r + m * q]; A[r + m * p] = c * arp - sn * arq; A[r + m * q] = sn * arp + c * arq; }
      for (int r = 0; r < m; r++) { double apr = A[p + m * r], aqr = A[q + m * r]; A[p + m * r] = c * apr - sn * aqr; A[q + m * r] = sn * apr + c * aqr; }
      for (int r = 0; r < m; r++) { double urp = U[r + m * p], urq = U[r + m * q]; U[r + m * p] = c * urp - sn * urq; U[r + m * q] = sn * urp + c * urq; }
    }
  }
  for (int j = 0; j < m; j++) s[j] = fabs(A[j + m * j]);
  for (int a = 0; a < m - 1; a++) { /* selection sort, descending */
    int best = a;
    for (int b = a + 1; b < m; b++) if (s[b] > s[best]) best = b;
    if (best != a) {
      double ts = s[a]; s[a] = s[best]; s[best] = ts;
      for (int r = 0; r < m; r++) { double tu = U[r + m * a]; U[r + m * a] = U[r + m * best]; U[r + m * best] = tu; }
    }
  }
  for (int j = 0; j < m; j++) {
    int big = 0;
    for (int r = 1; r < m; r++) if (fabs(U[r + m * j]) > fabs(U[big + m * j])) big = r;
    if (U[big + m * j] < 0) for (int r = 0; r < m; r++) U[r + m * j] = -U[r + m * j];
  }
}

/* conditional_dist.cpp:4-31.
 * ONE documented deviation (round 6), at the level of the reference's own rounding noise: Var(alpha_t | alpha_{t-1}, y) is formed by the
 * subtraction Vt.slice(t) -= tmp * Ct.slice(t-1) (:18), whose rounding error is ~eps * |Vt.slice(t)|; where that covariance is singular
 * (fewer noise terms than states: bsm_ng with a seasonal component has rank 3 of 25 / 53) the decomposition of :21-23 returns the null
 * directions with "singular values" of that size, and U * sqrt(s) carries them as columns of norm ~1e-8 -- in arbitrary, LAPACK- and
 * summation-order-dependent directions (measured here: 3e-8 ... 8e-8 of a factor whose real columns have norm 1e-2 ... 1e-1).  Columns with
 * s_c <= eps * m * trace(Vt.slice(t) before the subtraction) -- the rule of :15 applied at the magnitude the subtraction can resolve --
 * are set to ZERO here, in the product's host code (csrc/host_model.cpp) and in its kernels, so that all three agree to rounding instead
 * of to 1e-7; the kept columns are untouched.  The same rule at t = 0 (a singular smoothed covariance of the first time point). */
static void factor_with_cut(const double *U, const double *s, int m, double tr, double *out) {
  const double thr = ORC_EPS * m * tr;
  for (int c = 0; c < m; c++) {
    const double f = s[c] > thr ? sqrt(s[c]) : 0.0;
    for (int i = 0; i < m; i++) out[i + m * c] = f == 0.0 ? 0.0 : U[i + m * c] * f;
  }
}
static double mat_trace(const double *A, int m) { double t = 0.0; for (int i = 0; i < m; i++) t += A[i + m * i]; return t; }
void orc_conditional_cov(double *Vt, double *Ct, int m, int n1) {
  const int mm = m * m;
  double U[ORC_MAXM * ORC_MAXM], s[ORC_MAXM], W[ORC_MAXM * ORC_MAXM], tmp[ORC_MAXM * ORC_MAXM], A[ORC_MAXM * ORC_MAXM];
  for (int t = n1 - 1; t > 0; t--) {
    double *Vp = Vt + (size_t)mm * (t - 1), *Vc = Vt + (size_t)mm * t;
    double *Cp = Ct + (size_t)mm * (t - 1), *Cc = Ct + (size_t)mm * t;
    sym_svd(Vp, m, U, s);
    /* W = U[:,nz] diag(1/s) U[:,nz]' */
    memset(W, 0, sizeof(W));
    for (int c = 0; c < m; c++) {
      if (!(s[c] > ORC_EPS * m * s[0])) continue;
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) W[i + m * j] += U[i + m * c] * U[j + m * c] / s[c];
    }
    mat_mul_at(Cp, W, m, m, m, tmp); /* Ct(t-1)' * W */
    mat_mul(tmp, Cp, m, m, m, A);
    const double tr = mat_trace(Vc, m);
    for (int q = 0; q < mm; q++) Vc[q] -= A[q];
    memcpy(Cc, tmp, sizeof(double) * mm);
    sym_svd(Vc, m, U, s);
    factor_with_cut(U, s, m, tr, Vc);
  }
  const double tr0 = mat_trace(Vt, m);
  sym_svd(Vt, m, U, s);
  factor_with_cut(U, s, m, tr0, Vt);
}

/* ------------------------------------------------------------------ */
/* psi-APF set-up                                                      */
/* ------------------------------------------------------------------ */
orc_psi_setup *orc_psi_setup_alloc(int n, int m) {
  orc_psi_setup *s = (orc_psi_setup *)calloc(1, sizeof(orc_psi_setup));
  s->n = n; s->m = m;
  s->approx_y = (double *)calloc(n, sizeof(double));
  s->approx_H = (double *)calloc(n, sizeof(double));
  s->mode = (double *)calloc(n, sizeof(double));
  s->scales = (double *)calloc(n, sizeof(double));
  s->alphahat = (double *)calloc((size_t)m * (n + 1), sizeof(double));
  s->Vt = (double *)calloc((size_t)m * m * (n + 1), sizeof(double));
  s->Ct = (double *)calloc((size_t)m * m * (n + 1), sizeof(double));
  return s;
}
void orc_psi_setup_free(orc_psi_setup *s) {
  if (!s) return;
  free(s->approx_y); free(s->approx_H); free(s->mode); free(s->scales);
  free(s->alphahat); free(s->Vt); free(s->Ct); free(s);
}

/* model_ssm_ung.cpp:302-342 */
static void laplace_iter(const orc_ung_model *mod, const double *signal, double *ay, double *aH) {
  const double phi = mod->phi;
  for (int t = 0; t < mod->n; t++) {
    double y = mod->y[t], HH = 0, yy = 0, sg = signal[t];
    switch (mod->distribution) {
    case ORC_SVM: {
      double tmp = y;
      if (fabs(tmp) < 1e-4) tmp = 1e-4;
      HH = 2.0 * exp(sg) / ((tmp / phi) * (tmp / phi));
      yy = sg + 1.0 - 0.5 * HH;
    } break;
    case ORC_POISSON: HH = 1.0 / (exp(sg) * mod->u[t]); yy = y * HH + sg - 1.0; break;
    case ORC_BINOMIAL: { double e = exp(sg); HH = (1.0 + e) * (1.0 + e) / (mod->u[t] * e); yy = y * HH + sg - 1.0 - e; } break;
    case ORC_NEGBIN: { double e = exp(sg) * mod->u[t]; HH = (phi + e) * (phi + e) / (phi * e * (y + phi)); yy = sg + (phi + e) * (y - e) / ((y + phi) * e); } break;
    case ORC_GAMMA: { double e = exp(sg) * mod->u[t]; HH = e / (y * phi); yy = sg - e / y + 1; } break;
    }
    ay[t] = yy;
    aH[t] = sqrt(HH);
  }
}

/* model_ssm_ung.cpp:241-292 */
static void update_scales(const orc_ung_model *mod, orc_psi_setup *su) {
  const double phi = mod->phi;
  for (int t = 0; t < mod->n; t++) {
    su->scales[t] = 0.0;
    double y = mod->y[t];
    if (!isfinite(y)) continue;
    double md = su->mode[t];
    double g = 0.5 * pow((su->approx_y[t] - md) / su->approx_H[t], 2.0);
    switch (mod->distribution) {
    case ORC_SVM: su->scales[t] = -0.5 * (md + pow(y / phi, 2.0) * exp(-md)) + g; break;
    case ORC_POISSON: su->scales[t] = y * md - mod->u[t] * exp(md) + g; break;
    case ORC_BINOMIAL: su->scales[t] = y * md - mod->u[t] * log1p(exp(md)) + g; break;
    case ORC_NEGBIN: su->scales[t] = y * md - (y + phi) * log(phi + mod->u[t] * exp(md)) + g; break;
    case ORC_GAMMA: su->scales[t] = -phi * (md + (y * exp(-md) / mod->u[t])) + g; break;
    }
  }
}

static void make_approx_model(const orc_ung_model *mod, const orc_psi_setup *su, orc_ulg_model *g) {
  g->n = mod->n; g->m = mod->m; g->k = mod->k;
  g->Ztv = mod->Ztv; g->Ttv = mod->Ttv; g->Rtv = mod->Rtv; g->Dtv = mod->Dtv; g->Ctv = mod->Ctv; g->Htv = 1;
  g->y = su->approx_y; g->H = su->approx_H;
  g->Z = mod->Z; g->T = mod->T; g->R = mod->R; g->a1 = mod->a1; g->P1 = mod->P1;
  g->D = mod->D; g->C = mod->C; g->xbeta = mod->xbeta;
}

int orc_ung_psi_setup(const orc_ung_model *mod, const double *is_mode, orc_psi_setup *su) {
  const int n = mod->n, m = mod->m;
  orc_ulg_model g;
  make_approx_model(mod, su, &g);
  double *sm = (double *)malloc(sizeof(double) * m * (n + 1));
  if (is_mode) { /* approximate_for_is: model_ssm_ung.cpp:143-161 */
    memcpy(su->mode, is_mode, sizeof(double) * n);
    laplace_iter(mod, su->mode, su->approx_y, su->approx_H);
    update_scales(mod, su);
    su->approx_loglik = 0.0;
    su->laplace_iters = 0;
  } else { /* approximate(): :87-140, first call (approx_state == -1) */
    memcpy(su->mode, mod->initial_mode, sizeof(double) * n);
    int it = 0;
    double diff = mod->conv_tol + 1;
    double *newmode = (double *)malloc(sizeof(double) * n);
    while (it < mod->max_iter && diff > mod->conv_tol) {
      it++;
      laplace_iter(mod, su->mode, su->approx_y, su->approx_H);
      orc_ulg_fast_smoother(&g, sm);
      for (int t = 0; t < n; t++) {
        if (mod->distribution == ORC_SVM) newmode[t] = sm[(size_t)m * t];
        else {
          const double *Zt = mod->Z + (size_t)m * (t * mod->Ztv);
          newmode[t] = mod->xbeta[t] + mod->D[t * mod->Dtv] + dotv(Zt, sm + (size_t)m * t, m);
        }
      }
      diff = 0.0;
      for (int t = 0; t < n; t++) diff += (newmode[t] - su->mode[t]) * (newmode[t] - su->mode[t]);
      diff /= n;
      memcpy(su->mode, newmode, sizeof(double) * n);
    }
    free(newmode);
    su->laplace_iters = it;
    /* psi_filter :519-527 */
    double gll = orc_ulg_loglik(&g);
    update_scales(mod, su);
    double cterm = orc_ung_const_term(mod);
    { /* compute_const_term :369: minus norm_log_const(approx y, H) over finite y */
      int nf = 0; double sl = 0;
      for (int t = 0; t < n; t++) if (isfinite(mod->y[t])) { nf++; sl += log(su->approx_H[t]); }
      cterm -= (-0.5 * nf * log(2 * ORC_PI) - sl);
    }
    double ssc = 0; for (int t = 0; t < n; t++) ssc += su->scales[t];
    su->approx_loglik = gll + cterm + ssc;
  }
  orc_ulg_smoother_ccov(&g, su->alphahat, su->Vt, su->Ct); /* :533 */
  orc_conditional_cov(su->Vt, su->Ct, m, n + 1);            /* :534 */
  free(sm);
  return 0;
}

/* model_ssm_ung.cpp:536-609 */
double orc_ung_psi_filter(const orc_ung_model *mod, const orc_psi_setup *su, unsigned N,
                          const orc_rng *rng, double *alpha, double *weights, unsigned *indices) {
  const int n = mod->n, m = mod->m, mm = m * m;
  double *nw = (double *)malloc(sizeof(double) * N);
  double *lw = (double *)malloc(sizeof(double) * N);
  double *r = (double *)malloc(sizeof(double) * N);
  double *atmp = (double *)malloc(sizeof(double) * N * m);
  double um[ORC_MAXM];
  for (unsigned i = 0; i < N; i++) { /* :538-544 */
    rng_normals(rng, 0, 0, i, m, um);
    for (int j = 0; j < m; j++) {
      double s = 0.0;
      for (int c = 0; c < m; c++) s += su->Vt[j + m * c] * um[c];
      A_(j, 0, i) = su->alphahat[j] + s;
    }
  }
  double loglik = 0.0, inc;
  if (isfinite(mod->y[0])) { /* :549-561 */
    for (unsigned i = 0; i < N; i++) lw[i] = ung_log_weight(mod, su, 0, &A_(0, 0, i)) - su->scales[0];
    if (!normalise_step(lw, N, weights, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
    loglik = inc + su->approx_loglik; /* max + approx_loglik + log(sum/N) */
  } else {
    for (unsigned i = 0; i < N; i++) { weights[i] = 1.0; nw[i] = 1.0 / N; }
    loglik = su->approx_loglik;
  }
  for (int t = 0; t < n; t++) {
    for (unsigned i = 0; i < N; i++) r[i] = orc_rng_uniform(rng, 1, t, i, (uint32_t)m);
    filter_resample(nw, r, N, indices + (size_t)N * t, t);
    for (unsigned i = 0; i < N; i++)
      for (int j = 0; j < m; j++) atmp[j + (size_t)m * i] = A_(j, t, indices[i + (size_t)N * t]);
    const double *ah0 = su->alphahat + (size_t)m * t, *ah1 = su->alphahat + (size_t)m * (t + 1);
    const double *Ct1 = su->Ct + (size_t)mm * (t + 1), *Vt1 = su->Vt + (size_t)mm * (t + 1);
    for (unsigned i = 0; i < N; i++) { /* :580-587 */
      rng_normals(rng, 2, t, i, m, um);
      for (int j = 0; j < m; j++) {
        double s = 0.0, q = 0.0;
        for (int c = 0; c < m; c++) s += Ct1[j + m * c] * (atmp[c + (size_t)m * i] - ah0[c]);
        for (int c = 0; c < m; c++) q += Vt1[j + m * c] * um[c];
        A_(j, t + 1, i) = ah1[j] + s + q;
      }
    }
    double *wcol = weights + (size_t)N * (t + 1);
    if ((t < n - 1) && isfinite(mod->y[t + 1])) { /* :589-601 */
      for (unsigned i = 0; i < N; i++) lw[i] = ung_log_weight(mod, su, t + 1, &A_(0, t + 1, i)) - su->scales[t + 1];
      if (!normalise_step(lw, N, wcol, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
      loglik += inc;
    } else {
      for (unsigned i = 0; i < N; i++) { wcol[i] = 1.0; nw[i] = 1.0 / N; }
    }
  }
done:
  free(nw); free(lw); free(r); free(atmp);
  return loglik;
}

/* ------------------------------------------------------------------ */
/* smoothing + summaries                                               */
/* ------------------------------------------------------------------ */
/* filter_smoother.cpp:5-18 */
void orc_filter_smoother(double *alpha, const unsigned *indices, int m, int n, unsigned N) {
  unsigned *b = (unsigned *)malloc(sizeof(unsigned) * N);
  double *tube = (double *)malloc(sizeof(double) * m * N);
  for (unsigned i = 0; i < N; i++) b[i] = i;
  for (int t = n - 1; t >= 0; t--) {
    for (unsigned i = 0; i < N; i++) for (int j = 0; j < m; j++) tube[j + (size_t)m * i] = A_(j, t, i);
    for (unsigned i = 0; i < N; i++) {
      unsigned src = indices[b[i] + (size_t)N * t];
      for (int j = 0; j < m; j++) A_(j, t, i) = tube[j + (size_t)m * src];
    }
    for (unsigned i = 0; i < N; i++) b[i] = indices[b[i] + (size_t)N * t];
  }
  free(b); free(tube);
}

/* summary.cpp:4-17 ; alpha m x n1 x N */
void orc_summary(const double *alpha, int m, int n1, unsigned N, double *mean, double *cov) {
  const int n = n1 - 1;
  size_t mt = (size_t)m * n1;
  double *diff = (double *)malloc(sizeof(double) * mt);
  memset(mean, 0, sizeof(double) * mt);
  memset(cov, 0, sizeof(double) * mt * m);
  for (unsigned i = 0; i < N; i++) {
    for (int t = 0; t < n1; t++) for (int j = 0; j < m; j++) {
      double d = A_(j, t, i) - mean[j + (size_t)m * t];
      diff[j + (size_t)m * t] = d;
      mean[j + (size_t)m * t] += d / (i + 1);
    }
    for (int t = 0; t < n1; t++)
      for (int c = 0; c < m; c++) for (int j = 0; j < m; j++)
        cov[j + m * (c + (size_t)m * t)] += diff[j + (size_t)m * t] * (A_(c, t, i) - mean[c + (size_t)m * t]);
  }
  for (size_t q = 0; q < mt * m; q++) cov[q] /= N;
  free(diff);
}

/* summary.cpp:19-39 */
void orc_weighted_summary(const double *alpha, int m, int n1, unsigned N, const double *w, double *mean, double *cov) {
  const int n = n1 - 1;
  size_t mt = (size_t)m * n1;
  double *diff = (double *)malloc(sizeof(double) * mt);
  memset(mean, 0, sizeof(double) * mt);
  memset(cov, 0, sizeof(double) * mt * m);
  double cum = 0;
  for (unsigned i = 0; i < N; i++) {
    if (!(w[i] > 0)) continue;
    double tmp = w[i] + cum;
    for (int t = 0; t < n1; t++) for (int j = 0; j < m; j++) {
      double d = A_(j, t, i) - mean[j + (size_t)m * t];
      diff[j + (size_t)m * t] = d;
      mean[j + (size_t)m * t] += d * w[i] / tmp;
    }
    for (int t = 0; t < n1; t++)
      for (int c = 0; c < m; c++) for (int j = 0; j < m; j++)
        cov[j + m * (c + (size_t)m * t)] += w[i] * diff[j + (size_t)m * t] * (A_(c, t, i) - mean[c + (size_t)m * t]);
    cum = tmp;
  }
  for (size_t q = 0; q < mt * m; q++) cov[q] /= cum;
  free(diff);
}

/* summary.cpp:42-79 ; at m x (n+1), att m x n, Pt m x m x (n+1), Ptt m x m x n */
void orc_filter_summary(const double *alpha, int m, int n, unsigned N, const double *weights,
                        double *at, double *att, double *Pt, double *Ptt) {
  double *w = (double *)malloc(sizeof(double) * N);
  memset(at, 0, sizeof(double) * m * (n + 1));
  memset(att, 0, sizeof(double) * m * n);
  memset(Pt, 0, sizeof(double) * m * m * (n + 1));
  memset(Ptt, 0, sizeof(double) * m * m * n);
  for (int t = 0; t <= n; t++) {
    double acc = 0;
    if (t < n) {
      for (unsigned i = 0; i < N; i++) acc += weights[i + (size_t)N * t];
      for (unsigned i = 0; i < N; i++) w[i] = weights[i + (size_t)N * t] / acc;
    }
    for (unsigned i = 0; i < N; i++) for (int j = 0; j < m; j++) {
      if (t < n) att[j + (size_t)m * t] += A_(j, t, i) * w[i];
      at[j + (size_t)m * t] += A_(j, t, i);
    }
    for (int j = 0; j < m; j++) at[j + (size_t)m * t] /= N;
    for (unsigned i = 0; i < N; i++)
      for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) {
        Pt[j + m * (c + (size_t)m * t)] += (A_(j, t, i) - at[j + (size_t)m * t]) * (A_(c, t, i) - at[c + (size_t)m * t]);
        if (t < n)
          Ptt[j + m * (c + (size_t)m * t)] += w[i] * (A_(j, t, i) - att[j + (size_t)m * t]) * (A_(c, t, i) - att[c + (size_t)m * t]);
      }
    for (int q = 0; q < m * m; q++) Pt[q + (size_t)m * m * t] /= N;
  }
  free(w);
}

/* ------------------------------------------------------------------ */
/* IS post-correction: approx_mcmc.cpp:353-427 (psi), :447-518 (bsf)   */
/* ------------------------------------------------------------------ */
int orc_is_correction(const orc_is_job *job, double *weights, double *loglik_out,
                      double *alpha_out, double *alphahat, double *Vt) {
  const int ns = job->n_stored;
  if (ns <= 0) return 0;
  const int n = job->models[0].n, m = job->models[0].m;
  const size_t mt = (size_t)m * (n + 1);
  double *ah_all = NULL, *vt_all = NULL;
  if (job->output_type == 2) {
    ah_all = (double *)malloc(sizeof(double) * mt * ns);
    vt_all = (double *)malloc(sizeof(double) * mt * m * ns);
  }
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(job->n_threads > 0 ? job->n_threads : 1)
#endif
  for (int i = 0; i < ns; i++) {
    const orc_ung_model *mod = &job->models[i];
    unsigned nsimc = job->nsim;
    if (job->is_type == 1) nsimc *= job->counts[i]; /* :376-378 */
    double *alpha = (double *)calloc((size_t)m * (n + 1) * nsimc, sizeof(double));
    double *w = (double *)calloc((size_t)nsimc * (n + 1), sizeof(double));
    unsigned *idx = (unsigned *)calloc((size_t)nsimc * n, sizeof(unsigned));
    orc_rng rng = { 0, job->seed, (uint32_t)i, NULL, NULL, NULL, 0, 0 };
    double ll;
    if (job->method == 1) {
      orc_psi_setup *su = orc_psi_setup_alloc(n, m);
      orc_ung_psi_setup(mod, job->modes + (size_t)n * i, su); /* :372-373 */
      ll = orc_ung_psi_filter(mod, su, nsimc, &rng, alpha, w, idx);
      weights[i] = exp(ll); /* :385 */
      orc_psi_setup_free(su);
    } else {
      ll = orc_ung_bsf_filter(mod, nsimc, &rng, alpha, w, idx);
      weights[i] = exp(ll - job->approx_loglik[i]); /* :479 */
    }
    if (loglik_out) loglik_out[i] = ll;
    if (job->output_type != 3) {
      orc_filter_smoother(alpha, idx, m, n, nsimc); /* :388 */
      if (job->output_type == 1) { /* :390-392: uniform pick, stored transposed (n+1) x m */
        unsigned pick = (unsigned)(orc_rng_uniform(&rng, 3, 0, 0, 0) * nsimc);
        if (pick >= nsimc) pick = nsimc - 1;
        for (int t = 0; t <= n; t++) for (int j = 0; j < m; j++)
          alpha_out[t + (size_t)(n + 1) * (j + (size_t)m * i)] = A_(j, t, pick);
      } else {
        orc_summary(alpha, m, n + 1, nsimc, ah_all + mt * i, vt_all + mt * m * i); /* :396 */
      }
    }
    free(alpha); free(w); free(idx);
  }
  if (job->output_type == 2) { /* :397-414 merge (in theta order), :422-424 */
    double *Valpha = (double *)calloc(mt * m, sizeof(double));
    double *diff = (double *)malloc(sizeof(double) * mt);
    memset(alphahat, 0, sizeof(double) * mt);
    memset(Vt, 0, sizeof(double) * mt * m);
    double sum_w = 0.0;
    for (int i = 0; i < ns; i++) {
      const double *ahi = ah_all + mt * i, *vti = vt_all + mt * m * i;
      double wnew = weights[i] * job->counts[i];
      sum_w += wnew;
      for (size_t q = 0; q < mt; q++) { diff[q] = ahi[q] - alphahat[q]; alphahat[q] += wnew / sum_w * diff[q]; }
      for (int t = 0; t <= n; t++)
        for (int c = 0; c < m; c++) for (int j = 0; j < m; j++)
          Valpha[j + m * (c + (size_t)m * t)] += wnew * diff[j + (size_t)m * t] * (ahi[c + (size_t)m * t] - alphahat[c + (size_t)m * t]);
      for (size_t q = 0; q < mt * m; q++) Vt[q] += wnew / sum_w * (vti[q] - Vt[q]);
    }
    for (size_t q = 0; q < mt * m; q++) Vt[q] += Valpha[q] / sum_w;
    free(Valpha); free(diff); free(ah_all); free(vt_all);
  }
  return 0;
}


/* ================================================================== */
/* ssm_mng: multivariate observations                                  */
/* ================================================================== */
#define ORC_MAXP 16
#define MZ(t) (mod->Z + (size_t)mod->p * mod->m * ((t) * mod->Ztv))
#define MD(t) (mod->D + (size_t)mod->p * ((t) * mod->Dtv))

double orc_mng_const_term(const orc_mng_model *mod) { /* model_ssm_mng.cpp:632-666 */
  double res = 0.0;
  for (int i = 0; i < mod->p; i++)
    for (int t = 0; t < mod->n; t++) {
      double y = mod->y[i + (size_t)mod->p * t], u = mod->u[i + (size_t)mod->p * t], phi = mod->phi[i];
      if (!isfinite(y)) continue;
      switch (mod->distribution[i]) {
      case ORC_POISSON: res += y * log(u) - lgamma(y + 1); break;
      case ORC_BINOMIAL: res += orc_lchoose(u, y); break;
      case ORC_NEGBIN: res += orc_lchoose(y + phi - 1, y) + phi * log(phi) + y * log(u); break;
      case ORC_GAMMA: res += phi * log(phi) - lgamma(phi) + (phi - 1) * log(y) - phi * log(u); break;
      case ORC_GAUSSIAN: res += norm_log_const(phi); break;
      }
    }
  return res;
}

static int mng_observed(const orc_mng_model *mod, int t) { /* na_y.n_elem < p */
  for (int i = 0; i < mod->p; i++) if (isfinite(mod->y[i + (size_t)mod->p * t])) return 1;
  return 0;
}

/* model_ssm_mng.cpp:407-442 (su == NULL) and :357-393 (psi weights, su != NULL), one particle */
static double mng_log_weight(const orc_mng_model *mod, const orc_mng_setup *su, int t, const double *a) {
  const int p = mod->p, m = mod->m;
  const double *Zt = MZ(t), *Dt = MD(t);
  double w = 0.0;
  for (int j = 0; j < p; j++) {
    double y = mod->y[j + (size_t)p * t];
    if (!isfinite(y)) continue;
    double s = 0.0;
    for (int c = 0; c < m; c++) s += Zt[j + p * c] * a[c];
    s = Dt[j] + s;
    double u = mod->u[j + (size_t)p * t], phi = mod->phi[j];
    switch (mod->distribution[j]) {
    case ORC_POISSON: w += y * s - u * exp(s); break;
    case ORC_BINOMIAL: w += y * s - u * log1p(exp(s)); break;
    case ORC_NEGBIN: w += y * s - (y + phi) * log(phi + u * exp(s)); break;
    case ORC_GAMMA: w += -phi * (s + (y * exp(-s) / u)); break;
    case ORC_GAUSSIAN: w += -0.5 * pow((y - s) / phi, 2.0); break;
    }
    if (su) w += 0.5 * pow((su->approx_y[j + (size_t)p * t] - s) / su->approx_H[j + (size_t)p * t], 2.0);
  }
  return w;
}

static double mng_filter_core(const orc_mng_model *mod, const orc_mng_setup *su, unsigned N, const orc_rng *rng,
                              double *alpha, double *weights, unsigned *indices) {
  const int n = mod->n, m = mod->m, k = su ? m : mod->k, mm = m * m;
  double L[ORC_MAXM * ORC_MAXM];
  if (!su) ung_initial_chol(mod->P1, m, L); /* :565-570 */
  double *nw = (double *)malloc(sizeof(double) * N), *lw = (double *)malloc(sizeof(double) * N);
  double *r = (double *)malloc(sizeof(double) * N), *atmp = (double *)malloc(sizeof(double) * N * m);
  double um[ORC_MAXM];
  for (unsigned i = 0; i < N; i++) {
    rng_normals(rng, 0, 0, i, m, um);
    for (int j = 0; j < m; j++) {
      double s = 0.0;
      for (int c = 0; c < m; c++) s += (su ? su->Vt[j + m * c] : L[j + m * c]) * um[c];
      A_(j, 0, i) = (su ? su->alphahat[j] : mod->a1[j]) + s;
    }
  }
  double loglik = su ? su->approx_loglik : 0.0, inc;
  if (mng_observed(mod, 0)) {
    for (unsigned i = 0; i < N; i++) lw[i] = mng_log_weight(mod, su, 0, &A_(0, 0, i)) - (su ? su->scales[0] : 0.0);
    if (!normalise_step(lw, N, weights, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
    loglik += inc;
  } else {
    for (unsigned i = 0; i < N; i++) { weights[i] = 1.0; nw[i] = 1.0 / N; }
  }
  for (int t = 0; t < n; t++) {
    for (unsigned i = 0; i < N; i++) r[i] = orc_rng_uniform(rng, 1, t, i, (uint32_t)k);
    filter_resample(nw, r, N, indices + (size_t)N * t, t);
    for (unsigned i = 0; i < N; i++)
      for (int j = 0; j < m; j++) atmp[j + (size_t)m * i] = A_(j, t, indices[i + (size_t)N * t]);
    const double *Tt = mod->T + (size_t)mm * (t * mod->Ttv), *Rt = mod->R + (size_t)m * mod->k * (t * mod->Rtv);
    const double *Cv = mod->C + (size_t)m * (t * mod->Ctv);
    for (unsigned i = 0; i < N; i++) {
      rng_normals(rng, 2, t, i, k, um);
      for (int j = 0; j < m; j++) {
        double s = 0.0, q = 0.0;
        if (su) { /* :527-534 */
          const double *ah0 = su->alphahat + (size_t)m * t, *Ct1 = su->Ct + (size_t)mm * (t + 1), *Vt1 = su->Vt + (size_t)mm * (t + 1);
          for (int c = 0; c < m; c++) s += Ct1[j + m * c] * (atmp[c + (size_t)m * i] - ah0[c]);
          for (int c = 0; c < m; c++) q += Vt1[j + m * c] * um[c];
          A_(j, t + 1, i) = su->alphahat[j + (size_t)m * (t + 1)] + s + q;
        } else { /* :611-618 */
          for (int c = 0; c < m; c++) s += Tt[j + m * c] * atmp[c + (size_t)m * i];
          for (int c = 0; c < k; c++) q += Rt[j + m * c] * um[c];
          A_(j, t + 1, i) = Cv[j] + s + q;
        }
      }
    }
    double *wcol = weights + (size_t)N * (t + 1);
    if ((t < n - 1) && mng_observed(mod, t + 1)) {
      for (unsigned i = 0; i < N; i++) lw[i] = mng_log_weight(mod, su, t + 1, &A_(0, t + 1, i)) - (su ? su->scales[t + 1] : 0.0);
      if (!normalise_step(lw, N, wcol, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
      loglik += inc;
    } else {
      for (unsigned i = 0; i < N; i++) { wcol[i] = 1.0; nw[i] = 1.0 / N; }
    }
  }
  if (!su) loglik += orc_mng_const_term(mod); /* :632-666 */
done:
  free(nw); free(lw); free(r); free(atmp);
  return loglik;
}

double orc_mng_bsf_filter(const orc_mng_model *mod, unsigned N, const orc_rng *rng, double *alpha, double *weights, unsigned *indices) {
  return mng_filter_core(mod, NULL, N, rng, alpha, weights, indices);
}
double orc_mng_psi_filter(const orc_mng_model *mod, const orc_mng_setup *su, unsigned N, const orc_rng *rng,
                          double *alpha, double *weights, unsigned *indices) {
  return mng_filter_core(mod, su, N, rng, alpha, weights, indices);
}

orc_mng_setup *orc_mng_setup_alloc(int n, int m, int p) {
  orc_mng_setup *s = (orc_mng_setup *)calloc(1, sizeof(orc_mng_setup));
  s->n = n; s->m = m; s->p = p;
  s->approx_y = (double *)calloc((size_t)p * n, sizeof(double));
  s->approx_H = (double *)calloc((size_t)p * n, sizeof(double));
  s->mode = (double *)calloc((size_t)p * n, sizeof(double));
  s->scales = (double *)calloc(n, sizeof(double));
  s->alphahat = (double *)calloc((size_t)m * (n + 1), sizeof(double));
  s->Vt = (double *)calloc((size_t)m * m * (n + 1), sizeof(double));
  s->Ct = (double *)calloc((size_t)m * m * (n + 1), sizeof(double));
  return s;
}
void orc_mng_setup_free(orc_mng_setup *s) {
  if (!s) return;
  free(s->approx_y); free(s->approx_H); free(s->mode); free(s->scales); free(s->alphahat); free(s->Vt); free(s->Ct); free(s);
}

/* one multivariate Kalman prediction step, the "zeroed rows" form of src/model_ssm_mlg.cpp:259-291,318-356
 * (numerically the same update as the observed-rows form of :103-128: missing rows decouple exactly).
 * gy, gH: p x n pseudo-observations and their sd.  ZFinv (m x p) = Z' F^-1, K = P ZFinv. */
/* observation-noise covariance of a genuine ssm_mlg model (p x p x (1|n), HH = H H'): set by orc_mlg_psi_filter around
 * its smoother call; NULL = the diagonal pseudo-observation noise gH of the Gaussian approximations */
static const double *g_mlg_HH = NULL;
static int g_mlg_HHtv = 0;
static int mlg_step(const orc_mng_model *mod, const double *gy, const double *gH, int t, const double *at, const double *Pt,
                    double *at1, double *Pt1, double *v, double *ZFinv, double *K, double *Zz, int *nobs, double *logdetF) {
  const int m = mod->m, p = mod->p;
  const double *Zt = MZ(t), *Dt = MD(t);
  const double *Tt = mod->T + (size_t)m * m * (t * mod->Ttv), *Rt = mod->R + (size_t)m * mod->k * (t * mod->Rtv);
  const double *Cv = mod->C + (size_t)m * (t * mod->Ctv);
  double RR[ORC_MAXM * ORC_MAXM], tmp[ORC_MAXM * ORC_MAXM], tmp2[ORC_MAXM * ORC_MAXM];
  compute_RR(Rt, m, mod->k, RR);
  int na[ORC_MAXP], cnt = 0;
  for (int i = 0; i < p; i++) { na[i] = !isfinite(gy[i + (size_t)p * t]); cnt += !na[i]; }
  *nobs = cnt;
  if (cnt == 0) {
    for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += Tt[i + m * j] * at[j]; at1[i] = Cv[i] + s; }
    mat_mul(Tt, Pt, m, m, m, tmp); mat_mul_bt(tmp, Tt, m, m, m, tmp2);
    for (int q = 0; q < m * m; q++) Pt1[q] = tmp2[q] + RR[q];
    symmatu(Pt1, m);
    return 1;
  }
  double HH[ORC_MAXP * ORC_MAXP] = {0}, F[ORC_MAXP * ORC_MAXP], Lf[ORC_MAXP * ORC_MAXP] = {0}, Fi[ORC_MAXP * ORC_MAXP];
  for (int c = 0; c < m; c++) for (int i = 0; i < p; i++) Zz[i + p * c] = na[i] ? 0.0 : Zt[i + p * c];
  if (g_mlg_HH) { /* HHt.rows(na).zeros(); HHt.cols(na).zeros(); HHt(na, na) = I: src/model_ssm_mlg.cpp:324-328 */
    const double *Hs = g_mlg_HH + (size_t)p * p * (t * g_mlg_HHtv);
    for (int j = 0; j < p; j++) for (int i = 0; i < p; i++) HH[i + p * j] = (na[i] || na[j]) ? (i == j ? 1.0 : 0.0) : Hs[i + p * j];
  } else {
    for (int i = 0; i < p; i++) HH[i + p * i] = na[i] ? 1.0 : gH[i + (size_t)p * t] * gH[i + (size_t)p * t];
  }
  double ZP[ORC_MAXP * ORC_MAXM];
  mat_mul(Zz, Pt, p, m, m, ZP);
  mat_mul_bt(ZP, Zz, p, m, p, F);
  for (int q = 0; q < p * p; q++) F[q] += HH[q];
  for (int q = 0; q < p * p; q++) if (!isfinite(F[q])) return 0;
  for (int i = 0; i < p; i++) if (!(F[i + p * i] > 0)) return 0;
  int idx[ORC_MAXP]; for (int i = 0; i < p; i++) idx[i] = i;
  if (chol_lower_sub(F, p, idx, p, Lf) != 0) return 0;
  *logdetF = 0; for (int i = 0; i < p; i++) *logdetF += 2.0 * log(Lf[i + p * i]);
  /* F^-1 = L^-T L^-1 */
  double Li[ORC_MAXP * ORC_MAXP] = {0};
  for (int j = 0; j < p; j++) { /* invert lower triangular */
    Li[j + p * j] = 1.0 / Lf[j + p * j];
    for (int i = j + 1; i < p; i++) { double s = 0; for (int q = j; q < i; q++) s += Lf[i + p * q] * Li[q + p * j]; Li[i + p * j] = -s / Lf[i + p * i]; }
  }
  mat_mul_at(Li, Li, p, p, p, Fi);
  for (int i = 0; i < p; i++) {
    double za = 0; for (int c = 0; c < m; c++) za += Zz[i + p * c] * at[c];
    v[i] = na[i] ? 0.0 : gy[i + (size_t)p * t] - Dt[i] - za;
  }
  mat_mul_at(Zz, Fi, m, p, p, ZFinv);      /* Z' F^-1 : m x p */
  mat_mul(Pt, ZFinv, m, m, p, K);          /* m x p */
  double af[ORC_MAXM];
  for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < p; j++) s += K[i + m * j] * v[j]; af[i] = at[i] + s; }
  for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += Tt[i + m * j] * af[j]; at1[i] = Cv[i] + s; }
  double IKZ[ORC_MAXM * ORC_MAXM], KH[ORC_MAXM * ORC_MAXP], M[ORC_MAXM * ORC_MAXM];
  mat_mul(K, Zz, m, p, m, IKZ);
  for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) IKZ[i + m * j] = (i == j ? 1.0 : 0.0) - IKZ[i + m * j];
  mat_mul(IKZ, Pt, m, m, m, tmp); mat_mul_bt(tmp, IKZ, m, m, m, M);
  mat_mul(K, HH, m, p, p, KH); mat_mul_bt(KH, K, m, p, m, tmp);
  for (int q = 0; q < m * m; q++) M[q] += tmp[q];
  mat_mul(Tt, M, m, m, m, tmp); mat_mul_bt(tmp, Tt, m, m, m, tmp2);
  for (int q = 0; q < m * m; q++) Pt1[q] = tmp2[q] + RR[q];
  symmatu(Pt1, m);
  return 1;
}

double orc_mlg_loglik(const orc_mng_model *mod, const double *gy, const double *gH) { /* model_ssm_mlg.cpp:90-137 */
  const int n = mod->n, m = mod->m, p = mod->p;
  double at[ORC_MAXM], Pt[ORC_MAXM * ORC_MAXM], at1[ORC_MAXM], Pt1[ORC_MAXM * ORC_MAXM];
  double v[ORC_MAXP], ZF[ORC_MAXM * ORC_MAXP], K[ORC_MAXM * ORC_MAXP], Zz[ORC_MAXP * ORC_MAXM];
  /* :93 `if(arma::accu(H) + arma::accu(R) < zero_tol) logLik = -inf` -- the SUMS of the entries (of the approximating model's H cube: the standard
   * deviations on its diagonals; of R, signs and all): a model whose R sums below -accu(H) has log-likelihood -inf in the reference.  Missing from this
   * restatement until round 6 (found by scripts/fuzz_parity.py: the product's host code had the line, the oracle did not). */
  {
    double accH = 0, accR = 0;
    for (size_t q = 0; q < (size_t)p * n; q++) accH += gH[q];
    for (size_t q = 0; q < (size_t)m * mod->k * (mod->Rtv ? n : 1); q++) accR += mod->R[q];
    if (accH + accR < ORC_ZERO_TOL) return -INFINITY;
  }
  memcpy(at, mod->a1, sizeof(double) * m); memcpy(Pt, mod->P1, sizeof(double) * m * m);
  const double LOG2PI = log(2.0 * ORC_PI);
  double ll = 0.0;
  for (int t = 0; t < n; t++) {
    int nobs; double ld = 0;
    if (!mlg_step(mod, gy, gH, t, at, Pt, at1, Pt1, v, ZF, K, Zz, &nobs, &ld)) return -INFINITY;
    if (nobs > 0) {
      /* v' F^-1 v = |L^-1 v|^2 with F = L L' (the reference forms inv(chol(F))' v, :125-127) */
      double q = 0;
      double F[ORC_MAXP * ORC_MAXP], HH, ZP[ORC_MAXP * ORC_MAXM], Lf[ORC_MAXP * ORC_MAXP] = {0}, w[ORC_MAXP];
      mat_mul(Zz, Pt, p, m, m, ZP); mat_mul_bt(ZP, Zz, p, m, p, F);
      for (int i = 0; i < p; i++) { HH = isfinite(gy[i + (size_t)p * t]) ? gH[i + (size_t)p * t] * gH[i + (size_t)p * t] : 1.0; F[i + p * i] += HH; }
      int idx[ORC_MAXP]; for (int i = 0; i < p; i++) idx[i] = i;
      chol_lower_sub(F, p, idx, p, Lf);
      for (int i = 0; i < p; i++) { double s = v[i]; for (int c = 0; c < i; c++) s -= Lf[i + p * c] * w[c]; w[i] = s / Lf[i + p * i]; q += w[i] * w[i]; }
      ll -= 0.5 * (nobs * LOG2PI + ld + q);
    }
    memcpy(at, at1, sizeof(double) * m); memcpy(Pt, Pt1, sizeof(double) * m * m);
  }
  return ll;
}

/* model_ssm_mlg.cpp fast_smoother (:225-308 region) */
static int mlg_fast_smoother(const orc_mng_model *mod, const double *gy, const double *gH, double *at) {
  const int n = mod->n, m = mod->m, p = mod->p;
  double *vt = (double *)calloc((size_t)p * n, sizeof(double)), *ZF = (double *)calloc((size_t)m * p * n, sizeof(double));
  double *Kt = (double *)calloc((size_t)m * p * n, sizeof(double)), *Zz = (double *)calloc((size_t)p * m * n, sizeof(double));
  int *nob = (int *)calloc(n, sizeof(int));
  double *rt = (double *)calloc((size_t)m * n, sizeof(double));
  double Pt[ORC_MAXM * ORC_MAXM], Pt1[ORC_MAXM * ORC_MAXM], ld;
  int ok = 1;
  memcpy(at, mod->a1, sizeof(double) * m); memcpy(Pt, mod->P1, sizeof(double) * m * m);
  for (int t = 0; t < n && ok; t++) {
    ok = mlg_step(mod, gy, gH, t, at + (size_t)m * t, Pt, at + (size_t)m * (t + 1), Pt1, vt + (size_t)p * t, ZF + (size_t)m * p * t,
                  Kt + (size_t)m * p * t, Zz + (size_t)p * m * t, &nob[t], &ld);
    memcpy(Pt, Pt1, sizeof(double) * m * m);
  }
  if (ok) {
    double L[ORC_MAXM * ORC_MAXM], KZ[ORC_MAXM * ORC_MAXM];
    for (int t = n - 1; t >= 0; t--) {
      const double *Tt = mod->T + (size_t)m * m * (t * mod->Ttv);
      double rnew[ORC_MAXM];
      const double *r1 = rt + (size_t)m * t; /* r_t (r_{n-1} = 0) */
      if (nob[t] > 0) {
        mat_mul(Kt + (size_t)m * p * t, Zz + (size_t)p * m * t, m, p, m, KZ);
        for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) KZ[i + m * j] = (i == j ? 1.0 : 0.0) - KZ[i + m * j];
        mat_mul(Tt, KZ, m, m, m, L);
        for (int i = 0; i < m; i++) {
          double s = 0, q = 0;
          for (int j = 0; j < p; j++) s += ZF[i + m * j + (size_t)m * p * t] * vt[j + (size_t)p * t];
          for (int j = 0; j < m; j++) q += L[j + m * i] * r1[j];
          rnew[i] = s + q;
        }
      } else {
        for (int i = 0; i < m; i++) { double q = 0; for (int j = 0; j < m; j++) q += Tt[j + m * i] * r1[j]; rnew[i] = q; }
      }
      if (t > 0) memcpy(rt + (size_t)m * (t - 1), rnew, sizeof(double) * m);
      else { /* :292-299: a_0 = a1 + P1 * r_{-1} */
        for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += mod->P1[i + m * j] * rnew[j]; at[i] = mod->a1[i] + s; }
      }
    }
    for (int t = 0; t < n - 1; t++) {
      double RR[ORC_MAXM * ORC_MAXM];
      compute_RR(mod->R + (size_t)m * mod->k * (t * mod->Rtv), m, mod->k, RR);
      const double *Tt = mod->T + (size_t)m * m * (t * mod->Ttv), *Cv = mod->C + (size_t)m * (t * mod->Ctv);
      for (int i = 0; i < m; i++) {
        double s = 0, q = 0;
        for (int j = 0; j < m; j++) { s += Tt[i + m * j] * at[j + (size_t)m * t]; q += RR[i + m * j] * rt[j + (size_t)m * t]; }
        at[i + (size_t)m * (t + 1)] = Cv[i] + s + q;
      }
    }
  }
  free(vt); free(ZF); free(Kt); free(Zz); free(nob); free(rt);
  return ok;
}

/* model_ssm_mlg.cpp:310-396 */
static int mlg_smoother_ccov(const orc_mng_model *mod, const double *gy, const double *gH, double *at, double *Pt, double *ccov) {
  const int n = mod->n, m = mod->m, p = mod->p, mm = m * m;
  double *vt = (double *)calloc((size_t)p * n, sizeof(double)), *ZF = (double *)calloc((size_t)m * p * n, sizeof(double));
  double *Kt = (double *)calloc((size_t)m * p * n, sizeof(double)), *Zz = (double *)calloc((size_t)p * m * n, sizeof(double));
  int *nob = (int *)calloc(n, sizeof(int));
  double ld;
  int ok = 1;
  memcpy(at, mod->a1, sizeof(double) * m); memcpy(Pt, mod->P1, sizeof(double) * mm);
  for (int t = 0; t < n && ok; t++) {
    ok = mlg_step(mod, gy, gH, t, at + (size_t)m * t, Pt + (size_t)mm * t, at + (size_t)m * (t + 1), Pt + (size_t)mm * (t + 1),
                  vt + (size_t)p * t, ZF + (size_t)m * p * t, Kt + (size_t)m * p * t, Zz + (size_t)p * m * t, &nob[t], &ld);
    memcpy(ccov + (size_t)mm * t, Pt + (size_t)mm * (t + 1), sizeof(double) * mm);
  }
  if (ok) {
    double rt[ORC_MAXM] = {0}, Nt[ORC_MAXM * ORC_MAXM] = {0};
    double L[ORC_MAXM * ORC_MAXM], A[ORC_MAXM * ORC_MAXM], B[ORC_MAXM * ORC_MAXM], Cm[ORC_MAXM * ORC_MAXM], rn[ORC_MAXM];
    for (int t = n - 1; t >= 0; t--) {
      const double *Tt = mod->T + (size_t)mm * (t * mod->Ttv);
      double *Ptt = Pt + (size_t)mm * t, *cc = ccov + (size_t)mm * t;
      mat_mul(Nt, cc, m, m, m, A);
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) A[i + m * j] = (i == j ? 1.0 : 0.0) - A[i + m * j];
      if (nob[t] > 0) {
        mat_mul(Kt + (size_t)m * p * t, Zz + (size_t)p * m * t, m, p, m, B);
        for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) B[i + m * j] = (i == j ? 1.0 : 0.0) - B[i + m * j];
        mat_mul(Tt, B, m, m, m, L);
        mat_mul_bt(Ptt, L, m, m, m, B); mat_mul(B, A, m, m, m, Cm); memcpy(cc, Cm, sizeof(double) * mm);
        for (int i = 0; i < m; i++) {
          double s = 0, q = 0;
          for (int j = 0; j < p; j++) s += ZF[i + m * j + (size_t)m * p * t] * vt[j + (size_t)p * t];
          for (int j = 0; j < m; j++) q += L[j + m * i] * rt[j];
          rn[i] = s + q;
        }
        memcpy(rt, rn, sizeof(double) * m);
        mat_mul_at(L, Nt, m, m, m, A); mat_mul(A, L, m, m, m, B);
        mat_mul(ZF + (size_t)m * p * t, Zz + (size_t)p * m * t, m, p, m, A); /* ZFinv * Zt */
        for (int q = 0; q < mm; q++) Nt[q] = A[q] + B[q];
      } else {
        mat_mul_bt(Ptt, Tt, m, m, m, B); mat_mul(B, A, m, m, m, Cm); memcpy(cc, Cm, sizeof(double) * mm);
        for (int i = 0; i < m; i++) { double q = 0; for (int j = 0; j < m; j++) q += Tt[j + m * i] * rt[j]; rn[i] = q; }
        memcpy(rt, rn, sizeof(double) * m);
        mat_mul_at(Tt, Nt, m, m, m, A); mat_mul(A, Tt, m, m, m, B); memcpy(Nt, B, sizeof(double) * mm);
      }
      symmatu(Nt, m);
      for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += Ptt[i + m * j] * rt[j]; at[i + (size_t)m * t] += s; }
      mat_mul(Ptt, Nt, m, m, m, A); mat_mul(A, Ptt, m, m, m, B); symmatu(B, m);
      for (int q = 0; q < mm; q++) Ptt[q] -= B[q];
    }
    memset(ccov + (size_t)mm * n, 0, sizeof(double) * mm);
  }
  free(vt); free(ZF); free(Kt); free(Zz); free(nob);
  return ok;
}

static void mng_laplace_iter(const orc_mng_model *mod, const double *signal, double *ay, double *aH) { /* :269-315 */
  const int p = mod->p;
  for (int t = 0; t < mod->n; t++)
    for (int i = 0; i < p; i++) {
      size_t q = i + (size_t)p * t;
      double y = mod->y[q], sg = signal[q], u = mod->u[q], phi = mod->phi[i], HH = 0, yy = 0;
      switch (mod->distribution[i]) {
      case ORC_POISSON: HH = 1.0 / (exp(sg) * u); yy = y * HH + sg - 1.0; break;
      case ORC_BINOMIAL: { double e = exp(sg); HH = (1.0 + e) * (1.0 + e) / (u * e); yy = y * HH + sg - 1.0 - e; } break;
      case ORC_NEGBIN: { double e = exp(sg) * u; HH = (phi + e) * (phi + e) / (phi * e * (y + phi)); yy = sg + (phi + e) * (y - e) / ((y + phi) * e); } break;
      case ORC_GAMMA: { double e = exp(sg) * u; HH = e / (y * phi); yy = sg - e / y + 1; } break;
      case ORC_GAUSSIAN: HH = phi * phi; yy = y; break;
      }
      ay[q] = yy; aH[q] = sqrt(HH);
    }
}

static void mng_update_scales(const orc_mng_model *mod, orc_mng_setup *su) { /* :218-255 */
  const int p = mod->p;
  for (int t = 0; t < mod->n; t++) {
    su->scales[t] = 0.0;
    for (int i = 0; i < p; i++) {
      size_t q = i + (size_t)p * t;
      double y = mod->y[q];
      if (!isfinite(y)) continue;
      double md = su->mode[q], u = mod->u[q], phi = mod->phi[i];
      switch (mod->distribution[i]) {
      case ORC_POISSON: su->scales[t] += y * md - u * exp(md); break;
      case ORC_BINOMIAL: su->scales[t] += y * md - u * log1p(exp(md)); break;
      case ORC_NEGBIN: su->scales[t] += y * md - (y + phi) * log(phi + u * exp(md)); break;
      case ORC_GAMMA: su->scales[t] += -phi * md - (y * phi * exp(-md) / u); break;
      case ORC_GAUSSIAN: su->scales[t] += -0.5 * pow((y - md) / phi, 2.0); break;
      }
      su->scales[t] += 0.5 * pow((su->approx_y[q] - md) / su->approx_H[q], 2.0);
    }
  }
}

int orc_mng_psi_setup(const orc_mng_model *mod, const double *is_mode, orc_mng_setup *su) {
  const int n = mod->n, m = mod->m, p = mod->p;
  double *sm = (double *)malloc(sizeof(double) * m * (n + 1));
  int ok = 1;
  if (is_mode) { /* approximate_for_is :117-133 */
    memcpy(su->mode, is_mode, sizeof(double) * p * n);
    mng_laplace_iter(mod, su->mode, su->approx_y, su->approx_H);
    mng_update_scales(mod, su);
    su->approx_loglik = 0.0; su->laplace_iters = 0;
  } else { /* approximate() :73-114 */
    memcpy(su->mode, mod->initial_mode, sizeof(double) * p * n);
    int it = 0; double diff = mod->conv_tol + 1;
    double *nm = (double *)malloc(sizeof(double) * p * n);
    while (it < mod->max_iter && diff > mod->conv_tol) {
      it++;
      mng_laplace_iter(mod, su->mode, su->approx_y, su->approx_H);
      ok = mlg_fast_smoother(mod, su->approx_y, su->approx_H, sm);
      if (!ok) break;
      for (int t = 0; t < n; t++) {
        const double *Zt = MZ(t), *Dt = MD(t);
        for (int j = 0; j < p; j++) { double s = 0; for (int c = 0; c < m; c++) s += Zt[j + p * c] * sm[c + (size_t)m * t]; nm[j + (size_t)p * t] = Dt[j] + s; }
      }
      diff = 0; for (size_t q = 0; q < (size_t)p * n; q++) diff += (nm[q] - su->mode[q]) * (nm[q] - su->mode[q]);
      diff /= (double)n * p;
      memcpy(su->mode, nm, sizeof(double) * p * n);
    }
    free(nm);
    su->laplace_iters = it;
    double gll = orc_mlg_loglik(mod, su->approx_y, su->approx_H);
    mng_update_scales(mod, su);
    double ct = orc_mng_const_term(mod); /* compute_const_term :318-355 */
    for (int i = 0; i < p; i++) for (int t = 0; t < n; t++) if (isfinite(mod->y[i + (size_t)p * t])) ct -= norm_log_const(su->approx_H[i + (size_t)p * t]);
    double ssc = 0; for (int t = 0; t < n; t++) ssc += su->scales[t];
    su->approx_loglik = gll + ct + ssc;
  }
  if (ok) ok = mlg_smoother_ccov(mod, su->approx_y, su->approx_H, su->alphahat, su->Vt, su->Ct);
  if (ok) orc_conditional_cov(su->Vt, su->Ct, m, n + 1);
  free(sm);
  return ok ? 0 : -1;
}


/* ================================================================== */
/* ssm_nlg / ssm_sde bootstrap filters on the shipped model functors   */
/* ================================================================== */
static void fn_a1P1(const orc_fn_model *mod, double *a1, double *P1) { /* a1_fn / P1_fn of each snippet */
  const double *th = mod->theta, *kp = mod->known;
  a1[0] = a1[1] = 0; P1[0] = P1[1] = P1[2] = P1[3] = 0;
  switch (mod->id) {
  case 1: a1[0] = kp[2]; a1[1] = kp[3]; P1[0] = kp[4]; P1[3] = kp[5]; break;          /* R/cpp_example_models.R:351-366 */
  case 2: a1[0] = th[0]; P1[0] = pow(exp(th[2]), 2) / (1 - pow(th[1], 2)); break;     /* :210-225 */
  default: a1[0] = 0; P1[0] = 1; break;                                               /* :528-541, :651-664 */
  }
}
static double fn_H(const orc_fn_model *mod) { /* H_fn */
  switch (mod->id) { case 1: return exp(mod->theta[0]); case 2: return exp(mod->theta[3]); default: return exp(mod->theta[0]); }
}
static void fn_T(const orc_fn_model *mod, const double *a, double *x) { /* T_fn */
  const double *th = mod->theta, *kp = mod->known;
  switch (mod->id) {
  case 1: { /* :415-428 */
    double dT = kp[0], K = kp[1];
    x[0] = a[0];
    double r = exp(a[0]) / (1.0 + exp(a[0]));
    x[1] = K * a[1] * exp(r * dT) / (K + a[1] * (exp(r * dT) - 1));
  } break;
  case 2: x[0] = th[0] * (1 - th[1]) + th[1] * a[0]; break;
  case 3: x[0] = sin(a[0]); break;
  default: x[0] = a[0]; break;
  }
}
static void fn_Rdiag(const orc_fn_model *mod, double *r) { /* R_fn (diagonal in every shipped model) */
  const double *th = mod->theta;
  r[0] = r[1] = 0;
  switch (mod->id) { case 1: r[0] = exp(th[1]); r[1] = exp(th[2]); break; case 2: r[0] = exp(th[2]); break;
                     case 3: r[0] = exp(th[1]); break; default: r[0] = 1.0; break; }
}
static double fn_Z(const orc_fn_model *mod, const double *x) { /* Z_fn */
  switch (mod->id) { case 1: return x[1]; case 2: case 3: return exp(x[0]); default: return x[0]; }
}
static double sde_drift(const orc_fn_model *mod, double x) { return mod->id == 5 ? mod->theta[0] * x : exp(mod->theta[0]) * (mod->theta[1] - x); }
static double sde_diff(const orc_fn_model *mod, double x) { return mod->id == 5 ? fmax(0.0, mod->theta[1] * x) : exp(mod->theta[2]); }
static double sde_ddiff(const orc_fn_model *mod, double x) { return mod->id == 5 ? mod->theta[1] * (x > 0.0) : 0.0; }
/* milstein.cpp:18-48; increments N(0, sqrt(dt)) from Box-Muller pairs d = 0.. of (purpose, t, particle) */
static double fn_milstein_t(const orc_fn_model *mod, const orc_rng *rng, uint32_t purpose, uint32_t t, uint32_t i, double x, double t_end) {
  unsigned nsub = 1u << mod->L;
  double dt = t_end / nsub, sdt = sqrt(dt);
  for (unsigned k = 0; k < nsub; k++) {
    double z0, z1;
    orc_rng_normal_pair(rng, purpose, t, i, k / 2, &z0, &z1);
    double dB = ((k & 1) ? z1 : z0) * sdt;
    x += sde_drift(mod, x) * dt + sde_diff(mod, x) * dB + 0.5 * sde_diff(mod, x) * sde_ddiff(mod, x) * (dB * dB - dt);
    if (mod->positive) x = fabs(x);
  }
  return x;
}
static double fn_milstein(const orc_fn_model *mod, const orc_rng *rng, uint32_t purpose, uint32_t t, uint32_t i, double x) {
  return fn_milstein_t(mod, rng, purpose, t, i, x, 1.0); /* the filters step one unit of time, model_ssm_sde.cpp:27,61 */
}
/* R_milstein (R_milstein.cpp:4-22): one path from x0 over [0, t_end]; stream = the prior draw of particle 0 */
double orc_milstein(const orc_fn_model *mod, double t_end, const orc_rng *rng) { return fn_milstein_t(mod, rng, 0, 0, 0, mod->x0, t_end); }
static double fn_log_obs(const orc_fn_model *mod, int t, const double *x) {
  double y = mod->y[t];
  if (mod->id == 6) return y * x[0] - exp(x[0]) - lgamma(y + 1.0); /* R::dpois(y, exp(x), log) */
  double mu = mod->id == 5 ? log(x[0]) : fn_Z(mod, x), sd = mod->id == 5 ? mod->theta[2] : fn_H(mod);
  return -0.5 * (log(2.0 * ORC_PI) + 2.0 * log(sd) + pow((y - mu) / sd, 2)); /* dmvnorm.cpp:10-13 (lwr) = R::dnorm log */
}

double orc_fn_bsf_filter(const orc_fn_model *mod, unsigned N, const orc_rng *rng, double *alpha, double *weights, unsigned *indices) {
  const int n = mod->n, m = mod->id == 1 ? 2 : 1, sde = mod->id >= 5;
  double a1[2], P1[4], L[4], rdiag[2], um[2];
  double *nw = (double *)malloc(sizeof(double) * N), *lw = (double *)malloc(sizeof(double) * N);
  double *r = (double *)malloc(sizeof(double) * N), *atmp = (double *)malloc(sizeof(double) * N * m);
  fn_a1P1(mod, a1, P1);
  if (!sde) { double Pm[4] = {P1[0], 0, 0, P1[3]}; if (m == 1) Pm[0] = P1[0]; orc_psd_chol(m == 1 ? P1 : Pm, m, L); fn_Rdiag(mod, rdiag); }
  for (unsigned i = 0; i < N; i++) {
    if (sde) A_(0, 0, i) = fn_milstein(mod, rng, 0, 0, i, mod->x0); /* model_ssm_sde.cpp:25-28 */
    else {
      rng_normals_wide(rng, 0, 0, i, m, um);
      for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < m; c++) s += L[j + m * c] * um[c]; A_(j, 0, i) = a1[j] + s; }
    }
  }
  double loglik = 0.0, inc;
  if (isfinite(mod->y[0])) {
    for (unsigned i = 0; i < N; i++) lw[i] = fn_log_obs(mod, 0, &A_(0, 0, i));
    if (!normalise_step(lw, N, weights, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
    loglik = inc;
  } else for (unsigned i = 0; i < N; i++) { weights[i] = 1.0; nw[i] = 1.0 / N; }
  for (int t = 0; t < n; t++) {
    for (unsigned i = 0; i < N; i++) r[i] = orc_rng_uniform(rng, 1, t, i, 0); /* functor-model filters draw WIDE blocks */
    filter_resample(nw, r, N, indices + (size_t)N * t, t);
    for (unsigned i = 0; i < N; i++) for (int j = 0; j < m; j++) atmp[j + (size_t)m * i] = A_(j, t, indices[i + (size_t)N * t]);
    for (unsigned i = 0; i < N; i++) {
      if (sde) A_(0, t + 1, i) = fn_milstein(mod, rng, 4, t, i, atmp[i]); /* :59-62 */
      else {
        double x[2];
        rng_normals_wide(rng, 2, t, i, m, um);
        fn_T(mod, atmp + (size_t)m * i, x);
        for (int j = 0; j < m; j++) A_(j, t + 1, i) = x[j] + rdiag[j] * um[j]; /* model_ssm_nlg.cpp:1118-1126 */
      }
    }
    double *wcol = weights + (size_t)N * (t + 1);
    if ((t < n - 1) && isfinite(mod->y[t + 1])) {
      for (unsigned i = 0; i < N; i++) lw[i] = fn_log_obs(mod, t + 1, &A_(0, t + 1, i));
      if (!normalise_step(lw, N, wcol, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
      loglik += inc;
    } else for (unsigned i = 0; i < N; i++) { wcol[i] = 1.0; nw[i] = 1.0 / N; }
  }
done:
  free(nw); free(lw); free(r); free(atmp);
  return loglik;
}


/* ------------------------------------------------------------------ */
/* ssm_nlg::bsf_filter for ANY model behind function pointers -- the reference's own interface (src/model_ssm_nlg.h:10-21:
 * Z_fn, H_fn, T_fn, R_fn take (t, alpha, theta, known_params); H and R may depend on the state) -- restating
 * src/model_ssm_nlg.cpp:1066-1146 with log_obs_density (:943-956) for one observation series.  m <= ORC_MAXM, k <= m. */
double orc_nlg_general_bsf_filter(const orc_nlg_general *mod, unsigned N, const orc_rng *rng, double *alpha, double *weights,
                                  unsigned *indices) {
  const int n = mod->n, m = mod->m, k = mod->k;
  double L[ORC_MAXM * ORC_MAXM], um[ORC_MAXM], x[ORC_MAXM], Rm[ORC_MAXM * ORC_MAXM];
  double *nw = (double *)malloc(sizeof(double) * N), *lw = (double *)malloc(sizeof(double) * N);
  double *r = (double *)malloc(sizeof(double) * N), *atmp = (double *)malloc(sizeof(double) * N * m);
  orc_psd_chol(mod->P1, m, L); /* :1071 */
  for (unsigned i = 0; i < N; i++) {
    rng_normals_wide(rng, 0, 0, i, m, um);
    for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < m; c++) s += L[j + m * c] * um[c]; A_(j, 0, i) = mod->a1[j] + s; }
  }
  double loglik = 0.0, inc;
#define NLG_LOG_OBS(t_, xp_, out_) do { double z_, h_; mod->Z_fn((t_), (xp_), mod->theta, mod->known, &z_); \
    mod->H_fn((t_), (xp_), mod->theta, mod->known, &h_); \
    (out_) = -0.5 * (log(2.0 * ORC_PI) + 2.0 * log(h_) + pow((mod->y[t_] - z_) / h_, 2)); } while (0) /* dmvnorm.cpp:10-13 (lwr) */
  if (isfinite(mod->y[0])) {
    for (unsigned i = 0; i < N; i++) NLG_LOG_OBS(0, &A_(0, 0, i), lw[i]);
    if (!normalise_step(lw, N, weights, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
    loglik = inc;
  } else for (unsigned i = 0; i < N; i++) { weights[i] = 1.0; nw[i] = 1.0 / N; }
  for (int t = 0; t < n; t++) {
    for (unsigned i = 0; i < N; i++) r[i] = orc_rng_uniform(rng, 1, t, i, 0); /* functor-model filters draw WIDE blocks */
    filter_resample(nw, r, N, indices + (size_t)N * t, t);
    for (unsigned i = 0; i < N; i++) for (int j = 0; j < m; j++) atmp[j + (size_t)m * i] = A_(j, t, indices[i + (size_t)N * t]);
    for (unsigned i = 0; i < N; i++) { /* :1118-1126: T_fn and R_fn at the ancestor */
      rng_normals_wide(rng, 2, t, i, k, um);
      mod->T_fn(t, atmp + (size_t)m * i, mod->theta, mod->known, x);
      mod->R_fn(t, atmp + (size_t)m * i, mod->theta, mod->known, Rm);
      for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < k; c++) s += Rm[j + m * c] * um[c]; A_(j, t + 1, i) = x[j] + s; }
    }
    double *wcol = weights + (size_t)N * (t + 1);
    if ((t < n - 1) && isfinite(mod->y[t + 1])) {
      for (unsigned i = 0; i < N; i++) NLG_LOG_OBS(t + 1, &A_(0, t + 1, i), lw[i]);
      if (!normalise_step(lw, N, wcol, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
      loglik += inc;
    } else for (unsigned i = 0; i < N; i++) { wcol[i] = 1.0; nw[i] = 1.0 / N; }
  }
#undef NLG_LOG_OBS
done:
  free(nw); free(lw); free(r); free(atmp);
  return loglik;
}


/* ================================================================== */
/* ssm_nlg psi-APF                                                     */
/* ================================================================== */
static void fn_Tg(const orc_fn_model *mod, const double *a, double *Tg /* m x m */) { /* T_gn */
  const double *th = mod->theta, *kp = mod->known;
  switch (mod->id) {
  case 1: { /* R/cpp_example_models.R:432-450 */
    double dT = kp[0], K = kp[1];
    double r = exp(a[0]) / (1 + exp(a[0]));
    double tmp = exp(r * dT) / pow(K + a[1] * (exp(r * dT) - 1), 2);
    Tg[0] = 1.0; Tg[2] = 0;
    Tg[1] = dT * K * a[1] * (K - a[1]) * tmp * r / (1 + exp(a[0]));
    Tg[3] = K * K * tmp;
  } break;
  case 2: Tg[0] = th[1]; break;
  case 3: Tg[0] = cos(a[0]); break;
  default: Tg[0] = 1.0; break;
  }
}
static void fn_Zg(const orc_fn_model *mod, const double *a, double *Zg /* 1 x m */) { /* Z_gn */
  switch (mod->id) { case 1: Zg[0] = 0.0; Zg[1] = 1.0; break; case 2: case 3: Zg[0] = exp(a[0]); break; default: Zg[0] = 1.0; break; }
}

/* dmvnorm(x, mean, sigma, lwr = false, logd = true) for a covariance matrix (src/dmvnorm.cpp:4-58) */
static double dmvnorm_cov(const double *x, const double *mean, const double *S, int d) {
  if (d == 1) return -0.5 * (log(2.0 * ORC_PI) + log(S[0]) + pow(x[0] - mean[0], 2) / S[0]);
  int idx[ORC_MAXM], cnt = 0;
  for (int j = 0; j < d; j++) if (S[j + d * j] > ORC_EPS) idx[cnt++] = j;
  double L[ORC_MAXM * ORC_MAXM] = {0};
  chol_lower_sub(S, d, idx, cnt, L);
  double out = -0.5 * cnt * log(2.0 * ORC_PI), z2[ORC_MAXM];
  for (int a = 0; a < cnt; a++) { /* z2 = L^-1 z by forward substitution; log(rooti.diag) = -log(L.diag) */
    double s = x[idx[a]] - mean[idx[a]];
    for (int b = 0; b < a; b++) s -= L[idx[a] + d * idx[b]] * z2[b];
    z2[a] = s / L[idx[a] + d * idx[a]];
    out += -log(L[idx[a] + d * idx[a]]) - 0.5 * z2[a] * z2[a];
  }
  return out;
}
double orc_dmvnorm_cov(const double *x, const double *mean, const double *S, int d) { return dmvnorm_cov(x, mean, S, d); }
static double dnorm_sd(double y, double mean, double sd) { /* dmvnorm(..., lwr = true) with one element, dmvnorm.cpp:10-11 */
  return -0.5 * (log(2.0 * ORC_PI) + 2.0 * log(sd) + pow((y - mean) / sd, 2));
}

orc_fn_setup *orc_fn_setup_alloc(int n, int m) {
  orc_fn_setup *s = (orc_fn_setup *)calloc(1, sizeof(orc_fn_setup));
  s->n = n; s->m = m;
  s->Z = (double *)calloc((size_t)m * n, sizeof(double)); s->T = (double *)calloc((size_t)m * m * n, sizeof(double));
  s->D = (double *)calloc(n, sizeof(double)); s->C = (double *)calloc((size_t)m * n, sizeof(double));
  s->mode = (double *)calloc((size_t)m * n, sizeof(double)); s->scales = (double *)calloc(n, sizeof(double));
  s->alphahat = (double *)calloc((size_t)m * (n + 1), sizeof(double));
  s->Vt = (double *)calloc((size_t)m * m * (n + 1), sizeof(double)); s->Ct = (double *)calloc((size_t)m * m * (n + 1), sizeof(double));
  return s;
}
void orc_fn_setup_free(orc_fn_setup *s) {
  if (!s) return;
  free(s->Z); free(s->T); free(s->D); free(s->C); free(s->mode); free(s->scales); free(s->alphahat); free(s->Vt); free(s->Ct); free(s);
}

/* approx_mcmc::is_correction_psi<ssm_nlg> reduced to its log-weights (src/approx_mcmc.cpp:353-427, output_type 3): one
 * psi-APF per retained draw, OpenMP parallel for schedule(static) over the draws with n_threads threads exactly as the
 * reference (:364-369).  Used as the multi-core CPU baseline of the post-correction phase in bench.py. */
int orc_fn_is_logweights(const orc_fn_model *mods, int count, const double *modes, unsigned nsim, uint32_t seed,
                         uint32_t first_stream, int n_threads, double *loglik) {
  int bad = 0;
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1) reduction(+ : bad)
  for (int i = 0; i < count; i++) {
    const int n = mods[i].n, m = mods[i].id == 1 ? 2 : 1;
    orc_fn_setup *su = orc_fn_setup_alloc(n, m);
    double *alpha = (double *)malloc(sizeof(double) * (size_t)m * (n + 1) * nsim);
    double *weights = (double *)malloc(sizeof(double) * (size_t)nsim * (n + 1));
    unsigned *indices = (unsigned *)malloc(sizeof(unsigned) * (size_t)nsim * n);
    orc_rng rng;
    memset(&rng, 0, sizeof(rng));
    rng.seed = seed; rng.stream = first_stream + (uint32_t)i;
    if (orc_fn_psi_setup(&mods[i], modes + (size_t)m * n * i, 0, 1e-8, su) != 0) { loglik[i] = -INFINITY; bad++; }
    else loglik[i] = orc_fn_psi_filter(&mods[i], su, nsim, &rng, alpha, weights, indices);
    free(alpha); free(weights); free(indices);
    orc_fn_setup_free(su);
  }
  return bad;
}

/* ================================================================== */
/* ssm_nlg EKF-based particle filter (van der Merwe et al.):           */
/* src/model_ssm_nlg.cpp:1148-1262 (ekf_filter), :1264-1293 (ekf_update_step) */
/* ================================================================== */
static void fn_ekf_update_step(const orc_fn_model *mod, double y, const double *at, const double *Pt, int m, double *att, double *Ptt) {
  if (isfinite(y)) {
    double Zg[2], PZ[2] = {0, 0}, K[2], H = fn_H(mod), F = H * H;
    fn_Zg(mod, at, Zg);
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) PZ[i] += Pt[i + m * j] * Zg[j];
    for (int j = 0; j < m; j++) F += Zg[j] * PZ[j];
    double cF = sqrt(F), v = y - fn_Z(mod, at);
    for (int i = 0; i < m; i++) K[i] = PZ[i] * (1.0 / cF) * (1.0 / cF);
    for (int i = 0; i < m; i++) att[i] = at[i] + K[i] * v;
    double IKZ[4], tmp[4], tmp2[4];
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) IKZ[i + m * j] = (i == j ? 1.0 : 0.0) - K[i] * Zg[j];
    mat_mul(IKZ, Pt, m, m, m, tmp); mat_mul_bt(tmp, IKZ, m, m, m, tmp2);
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) Ptt[i + m * j] = tmp2[i + m * j] + K[i] * H * H * K[j];
  } else {
    memcpy(att, at, sizeof(double) * m); memcpy(Ptt, Pt, sizeof(double) * m * m);
  }
}
/* dmvnorm(x, mean, sigma, lwr = true, logd = true): sigma is a factor (src/dmvnorm.cpp:9-11, 23-25) */
static double dmvnorm_lwr(const double *x, const double *mean, const double *L, int d) {
  if (d == 1) return -0.5 * (log(2.0 * ORC_PI) + 2.0 * log(L[0]) + pow((x[0] - mean[0]) / L[0], 2));
  double S[ORC_MAXM * ORC_MAXM];
  mat_mul_bt(L, L, d, d, d, S);
  return dmvnorm_cov(x, mean, S, d);
}

double orc_fn_ekpf_filter(const orc_fn_model *mod, unsigned N, const orc_rng *rng, double *alpha, double *weights, unsigned *indices) {
  const int n = mod->n, m = mod->id == 1 ? 2 : 1, mm = m * m;
  if (mod->id > 4) return NAN; /* ssm_nlg only */
  double a1[2], P1d[4], P1[4] = {0, 0, 0, 0}, rd[2], RR[4] = {0, 0, 0, 0}, um[2];
  fn_a1P1(mod, a1, P1d); fn_Rdiag(mod, rd);
  if (m == 1) { P1[0] = P1d[0]; RR[0] = rd[0] * rd[0]; } else { P1[0] = P1d[0]; P1[3] = P1d[3]; RR[0] = rd[0] * rd[0]; RR[3] = rd[1] * rd[1]; }
  double *nw = (double *)malloc(sizeof(double) * N), *lw = (double *)malloc(sizeof(double) * N), *r = (double *)malloc(sizeof(double) * N);
  double *atmp = (double *)malloc(sizeof(double) * N * m), *att = (double *)malloc(sizeof(double) * N * m);
  double *Lp = (double *)malloc(sizeof(double) * N * mm);
  double att1[2], Ptt1[4], L[4];
  fn_ekf_update_step(mod, mod->y[0], a1, P1, m, att1, Ptt1); /* :1157-1158 */
  orc_psd_chol(Ptt1, m, L);                                   /* :1161 */
  for (unsigned i = 0; i < N; i++) {                          /* :1163-1172 */
    rng_normals_wide(rng, 0, 0, i, m, um);
    for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < m; c++) s += L[j + m * c] * um[c]; A_(j, 0, i) = att1[j] + s; }
  }
  double loglik = 0.0, inc;
  if (isfinite(mod->y[0])) {                                  /* :1178-1196 */
    for (unsigned i = 0; i < N; i++)
      lw[i] = fn_log_obs(mod, 0, &A_(0, 0, i)) + dmvnorm_cov(&A_(0, 0, i), a1, P1, m) - dmvnorm_lwr(&A_(0, 0, i), att1, L, m);
    if (!normalise_step(lw, N, weights, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
    loglik = inc;
  } else for (unsigned i = 0; i < N; i++) { weights[i] = 1.0; nw[i] = 1.0 / N; }
  for (int t = 0; t < n; t++) {
    for (unsigned i = 0; i < N; i++) r[i] = orc_rng_uniform(rng, 1, t, i, 0); /* functor-model filters draw WIDE blocks */
    filter_resample(nw, r, N, indices + (size_t)N * t, t);
    for (unsigned i = 0; i < N; i++) {                        /* :1212-1227 */
      double at[2], Ptt[4];
      for (int j = 0; j < m; j++) atmp[j + (size_t)m * i] = A_(j, t, indices[i + (size_t)N * t]);
      fn_T(mod, atmp + (size_t)m * i, at);
      if (t < n - 1) {
        fn_ekf_update_step(mod, mod->y[t + 1], at, RR, m, att + (size_t)m * i, Ptt);
        orc_psd_chol(Ptt, m, Lp + (size_t)mm * i);
      } else { /* the last, never-weighted slot uses Pt itself as the factor (:1223-1226) */
        memcpy(att + (size_t)m * i, at, sizeof(double) * m); memcpy(Lp + (size_t)mm * i, RR, sizeof(double) * mm);
      }
    }
    for (unsigned i = 0; i < N; i++) {                        /* :1229-1235 */
      rng_normals_wide(rng, 2, t, i, m, um);
      const double *Li = Lp + (size_t)mm * i;
      for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < m; c++) s += Li[j + m * c] * um[c]; A_(j, t + 1, i) = att[j + (size_t)m * i] + s; }
    }
    double *wcol = weights + (size_t)N * (t + 1);
    if ((t < n - 1) && isfinite(mod->y[t + 1])) {             /* :1236-1256 */
      for (unsigned i = 0; i < N; i++) {
        double mean[2];
        fn_T(mod, atmp + (size_t)m * i, mean);
        lw[i] = fn_log_obs(mod, t + 1, &A_(0, t + 1, i)) + dmvnorm_cov(&A_(0, t + 1, i), mean, RR, m) -
                dmvnorm_lwr(&A_(0, t + 1, i), att + (size_t)m * i, Lp + (size_t)mm * i, m);
      }
      if (!normalise_step(lw, N, wcol, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
      loglik += inc;
    } else for (unsigned i = 0; i < N; i++) { wcol[i] = 1.0; nw[i] = 1.0 / N; }
  }
done:
  free(nw); free(lw); free(r); free(atmp); free(att); free(Lp);
  return loglik;
}

/* extended Kalman filter with iekf_iter = 0: src/model_ssm_nlg.cpp:284-381 (p = 1) */
static int fn_ekf_ll(const orc_fn_model *mod, double *at, double *att, double *Pt, double *Ptt, double *loglik, int iekf_iter) {
  double ll = 0.0;
  const int n = mod->n, m = mod->id == 1 ? 2 : 1, mm = m * m;
  double a1[2], P1[4], H = fn_H(mod), rd[2];
  fn_a1P1(mod, a1, P1); fn_Rdiag(mod, rd);
  memcpy(at, a1, sizeof(double) * m);
  for (int q = 0; q < mm; q++) Pt[q] = (m == 1) ? P1[0] : P1[q];
  if (m == 2) { Pt[0] = P1[0]; Pt[1] = 0; Pt[2] = 0; Pt[3] = P1[3]; }
  for (int t = 0; t < n; t++) {
    const double *a = at + (size_t)m * t, *P = Pt + (size_t)mm * t;
    double *af = att + (size_t)m * t, *Pf = Ptt + (size_t)mm * t;
    if (isfinite(mod->y[t])) {
      double Zg[2], PZ[2] = {0, 0}, K[2], F = H * H;
      fn_Zg(mod, a, Zg);
      for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) PZ[i] += P[i + m * j] * Zg[j];
      for (int j = 0; j < m; j++) F += Zg[j] * PZ[j];
      if (!isfinite(F) || !(F > 0)) return 0;
      double cF = sqrt(F), v = mod->y[t] - fn_Z(mod, a);
      for (int i = 0; i < m; i++) K[i] = PZ[i] * (1.0 / cF) * (1.0 / cF);
      for (int i = 0; i < m; i++) af[i] = a[i] + K[i] * v;
      { /* iterated EKF (:326-355): re-linearise the observation equation at the filtered estimate */
        double diff = 1.0;
        int it = 0;
        while (diff > 1e-4 && it < iekf_iter) {
          it++;
          fn_Zg(mod, af, Zg);
          for (int i = 0; i < m; i++) { PZ[i] = 0; for (int j = 0; j < m; j++) PZ[i] += P[i + m * j] * Zg[j]; }
          F = H * H;
          for (int j = 0; j < m; j++) F += Zg[j] * PZ[j];
          if (!isfinite(F) || !(F > 0)) return 0;
          cF = sqrt(F);
          double zd = 0;
          for (int j = 0; j < m; j++) zd += Zg[j] * (a[j] - af[j]);
          v = mod->y[t] - fn_Z(mod, af) - zd;
          double an[2];
          diff = 0;
          for (int i = 0; i < m; i++) { K[i] = PZ[i] * (1.0 / cF) * (1.0 / cF); an[i] = a[i] + K[i] * v; diff += (af[i] - an[i]) * (af[i] - an[i]); }
          diff /= m;
          for (int i = 0; i < m; i++) af[i] = an[i];
        }
      }
      ll -= 0.5 * (log(2.0 * ORC_PI) + 2.0 * log(cF) + (v / cF) * (v / cF)); /* :363-365 */
      double IKZ[4], tmp[4], tmp2[4];
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) IKZ[i + m * j] = (i == j ? 1.0 : 0.0) - K[i] * Zg[j];
      mat_mul(IKZ, P, m, m, m, tmp); mat_mul_bt(tmp, IKZ, m, m, m, tmp2);
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) Pf[i + m * j] = tmp2[i + m * j] + K[i] * H * H * K[j];
    } else { memcpy(af, a, sizeof(double) * m); memcpy(Pf, P, sizeof(double) * mm); }
    double Tg[4], tmp[4], tmp2[4];
    fn_T(mod, af, at + (size_t)m * (t + 1));
    fn_Tg(mod, af, Tg);
    mat_mul(Tg, Pf, m, m, m, tmp); mat_mul_bt(tmp, Tg, m, m, m, tmp2);
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) Pt[i + m * j + (size_t)mm * (t + 1)] = tmp2[i + m * j] + (i == j ? rd[i] * rd[i] : 0.0);
  }
  if (loglik) *loglik = ll;
  return 1;
}
static int fn_ekf(const orc_fn_model *mod, double *at, double *att, double *Pt, double *Ptt) { return fn_ekf_ll(mod, at, att, Pt, Ptt, NULL, 0); }
/* ekf(model, iekf_iter)$logLik (src/model_ssm_nlg.cpp:284-381); pinned by tests/testthat/test_ekpf.R:64,67 */
double orc_fn_ekf_loglik(const orc_fn_model *mod, int iekf_iter) {
  const int n = mod->n, m = mod->id == 1 ? 2 : 1;
  double *at = (double *)calloc((size_t)m * (n + 1), sizeof(double)), *att = (double *)calloc((size_t)m * n, sizeof(double));
  double *Pt = (double *)calloc((size_t)m * m * (n + 1), sizeof(double)), *Ptt = (double *)calloc((size_t)m * m * n, sizeof(double));
  double ll = 0.0;
  if (!fn_ekf_ll(mod, at, att, Pt, Ptt, &ll, iekf_iter)) ll = -INFINITY;
  free(at); free(att); free(Pt); free(Ptt);
  return ll;
}

static void fn_linearise(const orc_fn_model *mod, const double *zpts /* m x n: where Z is linearised */,
                         const double *tpts /* m x n: where T is linearised */, orc_fn_setup *su) {
  const int n = mod->n, m = su->m;
  for (int t = 0; t < n; t++) { /* :81-101 / :156-170 / :196-215 */
    const double *zp = zpts + (size_t)m * t, *tp = tpts + (size_t)m * t;
    double *Zt = su->Z + (size_t)m * t, *Tt = su->T + (size_t)m * m * t, Tf[2];
    fn_Zg(mod, zp, Zt); fn_Tg(mod, tp, Tt);
    su->D[t] = fn_Z(mod, zp) - dotv(Zt, zp, m);
    fn_T(mod, tp, Tf);
    for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += Tt[i + m * j] * tp[j]; su->C[i + (size_t)m * t] = Tf[i] - s; }
  }
  su->H = fn_H(mod);
  double rd[2]; fn_Rdiag(mod, rd);
  memset(su->R, 0, sizeof(su->R));
  for (int i = 0; i < m; i++) su->R[i + m * i] = rd[i];
}

static void fn_approx_ulg(const orc_fn_model *mod, const orc_fn_setup *su, double *a1, double *P1, double *xb, orc_ulg_model *g) {
  const int m = su->m;
  double p1[4];
  fn_a1P1(mod, a1, p1);
  if (m == 1) P1[0] = p1[0]; else { P1[0] = p1[0]; P1[1] = 0; P1[2] = 0; P1[3] = p1[3]; }
  memset(xb, 0, sizeof(double) * mod->n);
  g->n = mod->n; g->m = m; g->k = m;
  g->Ztv = 1; g->Ttv = 1; g->Rtv = 0; g->Dtv = 1; g->Ctv = 1; g->Htv = 0;
  g->y = mod->y; g->Z = su->Z; g->T = su->T; g->R = su->R; g->a1 = a1; g->P1 = P1; g->D = su->D; g->C = su->C; g->xbeta = xb; g->H = &su->H;
}

/* src/model_ssm_nlg.cpp:1295-1319 */
static double fn_log_signal_pdf(const orc_fn_model *mod, const double *al, int m) {
  const int n = mod->n;
  double a1[2], p1[4], P1[4] = {0}, rd[2], RR[4] = {0};
  fn_a1P1(mod, a1, p1); fn_Rdiag(mod, rd);
  if (m == 1) { P1[0] = p1[0]; RR[0] = rd[0] * rd[0]; } else { P1[0] = p1[0]; P1[3] = p1[3]; RR[0] = rd[0] * rd[0]; RR[3] = rd[1] * rd[1]; }
  double ll = dmvnorm_cov(al, a1, P1, m);
  if (isfinite(mod->y[0])) ll += dnorm_sd(mod->y[0], fn_Z(mod, al), fn_H(mod));
  for (int t = 0; t < n - 1; t++) {
    double mean[2];
    fn_T(mod, al + (size_t)m * t, mean);
    ll += dmvnorm_cov(al + (size_t)m * (t + 1), mean, RR, m);
    if (isfinite(mod->y[t + 1])) ll += dnorm_sd(mod->y[t + 1], fn_Z(mod, al + (size_t)m * (t + 1)), fn_H(mod));
  }
  return ll;
}

int orc_fn_psi_setup(const orc_fn_model *mod, const double *is_mode, int max_iter, double conv_tol, orc_fn_setup *su) {
  const int n = mod->n, m = su->m, mm = m * m;
  double a1[2], P1[4];
  double *xb = (double *)calloc(n, sizeof(double));
  orc_ulg_model g;
  su->ok = 1; su->iters = 0;
  if (is_mode) { /* approximate_for_is: :146-176; scales stay 0 and approx_loglik 0 */
    memcpy(su->mode, is_mode, sizeof(double) * m * n);
    fn_linearise(mod, su->mode, su->mode, su);
    memset(su->scales, 0, sizeof(double) * n);
    su->approx_loglik = 0.0;
    fn_approx_ulg(mod, su, a1, P1, xb, &g);
  } else { /* approximate(): :62-143 */
    double *at = (double *)calloc((size_t)m * (n + 1), sizeof(double)), *att = (double *)calloc((size_t)m * n, sizeof(double));
    double *Pt = (double *)calloc((size_t)mm * (n + 1), sizeof(double)), *Ptt = (double *)calloc((size_t)mm * n, sizeof(double));
    double *sm = (double *)calloc((size_t)m * (n + 1), sizeof(double)), *newmode = (double *)calloc((size_t)m * n, sizeof(double));
    su->ok = fn_ekf(mod, at, att, Pt, Ptt);
    if (su->ok) {
      fn_linearise(mod, at, att, su); /* approximate_by_ekf :179-215: Z at a_t, T at a_t|t */
      fn_approx_ulg(mod, su, a1, P1, xb, &g);
      orc_ulg_fast_smoother(&g, sm);
      memcpy(su->mode, sm, sizeof(double) * m * n);
      for (size_t q = 0; q < (size_t)m * n; q++) if (!isfinite(su->mode[q])) su->ok = 0;
    }
    if (su->ok && max_iter > 0) {
      double ll = fn_log_signal_pdf(mod, su->mode, m), rel = 1.0e300, ad = 1;
      int i = 0;
      while (i < max_iter && rel > conv_tol && ad > 1e-4) {
        i++;
        fn_linearise(mod, su->mode, su->mode, su);
        orc_ulg_fast_smoother(&g, sm);
        memcpy(newmode, sm, sizeof(double) * m * n);
        double lln = fn_log_signal_pdf(mod, newmode, m);
        ad = lln - ll; rel = ad / fabs(ll);
        if (!isfinite(lln)) { su->ok = 0; break; }
        /* the reference's back-tracking branch (:118-138) needs rel < -conv_tol AND ad > 1e-4 at once: unreachable */
        memcpy(su->mode, newmode, sizeof(double) * m * n);
        ll = lln;
      }
      su->iters = i;
    }
    if (su->ok) { /* psi_filter :974-982 */
      double gll = orc_ulg_loglik(&g);
      /* update_scales :854-879 */
      double rd[2], RR[4] = {0}, HH = su->H * su->H;
      fn_Rdiag(mod, rd);
      if (m == 1) RR[0] = rd[0] * rd[0]; else { RR[0] = rd[0] * rd[0]; RR[3] = rd[1] * rd[1]; }
      for (int t = 0; t < n; t++) {
        su->scales[t] = 0.0;
        const double *md = su->mode + (size_t)m * t;
        if (isfinite(mod->y[t])) {
          double am = su->D[t] + dotv(su->Z + (size_t)m * t, md, m);
          su->scales[t] = dnorm_sd(mod->y[t], fn_Z(mod, md), fn_H(mod)) - dmvnorm_cov(&mod->y[t], &am, &HH, 1);
        }
      }
      for (int t = 1; t < n; t++) {
        const double *mp = su->mode + (size_t)m * (t - 1), *md = su->mode + (size_t)m * t;
        double mean[2], amean[2];
        fn_T(mod, mp, mean);
        for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += su->T[i + m * j + (size_t)mm * (t - 1)] * mp[j]; amean[i] = su->C[i + (size_t)m * (t - 1)] + s; }
        su->scales[t] += dmvnorm_cov(md, mean, RR, m) - dmvnorm_cov(md, amean, RR, m);
      }
      double ssc = 0; for (int t = 0; t < n; t++) ssc += su->scales[t];
      su->approx_loglik = gll + ssc;
    }
    free(at); free(att); free(Pt); free(Ptt); free(sm); free(newmode);
  }
  if (su->ok) {
    orc_ulg_smoother_ccov(&g, su->alphahat, su->Vt, su->Ct); /* :984-991 */
    for (size_t q = 0; q < (size_t)mm * (n + 1); q++) if (!isfinite(su->Vt[q]) || !isfinite(su->Ct[q])) su->ok = 0;
    if (su->ok) orc_conditional_cov(su->Vt, su->Ct, m, n + 1);
  }
  free(xb);
  return su->ok ? 0 : -1;
}

/* src/model_ssm_nlg.cpp:880-935, one particle; aprev = ancestor (used for t > 0) */
static double fn_psi_log_weight(const orc_fn_model *mod, const orc_fn_setup *su, int t, const double *x, const double *aprev) {
  const int m = su->m, mm = m * m;
  double w = 0.0, HH = su->H * su->H;
  if (isfinite(mod->y[t])) {
    double am = su->D[t] + dotv(su->Z + (size_t)m * t, x, m);
    w = dnorm_sd(mod->y[t], fn_Z(mod, x), fn_H(mod)) - dmvnorm_cov(&mod->y[t], &am, &HH, 1);
  }
  if (t > 0) {
    double mean[2], amean[2], rd[2], RR[4] = {0};
    fn_T(mod, aprev, mean); fn_Rdiag(mod, rd);
    if (m == 1) RR[0] = rd[0] * rd[0]; else { RR[0] = rd[0] * rd[0]; RR[3] = rd[1] * rd[1]; }
    for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += su->T[i + m * j + (size_t)mm * (t - 1)] * aprev[j]; amean[i] = su->C[i + (size_t)m * (t - 1)] + s; }
    w -= dmvnorm_cov(x, amean, RR, m) - dmvnorm_cov(x, mean, RR, m);
  }
  return w;
}

/* src/model_ssm_nlg.cpp:993-1063 */
double orc_fn_psi_filter(const orc_fn_model *mod, const orc_fn_setup *su, unsigned N, const orc_rng *rng,
                         double *alpha, double *weights, unsigned *indices) {
  const int n = mod->n, m = su->m, mm = m * m;
  if (!su->ok) return -INFINITY;
  double *nw = (double *)malloc(sizeof(double) * N), *lw = (double *)malloc(sizeof(double) * N);
  double *r = (double *)malloc(sizeof(double) * N), *atmp = (double *)calloc((size_t)N * m, sizeof(double));
  double um[2];
  for (unsigned i = 0; i < N; i++) {
    rng_normals_wide(rng, 0, 0, i, m, um);
    for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < m; c++) s += su->Vt[j + m * c] * um[c]; A_(j, 0, i) = su->alphahat[j] + s; }
  }
  double loglik = su->approx_loglik, inc;
  if (isfinite(mod->y[0])) {
    for (unsigned i = 0; i < N; i++) lw[i] = fn_psi_log_weight(mod, su, 0, &A_(0, 0, i), atmp) - su->scales[0];
    if (!normalise_step(lw, N, weights, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
    loglik += inc;
  } else for (unsigned i = 0; i < N; i++) { weights[i] = 1.0; nw[i] = 1.0 / N; }
  for (int t = 0; t < n; t++) {
    for (unsigned i = 0; i < N; i++) r[i] = orc_rng_uniform(rng, 1, t, i, 0); /* functor-model filters draw WIDE blocks */
    filter_resample(nw, r, N, indices + (size_t)N * t, t);
    for (unsigned i = 0; i < N; i++) for (int j = 0; j < m; j++) atmp[j + (size_t)m * i] = A_(j, t, indices[i + (size_t)N * t]);
    const double *ah0 = su->alphahat + (size_t)m * t, *ah1 = su->alphahat + (size_t)m * (t + 1);
    const double *Ct1 = su->Ct + (size_t)mm * (t + 1), *Vt1 = su->Vt + (size_t)mm * (t + 1);
    for (unsigned i = 0; i < N; i++) {
      rng_normals_wide(rng, 2, t, i, m, um);
      for (int j = 0; j < m; j++) {
        double s = 0, q = 0;
        for (int c = 0; c < m; c++) s += Ct1[j + m * c] * (atmp[c + (size_t)m * i] - ah0[c]);
        for (int c = 0; c < m; c++) q += Vt1[j + m * c] * um[c];
        A_(j, t + 1, i) = ah1[j] + s + q;
      }
    }
    double *wcol = weights + (size_t)N * (t + 1);
    if ((t < n - 1) && isfinite(mod->y[t + 1])) {
      for (unsigned i = 0; i < N; i++) lw[i] = fn_psi_log_weight(mod, su, t + 1, &A_(0, t + 1, i), atmp + (size_t)m * i) - su->scales[t + 1];
      if (!normalise_step(lw, N, wcol, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
      loglik += inc;
    } else for (unsigned i = 0; i < N; i++) { wcol[i] = 1.0; nw[i] = 1.0 / N; }
  }
done:
  free(nw); free(lw); free(r); free(atmp);
  return loglik;
}


/* ================================================================== */
/* ssm_nlg psi-APF behind function pointers (general m, k; p = 1)      */
/* ================================================================== */
orc_nlg_setup *orc_nlg_setup_alloc(int n, int m, int k) {
  orc_nlg_setup *s = (orc_nlg_setup *)calloc(1, sizeof(orc_nlg_setup));
  s->n = n; s->m = m; s->k = k;
  s->Z = (double *)calloc((size_t)m * n, sizeof(double)); s->T = (double *)calloc((size_t)m * m * n, sizeof(double));
  s->D = (double *)calloc(n, sizeof(double)); s->C = (double *)calloc((size_t)m * n, sizeof(double));
  s->H = (double *)calloc(n, sizeof(double)); s->R = (double *)calloc((size_t)m * k * n, sizeof(double));
  s->mode = (double *)calloc((size_t)m * n, sizeof(double)); s->scales = (double *)calloc(n, sizeof(double));
  s->alphahat = (double *)calloc((size_t)m * (n + 1), sizeof(double));
  s->Vt = (double *)calloc((size_t)m * m * (n + 1), sizeof(double)); s->Ct = (double *)calloc((size_t)m * m * (n + 1), sizeof(double));
  return s;
}
void orc_nlg_setup_free(orc_nlg_setup *s) {
  if (!s) return;
  free(s->Z); free(s->T); free(s->D); free(s->C); free(s->H); free(s->R); free(s->mode); free(s->scales);
  free(s->alphahat); free(s->Vt); free(s->Ct); free(s);
}
#define GEN_CALL(fn_, t_, a_, out_) mod->fn_((t_), (a_), mod->base.theta, mod->base.known, (out_))

/* ekf with iekf_iter = 0 (:284-381), one observation series; returns 0 when the innovation variance fails the checks */
static int gen_ekf(const orc_nlg_general_psi *mod, double *at, double *att, double *Pt, double *Ptt, double *loglik) {
  const orc_nlg_general *b = &mod->base;
  const int n = b->n, m = b->m, k = b->k, mm = m * m;
  double ll = 0.0;
  memcpy(at, b->a1, sizeof(double) * m);
  memcpy(Pt, b->P1, sizeof(double) * mm);
  for (int t = 0; t < n; t++) {
    const double *a = at + (size_t)m * t, *P = Pt + (size_t)mm * t;
    double *af = att + (size_t)m * t, *Pf = Ptt + (size_t)mm * t;
    if (isfinite(b->y[t])) {
      double Zg[ORC_MAXM], PZ[ORC_MAXM], K[ORC_MAXM], h, z;
      GEN_CALL(Z_gn, t, a, Zg); b->H_fn(t, a, b->theta, b->known, &h);
      double F = h * h;
      for (int i = 0; i < m; i++) { PZ[i] = 0; for (int j = 0; j < m; j++) PZ[i] += P[i + m * j] * Zg[j]; }
      for (int j = 0; j < m; j++) F += Zg[j] * PZ[j];
      if (!isfinite(F) || !(F > 0)) return 0;
      const double cF = sqrt(F);
      b->Z_fn(t, a, b->theta, b->known, &z);
      const double v = b->y[t] - z;
      for (int i = 0; i < m; i++) { K[i] = PZ[i] * (1.0 / cF) * (1.0 / cF); af[i] = a[i] + K[i] * v; }
      double IKZ[ORC_MAXM * ORC_MAXM], tmp[ORC_MAXM * ORC_MAXM], tmp2[ORC_MAXM * ORC_MAXM];
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) IKZ[i + m * j] = (i == j ? 1.0 : 0.0) - K[i] * Zg[j];
      mat_mul(IKZ, P, m, m, m, tmp); mat_mul_bt(tmp, IKZ, m, m, m, tmp2);
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) Pf[i + m * j] = tmp2[i + m * j] + K[i] * h * h * K[j];
      ll -= 0.5 * (log(2.0 * ORC_PI) + 2.0 * log(cF) + (v / cF) * (v / cF));
    } else { memcpy(af, a, sizeof(double) * m); memcpy(Pf, P, sizeof(double) * mm); }
    double Tg[ORC_MAXM * ORC_MAXM], Rt[ORC_MAXM * ORC_MAXM], RR[ORC_MAXM * ORC_MAXM], tmp[ORC_MAXM * ORC_MAXM], tmp2[ORC_MAXM * ORC_MAXM];
    b->T_fn(t, af, b->theta, b->known, at + (size_t)m * (t + 1));
    GEN_CALL(T_gn, t, af, Tg); b->R_fn(t, af, b->theta, b->known, Rt);
    compute_RR(Rt, m, k, RR);
    mat_mul(Tg, Pf, m, m, m, tmp); mat_mul_bt(tmp, Tg, m, m, m, tmp2);
    for (int q = 0; q < mm; q++) Pt[q + (size_t)mm * (t + 1)] = tmp2[q] + RR[q];
  }
  if (loglik) *loglik = ll;
  return 1;
}
double orc_nlg_general_ekf_loglik(const orc_nlg_general_psi *mod) {
  const int n = mod->base.n, m = mod->base.m;
  double *at = (double *)calloc((size_t)m * (n + 1), sizeof(double)), *att = (double *)calloc((size_t)m * n, sizeof(double));
  double *Pt = (double *)calloc((size_t)m * m * (n + 1), sizeof(double)), *Ptt = (double *)calloc((size_t)m * m * n, sizeof(double));
  double ll = 0.0;
  if (!gen_ekf(mod, at, att, Pt, Ptt, &ll)) ll = -INFINITY;
  free(at); free(att); free(Pt); free(Ptt);
  return ll;
}
/* the approximating model's operands from the points where the reference linearises: Z, H at zpts, T, R at tpts */
static void gen_linearise(const orc_nlg_general_psi *mod, const double *zpts, const double *tpts, orc_nlg_setup *su) {
  const orc_nlg_general *b = &mod->base;
  const int n = b->n, m = b->m, k = b->k;
  for (int t = 0; t < n; t++) {
    const double *zp = zpts + (size_t)m * t, *tp = tpts + (size_t)m * t;
    double *Zt = su->Z + (size_t)m * t, *Tt = su->T + (size_t)m * m * t, Tf[ORC_MAXM], z;
    GEN_CALL(Z_gn, t, zp, Zt); GEN_CALL(T_gn, t, tp, Tt);
    b->Z_fn(t, zp, b->theta, b->known, &z);
    su->D[t] = z - dotv(Zt, zp, m);
    b->T_fn(t, tp, b->theta, b->known, Tf);
    for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += Tt[i + m * j] * tp[j]; su->C[i + (size_t)m * t] = Tf[i] - s; }
    b->H_fn(t, zp, b->theta, b->known, &su->H[t]);
    b->R_fn(t, tp, b->theta, b->known, su->R + (size_t)m * k * t);
  }
}
static void gen_approx_ulg(const orc_nlg_general_psi *mod, const orc_nlg_setup *su, double *xb, orc_ulg_model *g) {
  const orc_nlg_general *b = &mod->base;
  memset(xb, 0, sizeof(double) * b->n);
  g->n = b->n; g->m = b->m; g->k = b->k;
  g->Ztv = 1; g->Ttv = 1; g->Rtv = 1; g->Dtv = 1; g->Ctv = 1; g->Htv = 1;
  g->y = b->y; g->Z = su->Z; g->T = su->T; g->R = su->R; g->a1 = b->a1; g->P1 = b->P1; g->D = su->D; g->C = su->C; g->xbeta = xb; g->H = su->H;
}
/* :1295-1319 */
static double gen_log_signal_pdf(const orc_nlg_general_psi *mod, const double *al) {
  const orc_nlg_general *b = &mod->base;
  const int n = b->n, m = b->m, k = b->k;
  double z, h, mean[ORC_MAXM], Rt[ORC_MAXM * ORC_MAXM], RR[ORC_MAXM * ORC_MAXM];
  double ll = dmvnorm_cov(al, b->a1, b->P1, m);
  if (isfinite(b->y[0])) { b->Z_fn(0, al, b->theta, b->known, &z); b->H_fn(0, al, b->theta, b->known, &h); ll += dnorm_sd(b->y[0], z, h); }
  for (int t = 0; t < n - 1; t++) {
    const double *a0 = al + (size_t)m * t, *a1 = al + (size_t)m * (t + 1);
    b->T_fn(t, a0, b->theta, b->known, mean); b->R_fn(t, a0, b->theta, b->known, Rt);
    compute_RR(Rt, m, k, RR);
    ll += dmvnorm_cov(a1, mean, RR, m);
    if (isfinite(b->y[t + 1])) { b->Z_fn(t + 1, a1, b->theta, b->known, &z); b->H_fn(t + 1, a1, b->theta, b->known, &h); ll += dnorm_sd(b->y[t + 1], z, h); }
  }
  return ll;
}
/* update_scales :854-879 */
static void gen_update_scales(const orc_nlg_general_psi *mod, orc_nlg_setup *su) {
  const orc_nlg_general *b = &mod->base;
  const int n = b->n, m = b->m, k = b->k, mm = m * m;
  for (int t = 0; t < n; t++) {
    su->scales[t] = 0.0;
    const double *md = su->mode + (size_t)m * t;
    if (isfinite(b->y[t])) {
      double z, h, am = su->D[t] + dotv(su->Z + (size_t)m * t, md, m), HH = su->H[t] * su->H[t];
      b->Z_fn(t, md, b->theta, b->known, &z); b->H_fn(t, md, b->theta, b->known, &h);
      su->scales[t] = dnorm_sd(b->y[t], z, h) - dmvnorm_cov(&b->y[t], &am, &HH, 1);
    }
  }
  for (int t = 1; t < n; t++) {
    const double *mp = su->mode + (size_t)m * (t - 1), *md = su->mode + (size_t)m * t;
    double mean[ORC_MAXM], amean[ORC_MAXM], Rt[ORC_MAXM * ORC_MAXM], RR[ORC_MAXM * ORC_MAXM], RRa[ORC_MAXM * ORC_MAXM];
    b->T_fn(t - 1, mp, b->theta, b->known, mean); b->R_fn(t - 1, mp, b->theta, b->known, Rt);
    compute_RR(Rt, m, k, RR);
    compute_RR(su->R + (size_t)m * k * (t - 1), m, k, RRa);
    for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += su->T[i + m * j + (size_t)mm * (t - 1)] * mp[j]; amean[i] = su->C[i + (size_t)m * (t - 1)] + s; }
    su->scales[t] += dmvnorm_cov(md, mean, RR, m) - dmvnorm_cov(md, amean, RRa, m);
  }
}
int orc_nlg_general_psi_setup(const orc_nlg_general_psi *mod, const double *is_mode, int max_iter, double conv_tol, orc_nlg_setup *su) {
  const orc_nlg_general *b = &mod->base;
  const int n = b->n, m = b->m, mm = m * m;
  double *xb = (double *)calloc(n, sizeof(double));
  orc_ulg_model g;
  su->ok = 1; su->iters = 0;
  if (is_mode) { /* approximate_for_is :146-176 */
    memcpy(su->mode, is_mode, sizeof(double) * m * n);
    gen_linearise(mod, su->mode, su->mode, su);
    gen_approx_ulg(mod, su, xb, &g);
  } else {
    double *at = (double *)calloc((size_t)m * (n + 1), sizeof(double)), *att = (double *)calloc((size_t)m * n, sizeof(double));
    double *Pt = (double *)calloc((size_t)mm * (n + 1), sizeof(double)), *Ptt = (double *)calloc((size_t)mm * n, sizeof(double));
    double *sm = (double *)calloc((size_t)m * (n + 1), sizeof(double)), *newmode = (double *)calloc((size_t)m * n, sizeof(double));
    double *oldmode = (double *)calloc((size_t)m * n, sizeof(double));
    su->ok = gen_ekf(mod, at, att, Pt, Ptt, NULL);
    if (su->ok) {
      gen_linearise(mod, at, att, su); /* approximate_by_ekf :179-222: Z, H at a_t; T, R at a_t|t */
      gen_approx_ulg(mod, su, xb, &g);
      orc_ulg_fast_smoother(&g, sm);
      memcpy(su->mode, sm, sizeof(double) * m * n);
      for (size_t q = 0; q < (size_t)m * n; q++) if (!isfinite(su->mode[q])) su->ok = 0;
    }
    if (su->ok && max_iter > 0) {
      double ll = gen_log_signal_pdf(mod, su->mode), rel = 1.0e300, ad = 1;
      int i = 0;
      while (i < max_iter && rel > conv_tol && ad > 1e-4) {
        i++;
        gen_linearise(mod, su->mode, su->mode, su);
        orc_ulg_fast_smoother(&g, sm);
        memcpy(newmode, sm, sizeof(double) * m * n);
        double lln = gen_log_signal_pdf(mod, newmode);
        ad = lln - ll; rel = ad / fabs(ll);
        int fin = isfinite(lln);
        for (size_t q = 0; q < (size_t)m * n && fin; q++) if (!isfinite(newmode[q])) fin = 0;
        if (!fin) { su->ok = 0; break; }
        if (rel < -conv_tol && i > 1 && ad > 1e-4) { /* :118-138 (unreachable: ad > 1e-4 makes rel positive; kept as written) */
          int ii = 0; double step = 1.0;
          memcpy(oldmode, su->mode, sizeof(double) * m * n);
          while (rel < -conv_tol && ii < 15 && ad > 1e-4) {
            step /= 2.0;
            for (size_t q = 0; q < (size_t)m * n; q++) su->mode[q] = (1.0 - step) * oldmode[q] + step * newmode[q];
            lln = gen_log_signal_pdf(mod, su->mode);
            ad = lln - ll; rel = ad / fabs(ll); ii++;
            if (!isfinite(lln)) { su->ok = 0; break; }
          }
          if (ii == 15) su->ok = 0;
          if (!su->ok) break;
          memcpy(newmode, su->mode, sizeof(double) * m * n);
        }
        memcpy(su->mode, newmode, sizeof(double) * m * n);
        ll = lln;
      }
      su->iters = i;
    }
    free(at); free(att); free(Pt); free(Ptt); free(sm); free(newmode); free(oldmode);
  }
  if (su->ok) {
    if (is_mode) { /* approximate_for_is leaves approx_state = 2: psi_filter skips its own scales / approx_loglik branch, and the
                      IS driver has set approx_loglik = 0 (src/approx_mcmc.cpp:372-373); at the linearisation point both density
                      ratios are exactly one, so the scales are zero */
      memset(su->scales, 0, sizeof(double) * n);
      su->approx_loglik = 0.0;
    } else { /* :974-982 */
      const double gll = orc_ulg_loglik(&g);
      gen_update_scales(mod, su);
      double ssc = 0; for (int t = 0; t < n; t++) ssc += su->scales[t];
      su->approx_loglik = gll + ssc;
    }
    orc_ulg_smoother_ccov(&g, su->alphahat, su->Vt, su->Ct); /* :984-991 */
    for (size_t q = 0; q < (size_t)mm * (n + 1); q++) if (!isfinite(su->Vt[q]) || !isfinite(su->Ct[q])) su->ok = 0;
    if (su->ok) orc_conditional_cov(su->Vt, su->Ct, m, n + 1);
  }
  free(xb);
  return su->ok ? 0 : -1;
}
/* log_weights :880-935 for one particle (x at time t, aprev its ancestor, used for t > 0) */
static double gen_psi_log_weight(const orc_nlg_general_psi *mod, const orc_nlg_setup *su, int t, const double *x, const double *aprev) {
  const orc_nlg_general *b = &mod->base;
  const int m = b->m, k = b->k, mm = m * m;
  double w = 0.0;
  if (isfinite(b->y[t])) {
    double z, h, am = su->D[t] + dotv(su->Z + (size_t)m * t, x, m), HH = su->H[t] * su->H[t];
    b->Z_fn(t, x, b->theta, b->known, &z); b->H_fn(t, x, b->theta, b->known, &h);
    w = dnorm_sd(b->y[t], z, h) - dmvnorm_cov(&b->y[t], &am, &HH, 1);
  }
  if (t > 0) {
    double mean[ORC_MAXM], amean[ORC_MAXM], Rt[ORC_MAXM * ORC_MAXM], RR[ORC_MAXM * ORC_MAXM], RRa[ORC_MAXM * ORC_MAXM];
    b->T_fn(t - 1, aprev, b->theta, b->known, mean); b->R_fn(t - 1, aprev, b->theta, b->known, Rt);
    compute_RR(Rt, m, k, RR);
    compute_RR(su->R + (size_t)m * k * (t - 1), m, k, RRa);
    for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < m; j++) s += su->T[i + m * j + (size_t)mm * (t - 1)] * aprev[j]; amean[i] = su->C[i + (size_t)m * (t - 1)] + s; }
    w -= dmvnorm_cov(x, amean, RRa, m) - dmvnorm_cov(x, mean, RR, m);
  }
  return w;
}
double orc_nlg_general_psi_filter(const orc_nlg_general_psi *mod, const orc_nlg_setup *su, unsigned N, const orc_rng *rng,
                                  double *alpha, double *weights, unsigned *indices) {
  const orc_nlg_general *b = &mod->base;
  const int n = b->n, m = b->m, mm = m * m;
  if (!su->ok) return -INFINITY;
  double *nw = (double *)malloc(sizeof(double) * N), *lw = (double *)malloc(sizeof(double) * N);
  double *r = (double *)malloc(sizeof(double) * N), *atmp = (double *)calloc((size_t)N * m, sizeof(double));
  double um[ORC_MAXM];
  for (unsigned i = 0; i < N; i++) {
    rng_normals_wide(rng, 0, 0, i, m, um);
    for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < m; c++) s += su->Vt[j + m * c] * um[c]; A_(j, 0, i) = su->alphahat[j] + s; }
  }
  double loglik = su->approx_loglik, inc;
  if (isfinite(b->y[0])) {
    for (unsigned i = 0; i < N; i++) lw[i] = gen_psi_log_weight(mod, su, 0, &A_(0, 0, i), atmp) - su->scales[0];
    if (!normalise_step(lw, N, weights, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
    loglik += inc;
  } else for (unsigned i = 0; i < N; i++) { weights[i] = 1.0; nw[i] = 1.0 / N; }
  for (int t = 0; t < n; t++) {
    for (unsigned i = 0; i < N; i++) r[i] = orc_rng_uniform(rng, 1, t, i, 0); /* functor-model filters draw WIDE blocks */
    filter_resample(nw, r, N, indices + (size_t)N * t, t);
    for (unsigned i = 0; i < N; i++) for (int j = 0; j < m; j++) atmp[j + (size_t)m * i] = A_(j, t, indices[i + (size_t)N * t]);
    const double *ah0 = su->alphahat + (size_t)m * t, *ah1 = su->alphahat + (size_t)m * (t + 1);
    const double *Ct1 = su->Ct + (size_t)mm * (t + 1), *Vt1 = su->Vt + (size_t)mm * (t + 1);
    for (unsigned i = 0; i < N; i++) {
      rng_normals_wide(rng, 2, t, i, m, um);
      for (int j = 0; j < m; j++) {
        double s = 0, q = 0;
        for (int c = 0; c < m; c++) s += Ct1[j + m * c] * (atmp[c + (size_t)m * i] - ah0[c]);
        for (int c = 0; c < m; c++) q += Vt1[j + m * c] * um[c];
        A_(j, t + 1, i) = ah1[j] + s + q;
      }
    }
    double *wcol = weights + (size_t)N * (t + 1);
    if ((t < n - 1) && isfinite(b->y[t + 1])) {
      for (unsigned i = 0; i < N; i++) lw[i] = gen_psi_log_weight(mod, su, t + 1, &A_(0, t + 1, i), atmp + (size_t)m * i) - su->scales[t + 1];
      if (!normalise_step(lw, N, wcol, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
      loglik += inc;
    } else for (unsigned i = 0; i < N; i++) { wcol[i] = 1.0; nw[i] = 1.0 / N; }
  }
done:
  free(nw); free(lw); free(r); free(atmp);
  return loglik;
}
/* ekf_update_step :1264-1293, one observation series; t = the time of the observation */
static void gen_ekf_update_step(const orc_nlg_general_psi *mod, int t, const double *at, const double *Pt, double *att, double *Ptt) {
  const orc_nlg_general *b = &mod->base;
  const int m = b->m;
  if (isfinite(b->y[t])) {
    double Zg[ORC_MAXM], PZ[ORC_MAXM], K[ORC_MAXM], h, z;
    GEN_CALL(Z_gn, t, at, Zg); b->H_fn(t, at, b->theta, b->known, &h);
    double F = h * h;
    for (int i = 0; i < m; i++) { PZ[i] = 0; for (int j = 0; j < m; j++) PZ[i] += Pt[i + m * j] * Zg[j]; }
    for (int j = 0; j < m; j++) F += Zg[j] * PZ[j];
    const double cF = sqrt(F);
    b->Z_fn(t, at, b->theta, b->known, &z);
    const double v = b->y[t] - z;
    for (int i = 0; i < m; i++) { K[i] = PZ[i] * (1.0 / cF) * (1.0 / cF); att[i] = at[i] + K[i] * v; }
    double IKZ[ORC_MAXM * ORC_MAXM], tmp[ORC_MAXM * ORC_MAXM], tmp2[ORC_MAXM * ORC_MAXM];
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) IKZ[i + m * j] = (i == j ? 1.0 : 0.0) - K[i] * Zg[j];
    mat_mul(IKZ, Pt, m, m, m, tmp); mat_mul_bt(tmp, IKZ, m, m, m, tmp2);
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) Ptt[i + m * j] = tmp2[i + m * j] + K[i] * h * h * K[j];
  } else {
    memcpy(att, at, sizeof(double) * m); memcpy(Ptt, Pt, sizeof(double) * m * m);
  }
}
/* ssm_nlg::ekf_filter (EKF-proposal particle filter) :1148-1262 over function pointers; R_fn and T_fn at the ancestor */
double orc_nlg_general_ekpf_filter(const orc_nlg_general_psi *mod, unsigned N, const orc_rng *rng, double *alpha, double *weights,
                                   unsigned *indices) {
  const orc_nlg_general *b = &mod->base;
  const int n = b->n, m = b->m, k = b->k, mm = m * m;
  double um[ORC_MAXM], att1[ORC_MAXM], Ptt1[ORC_MAXM * ORC_MAXM], L[ORC_MAXM * ORC_MAXM];
  double *nw = (double *)malloc(sizeof(double) * N), *lw = (double *)malloc(sizeof(double) * N), *r = (double *)malloc(sizeof(double) * N);
  double *atmp = (double *)malloc(sizeof(double) * N * m), *att = (double *)malloc(sizeof(double) * N * m);
  double *Lp = (double *)malloc(sizeof(double) * N * mm);
#define GEN_LOG_OBS(t_, xp_, out_) do { double z_, h_; b->Z_fn((t_), (xp_), b->theta, b->known, &z_); \
    b->H_fn((t_), (xp_), b->theta, b->known, &h_); (out_) = dnorm_sd(b->y[t_], z_, h_); } while (0)
  gen_ekf_update_step(mod, 0, b->a1, b->P1, att1, Ptt1); /* :1157-1158 */
  orc_psd_chol(Ptt1, m, L);                                /* :1161 */
  for (unsigned i = 0; i < N; i++) {
    rng_normals_wide(rng, 0, 0, i, m, um);
    for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < m; c++) s += L[j + m * c] * um[c]; A_(j, 0, i) = att1[j] + s; }
  }
  double loglik = 0.0, inc;
  if (isfinite(b->y[0])) {
    for (unsigned i = 0; i < N; i++) {
      double lo; GEN_LOG_OBS(0, &A_(0, 0, i), lo);
      lw[i] = lo + dmvnorm_cov(&A_(0, 0, i), b->a1, b->P1, m) - dmvnorm_lwr(&A_(0, 0, i), att1, L, m);
    }
    if (!normalise_step(lw, N, weights, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
    loglik = inc;
  } else for (unsigned i = 0; i < N; i++) { weights[i] = 1.0; nw[i] = 1.0 / N; }
  for (int t = 0; t < n; t++) {
    for (unsigned i = 0; i < N; i++) r[i] = orc_rng_uniform(rng, 1, t, i, 0); /* functor-model filters draw WIDE blocks */
    filter_resample(nw, r, N, indices + (size_t)N * t, t);
    for (unsigned i = 0; i < N; i++) { /* :1212-1227 */
      double at[ORC_MAXM], Rt[ORC_MAXM * ORC_MAXM], Pt[ORC_MAXM * ORC_MAXM], Ptt[ORC_MAXM * ORC_MAXM];
      for (int j = 0; j < m; j++) atmp[j + (size_t)m * i] = A_(j, t, indices[i + (size_t)N * t]);
      b->R_fn(t, atmp + (size_t)m * i, b->theta, b->known, Rt);
      compute_RR(Rt, m, k, Pt);
      b->T_fn(t, atmp + (size_t)m * i, b->theta, b->known, at);
      if (t < n - 1) {
        gen_ekf_update_step(mod, t + 1, at, Pt, att + (size_t)m * i, Ptt);
        orc_psd_chol(Ptt, m, Lp + (size_t)mm * i);
      } else { /* the last, never-weighted slot uses Pt itself as the factor (:1223-1226) */
        memcpy(att + (size_t)m * i, at, sizeof(double) * m); memcpy(Lp + (size_t)mm * i, Pt, sizeof(double) * mm);
      }
    }
    for (unsigned i = 0; i < N; i++) {
      rng_normals_wide(rng, 2, t, i, m, um);
      const double *Li = Lp + (size_t)mm * i;
      for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < m; c++) s += Li[j + m * c] * um[c]; A_(j, t + 1, i) = att[j + (size_t)m * i] + s; }
    }
    double *wcol = weights + (size_t)N * (t + 1);
    if ((t < n - 1) && isfinite(b->y[t + 1])) { /* :1236-1256 */
      for (unsigned i = 0; i < N; i++) {
        double mean[ORC_MAXM], Rt[ORC_MAXM * ORC_MAXM], RR[ORC_MAXM * ORC_MAXM], lo;
        b->R_fn(t, atmp + (size_t)m * i, b->theta, b->known, Rt);
        compute_RR(Rt, m, k, RR);
        b->T_fn(t, atmp + (size_t)m * i, b->theta, b->known, mean);
        GEN_LOG_OBS(t + 1, &A_(0, t + 1, i), lo);
        lw[i] = lo + dmvnorm_cov(&A_(0, t + 1, i), mean, RR, m) - dmvnorm_lwr(&A_(0, t + 1, i), att + (size_t)m * i, Lp + (size_t)mm * i, m);
      }
      if (!normalise_step(lw, N, wcol, 1, nw, &inc)) { loglik = -INFINITY; goto done; }
      loglik += inc;
    } else for (unsigned i = 0; i < N; i++) { wcol[i] = 1.0; nw[i] = 1.0 / N; }
  }
#undef GEN_LOG_OBS
done:
  free(nw); free(lw); free(r); free(atmp); free(att); free(Lp);
  return loglik;
}
#undef GEN_CALL


/* ssm_ulg::psi_filter, src/model_ssm_ulg.cpp:644-672: exact draws from the joint smoothing distribution */
void orc_ulg_psi_filter(const orc_ulg_model *g, unsigned N, const orc_rng *rng, double *alpha) {
  const int n = g->n, m = g->m, mm = m * m;
  double *ah = (double *)calloc((size_t)m * (n + 1), sizeof(double));
  double *Vt = (double *)calloc((size_t)mm * (n + 1), sizeof(double)), *Ct = (double *)calloc((size_t)mm * (n + 1), sizeof(double));
  double um[ORC_MAXM];
  orc_ulg_smoother_ccov(g, ah, Vt, Ct);
  orc_conditional_cov(Vt, Ct, m, n + 1);
  for (unsigned i = 0; i < N; i++) {
    rng_normals(rng, 0, 0, i, m, um);
    for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < m; c++) s += Vt[j + m * c] * um[c]; A_(j, 0, i) = ah[j] + s; }
  }
  for (int t = 0; t < n; t++)
    for (unsigned i = 0; i < N; i++) {
      rng_normals(rng, 2, t, i, m, um);
      for (int j = 0; j < m; j++) {
        double s = 0, q = 0;
        for (int c = 0; c < m; c++) s += Ct[j + m * c + (size_t)mm * (t + 1)] * (A_(c, t, i) - ah[c + (size_t)m * t]);
        for (int c = 0; c < m; c++) q += Vt[j + m * c + (size_t)mm * (t + 1)] * um[c];
        A_(j, t + 1, i) = ah[j + (size_t)m * (t + 1)] + s + q;
      }
    }
  free(ah); free(Vt); free(Ct);
}

/* ssm_mlg::psi_filter, src/model_ssm_mlg.cpp:473-501 (gaussian_psi_smoother case 0, src/R_psi.cpp:20-25): smoother_ccov
 * (:310-396) + conditional_cov, then exact draws from the joint smoothing distribution.  `mod` carries y (p x n), Z, T, R,
 * a1, P1, D, C; H is the p x p x (1|n) lower factor of the observation noise (HH = H H', compute_HH). */
int orc_mlg_psi_filter(const orc_mng_model *mod, const double *H, int Htv, unsigned N, const orc_rng *rng, double *alpha) {
  const int n = mod->n, m = mod->m, p = mod->p, mm = m * m, ns = Htv ? n : 1;
  double *HH = (double *)calloc((size_t)p * p * ns, sizeof(double));
  for (int s = 0; s < ns; s++) mat_mul_bt(H + (size_t)p * p * s, H + (size_t)p * p * s, p, p, p, HH + (size_t)p * p * s);
  double *ah = (double *)calloc((size_t)m * (n + 1), sizeof(double));
  double *Vt = (double *)calloc((size_t)mm * (n + 1), sizeof(double)), *Ct = (double *)calloc((size_t)mm * (n + 1), sizeof(double));
  double um[ORC_MAXM];
  g_mlg_HH = HH; g_mlg_HHtv = Htv;
  const int ok = mlg_smoother_ccov(mod, mod->y, NULL, ah, Vt, Ct);
  g_mlg_HH = NULL;
  if (ok) {
    orc_conditional_cov(Vt, Ct, m, n + 1);
    for (unsigned i = 0; i < N; i++) {
      rng_normals(rng, 0, 0, i, m, um);
      for (int j = 0; j < m; j++) { double s = 0; for (int c = 0; c < m; c++) s += Vt[j + m * c] * um[c]; A_(j, 0, i) = ah[j] + s; }
    }
    for (int t = 0; t < n; t++)
      for (unsigned i = 0; i < N; i++) {
        rng_normals(rng, 2, t, i, m, um);
        for (int j = 0; j < m; j++) {
          double s = 0, q = 0;
          for (int c = 0; c < m; c++) s += Ct[j + m * c + (size_t)mm * (t + 1)] * (A_(c, t, i) - ah[c + (size_t)m * t]);
          for (int c = 0; c < m; c++) q += Vt[j + m * c + (size_t)mm * (t + 1)] * um[c];
          A_(j, t + 1, i) = ah[j + (size_t)m * (t + 1)] + s + q;
        }
      }
  }
  free(HH); free(ah); free(Vt); free(Ct);
  return ok;
}
