// host_model.cpp -- host-side model handling for the MI355X particle-filter engine.
// The O(n m^3) Gaussian-approximation set-up of the psi-APF stays on the host (SURVEY.md
// section 8a row a12: serial per theta, small); it is threaded over theta-draws by the caller.
#include "host_model.h"
#include "options.h"

#include <algorithm>
#include <cstdio>
#include <chrono>
#include <thread>
#include <atomic>
#include <cmath>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "lg_smoother.h"
#include "pf_device.h"

namespace bssm {

namespace {

constexpr double kZeroTol = 1e-12;  // src/model_ssm_ung.h:21, src/model_ssm_ulg.h:19
constexpr double kEps = std::numeric_limits<double>::epsilon();
constexpr double kPi = 3.14159265358979323846;

// ------------------------------------------------------------------ tiny dense matrix
// Small matrices live inline (no heap): the psi set-up builds thousands of these per draw, and with a std::vector behind each one
// the allocator was most of the set-up time.  Round 6: 16 x 16 inline (every model of up to 16 states -- 2 KB per Mat instead of the
// 32 KB a 64 x 64 inline array cost EVERY model: ~1 MB of stack per set-up thread, a page touched per m x 1 vector); larger ones
// (seasonal components of period 24 / 52, R/models.R:995-997) spill to the heap.  Copies move only the r * c entries in use.
struct Mat {
  static constexpr int kInline = 16 * 16;
  int r = 0, c = 0;
  double* a = inl;
  double inl[kInline];
  std::unique_ptr<double[]> heap;
  void room(int n) {   // storage for n entries; the old contents are not kept
    if (n > kInline) { heap.reset(new double[(size_t)n]); a = heap.get(); }
    else { heap.reset(); a = inl; }
  }
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_) { room(r * c); for (int i = 0; i < r * c; i++) a[i] = 0.0; }
  Mat(int r_, int c_, const double* src) : r(r_), c(c_) { room(r * c); std::memcpy(a, src, sizeof(double) * (size_t)(r * c)); }
  Mat(const Mat& o) : r(o.r), c(o.c) { room(r * c); std::memcpy(a, o.a, sizeof(double) * (size_t)(r * c)); }
  Mat(Mat&& o) noexcept : r(o.r), c(o.c) {
    if (o.heap) { heap = std::move(o.heap); a = heap.get(); }
    else std::memcpy(inl, o.inl, sizeof(double) * (size_t)(r * c));
    o.r = o.c = 0; o.a = o.inl;
  }
  Mat& operator=(const Mat& o) {
    if (this == &o) return *this;
    if (o.size() != size() || o.size() <= kInline) room(o.size());
    r = o.r; c = o.c;
    std::memcpy(a, o.a, sizeof(double) * (size_t)(r * c));
    return *this;
  }
  Mat& operator=(Mat&& o) noexcept {
    if (this == &o) return *this;
    r = o.r; c = o.c;
    if (o.heap) { heap = std::move(o.heap); a = heap.get(); }
    else { heap.reset(); a = inl; std::memcpy(inl, o.inl, sizeof(double) * (size_t)(r * c)); }
    o.r = o.c = 0; o.a = o.inl;
    return *this;
  }
  int size() const { return r * c; }
  double& operator()(int i, int j) { return a[i + r * j]; }
  double operator()(int i, int j) const { return a[i + r * j]; }
  static Mat eye(int m) { Mat I(m, m); for (int i = 0; i < m; i++) I(i, i) = 1.0; return I; }
};
Mat operator*(const Mat& A, const Mat& B) {
  Mat C(A.r, B.c);
  for (int j = 0; j < B.c; j++)
    for (int q = 0; q < A.c; q++) {
      const double b = B(q, j);
      for (int i = 0; i < A.r; i++) C(i, j) += A(i, q) * b;
    }
  return C;
}
Mat operator+(const Mat& A, const Mat& B) { Mat C = A; for (int i = 0; i < C.size(); i++) C.a[i] += B.a[i]; return C; }
Mat operator-(const Mat& A, const Mat& B) { Mat C = A; for (int i = 0; i < C.size(); i++) C.a[i] -= B.a[i]; return C; }
Mat operator*(const Mat& A, double s) { Mat C = A; for (int i = 0; i < C.size(); i++) C.a[i] *= s; return C; }
Mat tr(const Mat& A) { Mat B(A.c, A.r); for (int j = 0; j < A.c; j++) for (int i = 0; i < A.r; i++) B(j, i) = A(i, j); return B; }
void symmatu(Mat& A) { for (int j = 0; j < A.c; j++) for (int i = j + 1; i < A.r; i++) A(i, j) = A(j, i); }

// lower Cholesky restricted to the index set `idx`, zeros elsewhere
bool chol_lower_on(const Mat& P, const std::vector<int>& idx, Mat* L) {
  const int m = P.r;
  *L = Mat(m, m);
  for (size_t a = 0; a < idx.size(); a++)
    for (size_t b = 0; b <= a; b++) {
      double s = P(idx[a], idx[b]);
      for (size_t q = 0; q < b; q++) s -= (*L)(idx[a], idx[q]) * (*L)(idx[b], idx[q]);
      if (a == b) { if (!(s > 0.0)) return false; (*L)(idx[a], idx[a]) = std::sqrt(s); }
      else (*L)(idx[a], idx[b]) = s / (*L)(idx[b], idx[b]);
    }
  return true;
}

// eigen-decomposition of a symmetric matrix by cyclic Jacobi; returns |eigenvalues| descending and the
// matching vectors (columns), i.e. the left singular pairs arma::svd_econ(...,"left") yields for a PSD input
void sym_singular(const Mat& Ain, Mat* U, std::vector<double>* s) {
  const int m = Ain.r;
  Mat A(m, m);
  for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) A(i, j) = 0.5 * (Ain(i, j) + Ain(j, i));
  *U = Mat::eye(m);
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0.0, dg = 0.0;
    for (int j = 0; j < m; j++) for (int i = 0; i < j; i++) off += A(i, j) * A(i, j);
    for (int j = 0; j < m; j++) dg += A(j, j) * A(j, j);
    if (off < 1e-300 || off < 1e-60 * dg) break;  // (the same stopping rule as LgSmoother::sym_singular, lg_smoother.h)
    for (int p = 0; p < m - 1; p++)
      for (int q = p + 1; q < m; q++) {
        const double apq = A(p, q);
        if (apq == 0.0) continue;
        const double tau = (A(q, q) - A(p, p)) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = t * cs;
        for (int r = 0; r < m; r++) { const double x = A(r, p), y = A(r, q); A(r, p) = cs * x - sn * y; A(r, q) = sn * x + cs * y; }
        for (int r = 0; r < m; r++) { const double x = A(p, r), y = A(q, r); A(p, r) = cs * x - sn * y; A(q, r) = sn * x + cs * y; }
        for (int r = 0; r < m; r++) { const double x = (*U)(r, p), y = (*U)(r, q); (*U)(r, p) = cs * x - sn * y; (*U)(r, q) = sn * x + cs * y; }
      }
  }
  s->resize(m);
  std::vector<int> ord(m);
  for (int j = 0; j < m; j++) { (*s)[j] = std::fabs(A(j, j)); ord[j] = j; }
  std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return (*s)[a] > (*s)[b]; });
  Mat V(m, m);
  std::vector<double> sv(m);
  for (int j = 0; j < m; j++) {
    sv[j] = (*s)[ord[j]];
    int big = 0;
    for (int r = 1; r < m; r++) if (std::fabs((*U)(r, ord[j])) > std::fabs((*U)(big, ord[j]))) big = r;
    const double sg = (*U)(big, ord[j]) < 0 ? -1.0 : 1.0;  // fix the (arbitrary) column sign
    for (int r = 0; r < m; r++) V(r, j) = sg * (*U)(r, ord[j]);
  }
  *U = V;
  *s = sv;
}

double lchoose_ts(double nn, double kk) {  // src/distr_consts.cpp:7-9
  return -std::log(nn + 1) - std::lgamma(nn - kk + 1) - std::lgamma(kk + 1) + std::lgamma(nn + 2);
}
double norm_log_const(double sd) { return -0.5 * std::log(2.0 * kPi) - std::log(sd); }

// views into a HostModel at time t (observation arrays are p-major: Z is p x m)
struct Sys {
  const HostModel& h;
  explicit Sys(const HostModel& hm) : h(hm) {}
  Mat Zm(int t) const { return Mat(h.p, h.m, h.Z.data() + (size_t)h.p * h.m * (t * h.Ztv)); }
  Mat Tm(int t) const { return Mat(h.m, h.m, h.T.data() + (size_t)h.m * h.m * (t * h.Ttv)); }
  Mat Rm(int t) const { return Mat(h.m, h.k, h.R.data() + (size_t)h.m * h.k * (t * h.Rtv)); }
  Mat Cc(int t) const { return Mat(h.m, 1, h.C.data() + (size_t)h.m * (t * h.Ctv)); }
  double Dv(int j, int t) const { return h.D[j + (size_t)h.p * (t * h.Dtv)]; }
};

// ------------------------------------------------------------------ Kalman machinery of the approximating model
struct Approx {  // pseudo-observations of the Gaussian approximation, p x n (H = diagonal sd)
  std::vector<double> y, H;
  // a genuine ssm_mlg model instead: full observation-noise covariance HH = H H', p x p x (1|n) (src/model_ssm_mlg.cpp compute_HH)
  std::vector<double> HHfull;
  int HHtv = 0;
};

// one prediction step.  Univariate: src/model_ssm_ulg.cpp:122-136 / :268-283; multivariate with missing
// series zeroed out of Z and given unit variance: src/model_ssm_mlg.cpp:259-291, 318-356.
struct KfStep {
  int nobs = 0;       // observed series at t (0 = pure prediction)
  bool ok = true;     // F finite and positive definite
  double logdetF = 0, quad = 0;  // log|F|, v' F^-1 v over the observed series
  double Fs = 0;                 // the scalar F when p == 1
  Mat v, ZF, K, Zz;   // innovations (p), Z' F^-1 (m x p), gain (m x p), Z with missing rows zeroed (p x m)
};

KfStep kf_predict(const Sys& S, const Approx& ap, int t, const Mat& at, const Mat& Pt, bool joseph, Mat* at1, Mat* Pt1) {
  const HostModel& h = S.h;
  const int m = h.m, p = h.p;
  const Mat T = S.Tm(t), R = S.Rm(t);
  const Mat RR = R * tr(R);
  KfStep st;
  st.v = Mat(p, 1); st.ZF = Mat(m, p); st.K = Mat(m, p); st.Zz = S.Zm(t);
  Mat HH(p, p);
  std::vector<char> na(p);
  for (int j = 0; j < p; j++) {
    const double yt = ap.y[j + (size_t)p * t] - (p == 1 ? h.xbeta[t] : 0.0);
    na[j] = !std::isfinite(yt);
    if (na[j]) { for (int c = 0; c < m; c++) st.Zz(j, c) = 0.0; HH(j, j) = 1.0; }
    else { if (ap.HHfull.empty()) HH(j, j) = ap.H[j + (size_t)p * t] * ap.H[j + (size_t)p * t]; st.nobs++; }
  }
  if (!ap.HHfull.empty()) {  // rows / columns of missing series zeroed, unit diagonal there (src/model_ssm_mlg.cpp:324-328)
    const double* Hs = ap.HHfull.data() + (size_t)p * p * (t * ap.HHtv);
    for (int j = 0; j < p; j++) for (int i = 0; i < p; i++) HH(i, j) = (na[i] || na[j]) ? (i == j ? 1.0 : 0.0) : Hs[i + (size_t)p * j];
  }
  Mat F = st.Zz * Pt * tr(st.Zz) + HH;
  if (p == 1 && st.nobs == 1 && !(F(0, 0) > kZeroTol)) st.nobs = 0;  // univariate guard F > zero_tol (:123)
  if (st.nobs == 0) {
    *at1 = S.Cc(t) + T * at;
    *Pt1 = T * Pt * tr(T) + RR;
    symmatu(*Pt1);
    return st;
  }
  if (p == 1) {  // scalar form of src/model_ssm_ulg.cpp:124-127: K = P Z / F, no factorisation
    const double Fs = F(0, 0);
    st.Fs = Fs;
    double za = 0;
    for (int c = 0; c < m; c++) za += st.Zz(0, c) * at(c, 0);
    st.v(0, 0) = ap.y[t] - h.xbeta[t] - S.Dv(0, t) - za;
    const Mat PZ = Pt * tr(st.Zz);
    for (int i = 0; i < m; i++) { st.K(i, 0) = PZ(i, 0) / Fs; st.ZF(i, 0) = st.Zz(0, i) / Fs; }
    st.logdetF = std::log(Fs);
    st.quad = st.v(0, 0) * st.v(0, 0) / Fs;
  } else {
    for (int q = 0; q < F.size(); q++) if (!std::isfinite(F.a[q])) { st.ok = false; return st; }
    for (int j = 0; j < p; j++) if (!(F(j, j) > 0)) { st.ok = false; return st; }
    std::vector<int> all(p);
    for (int j = 0; j < p; j++) all[j] = j;
    Mat L;
    if (!chol_lower_on(F, all, &L)) { st.ok = false; return st; }
    Mat Li(p, p);  // inverse of the lower factor
    for (int j = 0; j < p; j++) {
      Li(j, j) = 1.0 / L(j, j);
      for (int i = j + 1; i < p; i++) { double sacc = 0; for (int q = j; q < i; q++) sacc += L(i, q) * Li(q, j); Li(i, j) = -sacc / L(i, i); }
    }
    const Mat Fi = tr(Li) * Li;
    for (int j = 0; j < p; j++) {
      st.logdetF += 2.0 * std::log(L(j, j));
      if (na[j]) continue;
      double za = 0;
      for (int c = 0; c < m; c++) za += st.Zz(j, c) * at(c, 0);
      st.v(j, 0) = ap.y[j + (size_t)p * t] - S.Dv(j, t) - za;
    }
    const Mat w = Li * st.v;
    for (int j = 0; j < p; j++) st.quad += w(j, 0) * w(j, 0);
    st.ZF = tr(st.Zz) * Fi;
    st.K = Pt * st.ZF;
  }
  *at1 = S.Cc(t) + T * (at + st.K * st.v);
  Mat inner;
  if (joseph) {
    const Mat IKZ = Mat::eye(m) - st.K * st.Zz;
    inner = IKZ * Pt * tr(IKZ) + st.K * HH * tr(st.K);
  } else {  // univariate log_likelihood() uses P - K K' F (src/model_ssm_ulg.cpp:129)
    inner = Pt - (st.K * F) * tr(st.K);
  }
  *Pt1 = T * inner * tr(T) + RR;
  symmatu(*Pt1);
  return st;
}

// src/model_ssm_ulg.cpp:106-138, src/model_ssm_mlg.cpp:90-137
double gaussian_loglik(const Sys& S, const Approx& ap) {
  const HostModel& h = S.h;
  double accH = 0, accR = 0;
  for (double v : ap.H) accH += v;
  for (double v : h.R) accR += v;
  if (accH + accR < kZeroTol) return -INFINITY;
  Mat at(h.m, 1, h.a1.data()), Pt(h.m, h.m, h.P1.data()), at1, Pt1;
  const double LOG2PI = std::log(2.0 * kPi);
  double ll = 0.0;
  for (int t = 0; t < h.n; t++) {
    const KfStep st = kf_predict(S, ap, t, at, Pt, h.p > 1, &at1, &Pt1);
    if (!st.ok) return -INFINITY;
    if (st.nobs > 0) ll -= 0.5 * (st.nobs * LOG2PI + st.logdetF + st.quad);
    at = at1; Pt = Pt1;
  }
  return ll;
}

// src/model_ssm_ulg.cpp:252-305, src/model_ssm_mlg.cpp fast_smoother -> smoothed state means, m x (n+1)
bool lg_fast_mean(const HostModel& h, const Approx& ap, std::vector<Mat>* atp, double* gll = nullptr);

// gll (optional): set to the Gaussian log-likelihood of the same model when the fixed-size path ran (its forward pass has
// the innovations and their variances), left untouched otherwise
bool fast_smoother(const Sys& S, const Approx& ap, std::vector<Mat>* atp, double* gll = nullptr) {
  const HostModel& h = S.h;
  const int n = h.n, m = h.m;
  if (lg_fast_mean(h, ap, atp, gll)) return true;
  std::vector<Mat>& at = *atp;
  at.assign(n + 1, Mat(m, 1));
  std::vector<KfStep> st(n);
  at[0] = Mat(m, 1, h.a1.data());
  Mat Pt(m, m, h.P1.data()), Pt1;
  for (int t = 0; t < n; t++) {
    st[t] = kf_predict(S, ap, t, at[t], Pt, true, &at[t + 1], &Pt1);
    if (!st[t].ok) return false;
    Pt = Pt1;
  }
  std::vector<Mat> rt(n, Mat(m, 1));
  Mat rprev(m, 1);
  for (int t = n - 1; t >= 0; t--) {
    Mat r0;
    if (st[t].nobs > 0) {
      const Mat L = S.Tm(t) * (Mat::eye(m) - st[t].K * st[t].Zz);
      r0 = st[t].ZF * st[t].v + tr(L) * rt[t];
    } else {
      r0 = tr(S.Tm(t)) * rt[t];
    }
    if (t > 0) rt[t - 1] = r0; else rprev = r0;
  }
  at[0] = Mat(m, 1, h.a1.data()) + Mat(m, m, h.P1.data()) * rprev;
  for (int t = 0; t < n - 1; t++) {
    const Mat R = S.Rm(t);
    at[t + 1] = S.Cc(t) + S.Tm(t) * at[t] + (R * tr(R)) * rt[t];
  }
  return true;
}

// src/model_ssm_ulg.cpp:419-473, src/model_ssm_mlg.cpp:310-396
bool smoother_ccov(const Sys& S, const Approx& ap, std::vector<Mat>* at, std::vector<Mat>* Pt, std::vector<Mat>* cc) {
  const HostModel& h = S.h;
  const int n = h.n, m = h.m;
  at->assign(n + 1, Mat(m, 1));
  Pt->assign(n + 1, Mat(m, m));
  cc->assign(n + 1, Mat(m, m));
  std::vector<KfStep> st(n);
  (*at)[0] = Mat(m, 1, h.a1.data());
  (*Pt)[0] = Mat(m, m, h.P1.data());
  for (int t = 0; t < n; t++) {
    st[t] = kf_predict(S, ap, t, (*at)[t], (*Pt)[t], true, &(*at)[t + 1], &(*Pt)[t + 1]);
    if (!st[t].ok) return false;
    (*cc)[t] = (*Pt)[t + 1];
  }
  Mat rt(m, 1), Nt(m, m);
  const Mat I = Mat::eye(m);
  for (int t = n - 1; t >= 0; t--) {
    const Mat T = S.Tm(t);
    if (st[t].nobs > 0) {
      const Mat L = T * (I - st[t].K * st[t].Zz);
      (*cc)[t] = (*Pt)[t] * tr(L) * (I - Nt * (*cc)[t]);
      rt = st[t].ZF * st[t].v + tr(L) * rt;
      Mat N2 = tr(L) * Nt * L;
      if (h.p == 1) {  // Z Z' / F in the reference's operation order (src/model_ssm_ulg.cpp:462)
        for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) N2(i, j) = st[t].Zz(0, i) * st[t].Zz(0, j) / st[t].Fs + N2(i, j);
      } else {
        N2 = st[t].ZF * st[t].Zz + N2;
      }
      Nt = N2;
    } else {
      (*cc)[t] = (*Pt)[t] * tr(T) * (I - Nt * (*cc)[t]);
      rt = tr(T) * rt;
      Nt = tr(T) * Nt * T;
    }
    symmatu(Nt);
    (*at)[t] = (*at)[t] + (*Pt)[t] * rt;
    Mat dP = (*Pt)[t] * Nt * (*Pt)[t];
    symmatu(dP);
    (*Pt)[t] = (*Pt)[t] - dP;
  }
  (*cc)[n] = Mat(m, m);
  return true;
}

// src/conditional_dist.cpp:4-31 (use_svd = true).  Round 6: factor columns whose singular value is at the rounding level of the
// subtraction that formed the matrix -- s_c <= eps m trace(Vt[t] before the subtraction) -- are ZERO (the rule and its reason: the header
// of orc_conditional_cov in oracle/bssm_oracle.c; LgSmoother<M>::cond_step and the factor kernels apply the same cut)
namespace {
void factor_with_cut(const Mat& U, const std::vector<double>& s, double tr, Mat* out) {
  const int m = U.r;
  const double thr = kEps * m * tr;
  for (int c = 0; c < m; c++) {
    const double f = s[c] > thr ? std::sqrt(s[c]) : 0.0;
    for (int i = 0; i < m; i++) (*out)(i, c) = f == 0.0 ? 0.0 : U(i, c) * f;
  }
}
double mat_trace(const Mat& A) { double t = 0.0; for (int i = 0; i < A.r; i++) t += A(i, i); return t; }
}  // namespace
// one time point of conditional_cov from the smoother's (unchanged) moments: V_{t-1}, C_{t-1}, V_t -> the factor of Var(alpha_t | alpha_{t-1}, y) and C_t
static void conditional_cov_step(const Mat& Vprev, const Mat& Cprev, const Mat& Vcur, Mat* Vout, Mat* Cout) {
  const int m = Vprev.r;
  Mat U;
  std::vector<double> s;
  sym_singular(Vprev, &U, &s);
  Mat W(m, m);
  for (int c = 0; c < m; c++) {
    if (!(s[c] > kEps * m * s[0])) continue;
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) W(i, j) += U(i, c) * U(j, c) / s[c];
  }
  const Mat tmp = tr(Cprev) * W;
  const double trc = mat_trace(Vcur);
  Mat D = Vcur - tmp * Cprev;
  *Cout = tmp;
  sym_singular(D, &U, &s);
  factor_with_cut(U, s, trc, &D);
  *Vout = D;
}
// (late round 6) The time points are independent given the smoother's moments -- the serial loop runs downwards only so that it can work in place -- and a
// model with many states spends its whole single psi call here (53 states, T = 200: 0.58 s of a 0.68 s call, two 53 x 53 Jacobi decompositions per time
// point).  With 9 and more states and nobody else inside, the time points go to host threads on a copy of the moments: the same arithmetic per time point,
// the same bits.
static std::atomic<int> g_condcov_active{0};
void conditional_cov(std::vector<Mat>* Vt, std::vector<Mat>* Ct) {
  const int n1 = (int)Vt->size(), m = (*Vt)[0].r;
  const int before = g_condcov_active.fetch_add(1);
  const unsigned hw = std::thread::hardware_concurrency();
  const int forced = bssm::opt().host_factor_threads;   // (0: automatic; 1: the serial loop -- the comparison the tests make)
  const int nthr = forced > 0 ? std::min(forced, std::max(1, n1 - 1))
                              : ((before == 0 && m >= 9 && n1 > 8) ? (int)std::min<unsigned>(std::min<unsigned>(hw ? hw : 1u, m >= 16 ? 32u : 16u), (unsigned)(n1 - 1)) : 1);
  if (nthr > 1) {
    const std::vector<Mat> V0 = *Vt, C0 = *Ct;
    std::atomic<int> next{1};
    auto work = [&]() { for (int t = next++; t < n1; t = next++) conditional_cov_step(V0[t - 1], C0[t - 1], V0[t], &(*Vt)[t], &(*Ct)[t]); };
    std::vector<std::thread> pool;
    try {
      for (int q = 1; q < nthr; q++) pool.emplace_back(work);
    } catch (...) {   // (no more threads to be had: the ones that started and this one share the time points)
    }
    work();
    for (auto& th : pool) th.join();
  } else {
    for (int t = n1 - 1; t > 0; t--) {
      Mat Vn, Cn;
      conditional_cov_step((*Vt)[t - 1], (*Ct)[t - 1], (*Vt)[t], &Vn, &Cn);
      (*Vt)[t] = Vn; (*Ct)[t] = Cn;
    }
  }
  g_condcov_active.fetch_sub(1);
  Mat U;
  std::vector<double> s;
  const double tr0 = mat_trace((*Vt)[0]);
  sym_singular((*Vt)[0], &U, &s);
  factor_with_cut(U, s, tr0, &(*Vt)[0]);
}

// Fixed-size fast paths of smoother_ccov + conditional_cov and of fast_smoother for univariate observations and
// m <= 8 (lg_smoother.h: the same operation order as the generic routines above, so the numbers are the same).
struct LgHostModel {
  LgModel g;
  std::vector<double> RR, yx;
  static bool applies(const HostModel& h) {
    if (bssm::opt().generic_setup) return false;  // test hook: force the generic Mat routines
    return h.p == 1 && h.m >= 1 && h.m <= 8;
  }
  LgHostModel(const HostModel& h, const Approx& ap) {
    const int n = h.n, m = h.m, k = h.k, nR = h.Rtv ? n : 1;
    RR.resize((size_t)nR * m * m); yx.resize(n);
    for (int t = 0; t < nR; t++) {
      const double* R = h.R.data() + (size_t)m * k * t;
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) {
        double sacc = 0.0;
        for (int q = 0; q < k; q++) sacc += R[i + m * q] * R[j + m * q];
        RR[i + m * j + (size_t)m * m * t] = sacc;
      }
    }
    for (int t = 0; t < n; t++) yx[t] = ap.y[t] - h.xbeta[t];
    g.n = n;
    g.Z = h.Z.data(); g.Zs = h.Ztv ? m : 0;
    g.T = h.T.data(); g.Ts = h.Ttv ? m * m : 0;
    g.RR = RR.data(); g.Rs = h.Rtv ? m * m : 0;
    g.D = h.D.data(); g.Ds = h.Dtv ? 1 : 0;
    g.C = h.C.data(); g.Cs = h.Ctv ? m : 0;
    g.y = yx.data();
    g.H = ap.H.data(); g.Hs = 1;
    g.a1 = h.a1.data(); g.P1 = h.P1.data();
  }
};
bool lg_fast_mean(const HostModel& h, const Approx& ap, std::vector<Mat>* atp, double* gll) {
  if (!LgHostModel::applies(h)) return false;
  const int n = h.n, m = h.m;
  LgHostModel lm(h, ap);
  std::vector<double> at((size_t)m * (n + 1)), rts((size_t)n * m), K((size_t)n * m), v(n), Fs(n);
  lg_smoother_mean(m, lm.g, at.data(), rts.data(), K.data(), v.data(), Fs.data());
  if (gll) {  // gaussian_loglik() of the same model from this forward pass: the same v_t, F_t and the same guards
    double accH = 0, accR = 0;
    for (double q : ap.H) accH += q;
    for (double q : h.R) accR += q;
    if (accH + accR < kZeroTol) {
      *gll = -INFINITY;
    } else {
      const double LOG2PI = std::log(2.0 * kPi);
      double ll = 0.0;
      for (int t = 0; t < n; t++) if (Fs[t] > 0.0) ll -= 0.5 * (LOG2PI + std::log(Fs[t]) + v[t] * v[t] / Fs[t]);
      *gll = ll;
    }
  }
  atp->assign(n + 1, Mat(m, 1));
  for (int t = 0; t <= n; t++) for (int i = 0; i < m; i++) (*atp)[t](i, 0) = at[i + (size_t)m * t];
  return true;
}
// Returns false when it does not apply; *ok then is untouched.
bool lg_fast_path(const HostModel& h, const Approx& ap, std::vector<double>* alphahat, std::vector<double>* Vt,
                  std::vector<double>* Ct, bool* ok) {
  if (!LgHostModel::applies(h)) return false;
  const int n = h.n, m = h.m;
  LgHostModel lm(h, ap);
  std::vector<double> K((size_t)n * m), v(n), Fs(n);
  alphahat->assign((size_t)m * (n + 1), 0.0);
  Vt->assign((size_t)m * m * (n + 1), 0.0);
  Ct->assign((size_t)m * m * (n + 1), 0.0);
  *ok = lg_smoother_run(m, lm.g, alphahat->data(), Vt->data(), Ct->data(), K.data(), v.data(), Fs.data());
  return true;
}

// src/model_ssm_ung.cpp:302-342, src/model_ssm_mng.cpp:269-315
void laplace_iter(const HostModel& h, const std::vector<double>& signal, Approx* ap) {
  const int p = h.p;
  for (int t = 0; t < h.n; t++)
    for (int j = 0; j < p; j++) {
      const size_t q = j + (size_t)p * t;
      const double y = h.y[q], sg = signal[q], u = h.u[q], phi = h.phis[j];
      double HH = 0, yy = 0;
      switch (h.dist[j]) {
        case BSSM_SVM: {
          double tmp = y;
          if (std::fabs(tmp) < 1e-4) tmp = 1e-4;
          HH = 2.0 * std::exp(sg) / ((tmp / phi) * (tmp / phi));
          yy = sg + 1.0 - 0.5 * HH;
        } break;
        case BSSM_POISSON: HH = 1.0 / (std::exp(sg) * u); yy = y * HH + sg - 1.0; break;
        case BSSM_BINOMIAL: { const double e = std::exp(sg); HH = (1.0 + e) * (1.0 + e) / (u * e); yy = y * HH + sg - 1.0 - e; } break;
        case BSSM_NEGBIN: { const double e = std::exp(sg) * u; HH = (phi + e) * (phi + e) / (phi * e * (y + phi)); yy = sg + (phi + e) * (y - e) / ((y + phi) * e); } break;
        case BSSM_GAMMA: { const double e = std::exp(sg) * u; HH = e / (y * phi); yy = sg - e / y + 1; } break;
        case BSSM_GAUSSIAN: HH = phi * phi; yy = y; break;  // mng only (src/model_ssm_mng.cpp:307-311)
        default: break;
      }
      ap->y[q] = yy;
      ap->H[q] = std::sqrt(HH);
    }
}

// src/model_ssm_ung.cpp:241-292, src/model_ssm_mng.cpp:218-255
void update_scales(const HostModel& h, const Approx& ap, const std::vector<double>& mode, std::vector<double>* scales) {
  const int p = h.p;
  scales->assign(h.n, 0.0);
  for (int t = 0; t < h.n; t++)
    for (int j = 0; j < p; j++) {
      const size_t q = j + (size_t)p * t;
      const double y = h.y[q];
      if (!std::isfinite(y)) continue;
      const double md = mode[q], u = h.u[q], phi = h.phis[j];
      double v = 0;
      switch (h.dist[j]) {
        case BSSM_SVM: v = -0.5 * (md + std::pow(y / phi, 2.0) * std::exp(-md)); break;
        case BSSM_POISSON: v = y * md - u * std::exp(md); break;
        case BSSM_BINOMIAL: v = y * md - u * std::log1p(std::exp(md)); break;
        case BSSM_NEGBIN: v = y * md - (y + phi) * std::log(phi + u * std::exp(md)); break;
        case BSSM_GAMMA: v = p == 1 ? -phi * (md + (y * std::exp(-md) / u)) : -phi * md - (y * phi * std::exp(-md) / u); break;
        case BSSM_GAUSSIAN: v = -0.5 * std::pow((y - md) / phi, 2.0); break;
        default: break;
      }
      (*scales)[t] += v + 0.5 * std::pow((ap.y[q] - md) / ap.H[q], 2.0);
    }
}

// signal mode from smoothed states: src/model_ssm_ung.cpp:124-131, src/model_ssm_mng.cpp:101-105
void signal_from_states(const HostModel& h, const Sys& S, const std::vector<Mat>& sm, std::vector<double>* out) {
  const int p = h.p, m = h.m;
  out->resize((size_t)p * h.n);
  for (int t = 0; t < h.n; t++) {
    if (p == 1 && h.dist[0] == BSSM_SVM) { (*out)[t] = sm[t](0, 0); continue; }
    const Mat Z = S.Zm(t);
    for (int j = 0; j < p; j++) {
      double za = 0;
      for (int c = 0; c < m; c++) za += Z(j, c) * sm[t](c, 0);
      (*out)[j + (size_t)p * t] = (p == 1 ? h.xbeta[t] : 0.0) + S.Dv(j, t) + za;
    }
  }
}

// approximate(): src/model_ssm_ung.cpp:87-140, src/model_ssm_mng.cpp:73-114 (first call, approx_state == -1)
// gll (optional): Gaussian log-likelihood of the final approximating model when the last smoother pass could supply it
// (NaN otherwise): that pass ran on exactly the (approx y, H) the loop ends with
int run_laplace(const HostModel& h, const Sys& S, Approx* ap, std::vector<double>* mode, bool* ok, double* gll = nullptr) {
  if (gll) *gll = std::nan("");
  *mode = h.initial_mode;
  int it = 0;
  double diff = h.conv_tol + 1;
  std::vector<double> nm;
  std::vector<Mat> sm;
  *ok = true;
  while (it < h.max_iter && diff > h.conv_tol) {
    it++;
    laplace_iter(h, *mode, ap);
    if (gll) *gll = std::nan("");
    if (!fast_smoother(S, *ap, &sm, gll)) { *ok = false; break; }
    signal_from_states(h, S, sm, &nm);
    diff = 0.0;
    for (size_t q = 0; q < nm.size(); q++) diff += (nm[q] - (*mode)[q]) * (nm[q] - (*mode)[q]);
    diff /= (double)h.n * h.p;
    *mode = nm;
  }
  return it;
}

double approx_const_term(const HostModel& h, const Approx& ap) {  // src/model_ssm_ung.cpp:347-370, model_ssm_mng.cpp:318-355
  double c = const_term(h);
  for (size_t q = 0; q < h.y.size(); q++)
    if (std::isfinite(h.y[q])) c -= norm_log_const(ap.H[q]);
  return c;
}

}  // namespace

// ------------------------------------------------------------------ public pieces
int pad_dim(int d) { int p = 1; while (p < d) p <<= 1; return p; }
// padded noise dimension the kernels are instantiated for (pf_inst_mp*.hip): every power of two up to MP for MP <= 4,
// {2, 4, 8} for MP = 8 and {2, 4, 16} for MP = 16 -- the smallest of those that holds k.  The random-stream format of a step
// draw (pf_device.h: WIDE up to two normals, PACKED above) is a function of the REAL k; the kernels choose it from the padded
// KP, so KP > 2 must imply k > 2: every MP has an instantiation with KP = 2.
int pad_noise_dim(int MP, int k) {
  if (MP >= 32) {  // the matrix-core filters of 17 ... 64 states (pf_big.h) take the noise in chunks of four columns; the stream format follows the real k
    const int kp = (std::max(k, 1) + 3) / 4 * 4;
    return kp > MP ? MP : kp;
  }
  int kp = pad_dim(k);
  if (MP >= 8 && kp < 2) kp = 2;
  if (MP == 16) kp = kp <= 2 ? 2 : (kp <= 4 ? 4 : 16);
  return kp > MP ? MP : kp;
}

std::string ingest_model(const bssm_ung_model* s, bool need_mode, HostModel* o) {
  if (!s) return "model is NULL";
  if (s->n < 1 || s->m < 1 || s->k < 1) return "n, m, k must be positive";
  if (s->m > 64) return "state dimension m > 64 is not supported by the kernels";
  if (s->k > s->m) return "k (columns of R) must not exceed m";
  if (!s->y || !s->Z || !s->T || !s->R || !s->a1 || !s->P1) return "y, Z, T, R, a1, P1 must be non-NULL";
  const int n = s->n, m = s->m, k = s->k, p = s->p > 1 ? s->p : 1;
  if (p > 16) return "more than 16 observation series is not supported";
  o->n = n; o->m = m; o->k = k; o->p = p;
  o->Ztv = s->Ztv != 0; o->Ttv = s->Ttv != 0; o->Rtv = s->Rtv != 0; o->Dtv = s->Dtv != 0; o->Ctv = s->Ctv != 0; o->Htv = s->Htv != 0;
  o->max_iter = s->max_iter > 0 ? s->max_iter : 100;
  o->local_approx = s->local_approx;
  o->conv_tol = s->conv_tol > 0 ? s->conv_tol : 1e-8;
  if (p == 1) {
    if (s->distribution < 0 || s->distribution > 5) return "unknown distribution code";
    o->dist.assign(1, s->distribution);
    o->phis.assign(1, s->phi);
  } else {
    if (!s->distribution_p || !s->phi_p) return "multivariate model needs distribution_p and phi_p";
    o->dist.assign(s->distribution_p, s->distribution_p + p);
    o->phis.assign(s->phi_p, s->phi_p + p);
    for (int j = 0; j < p; j++) if (o->dist[j] < 1 || o->dist[j] > 5) return "multivariate distribution codes must be 1..5";
  }
  o->distribution = o->dist[0];
  o->phi = o->phis[0];
  // observation arrays -> p-major (the reference transposes y and u on entry, src/model_ssm_mng.cpp:8,17)
  o->y.resize((size_t)p * n); o->u.assign((size_t)p * n, 1.0);
  for (int t = 0; t < n; t++) for (int j = 0; j < p; j++) {
    o->y[j + (size_t)p * t] = s->y[t + (size_t)n * j];
    if (s->u) o->u[j + (size_t)p * t] = s->u[t + (size_t)n * j];
  }
  o->Z.assign(s->Z, s->Z + (size_t)p * m * (o->Ztv ? n : 1));
  o->T.assign(s->T, s->T + (size_t)m * m * (o->Ttv ? n : 1));
  o->R.assign(s->R, s->R + (size_t)m * k * (o->Rtv ? n : 1));
  o->a1.assign(s->a1, s->a1 + m);
  o->P1.assign(s->P1, s->P1 + (size_t)m * m);
  if (s->D) o->D.assign(s->D, s->D + (size_t)p * (o->Dtv ? n : 1)); else { o->D.assign(p, 0.0); o->Dtv = 0; }
  if (s->C) o->C.assign(s->C, s->C + (size_t)m * (o->Ctv ? n : 1)); else { o->C.assign(m, 0.0); o->Ctv = 0; }
  o->xbeta.assign(n, 0.0);
  if (s->n_xreg > 0) {  // xbeta = xreg * beta (src/model_ssm_ung.h:84); univariate models only
    if (p > 1) return "regression part is not available for multivariate models";
    if (!s->xreg || !s->beta) return "xreg/beta NULL with n_xreg > 0";
    for (int q = 0; q < s->n_xreg; q++)
      for (int t = 0; t < n; t++) o->xbeta[t] += s->xreg[t + (size_t)n * q] * s->beta[q];
  }
  if (o->gaussian()) {
    if (!s->H) return "gaussian model needs H";
    o->H.assign(s->H, s->H + (o->Htv ? n : 1));
  }
  if (need_mode) {
    if (o->gaussian()) return "psi-APF of a gaussian model is exact simulation; not routed through this entry";
    if (!s->initial_mode) return "initial_mode is required for the psi-APF";
    o->initial_mode.resize((size_t)p * n);
    for (int t = 0; t < n; t++) for (int j = 0; j < p; j++) o->initial_mode[j + (size_t)p * t] = s->initial_mode[t + (size_t)n * j];
  }
  for (int j = 0; j < p; j++)
    if ((o->dist[j] == BSSM_NEGBIN || o->dist[j] == BSSM_GAMMA || o->dist[j] == BSSM_SVM || (p > 1 && o->dist[j] == BSSM_GAUSSIAN)) &&
        !(o->phis[j] > 0))
      return "phi must be positive";
  return "";
}

double const_term(const HostModel& h) {
  double res = 0.0;
  const int p = h.p;
  int nfin = 0;
  for (int j = 0; j < p; j++)
    for (int t = 0; t < h.n; t++) {
      const size_t q = j + (size_t)p * t;
      const double y = h.y[q];
      if (!std::isfinite(y)) continue;
      nfin++;
      const double u = h.u[q], phi = h.phis[j];
      switch (h.dist[j]) {
        case BSSM_POISSON: res += y * std::log(u) - std::lgamma(y + 1); break;
        case BSSM_BINOMIAL: res += lchoose_ts(u, y); break;
        case BSSM_NEGBIN: res += lchoose_ts(y + phi - 1, y) + phi * std::log(phi) + y * std::log(u); break;
        case BSSM_GAMMA: res += phi * std::log(phi) - std::lgamma(phi) + (phi - 1) * std::log(y) - phi * std::log(u); break;
        case BSSM_GAUSSIAN: res += norm_log_const(h.gaussian() ? h.H[t * h.Htv] : phi); break;
        default: break;
      }
    }
  if (p == 1 && h.dist[0] == BSSM_SVM) res = nfin * norm_log_const(h.phi);
  return res;
}

DeviceFactorsFn& device_factors_hook() { static DeviceFactorsFn fn = nullptr; return fn; }

bool psi_setup(const HostModel& h, const double* is_mode, PsiSetup* out, const GlobalApprox* global) {
  if (is_mode && h.p > 1 && bssm::opt().mv_sequential) {
    // test hook: the series-at-a-time smoother the device batch runs for multivariate observations, on the host
    LgPacked lg;
    if (psi_setup_begin(h, is_mode, out, &lg)) {
      const bool ok = lg_packed_run(lg, out->alphahat.data(), out->Vt.data(), out->Ct.data());
      psi_setup_compact(lg, out);
      return ok;
    }
  }
  const int n = h.n, m = h.m, p = h.p;
  const Sys S(h);
  Approx ap;
  ap.y.assign((size_t)p * n, 0.0);
  ap.H.assign((size_t)p * n, 0.0);
  bool ok = true;
  if (is_mode) {  // approximate_for_is: src/model_ssm_ung.cpp:143-161, src/model_ssm_mng.cpp:117-133 (mode p x n, as mode_storage)
    out->mode.assign(is_mode, is_mode + (size_t)p * n);
    laplace_iter(h, out->mode, &ap);
    update_scales(h, ap, out->mode, &out->scales);
    out->approx_loglik = 0.0;
    out->iterations = 0;
  } else {  // psi_filter set-up: src/model_ssm_ung.cpp:514-528
    if (global && global->set) {  // local_approx = FALSE after the first approximation: src/model_ssm_ung.cpp:104-116
      ap.y = global->y; ap.H = global->H;
      std::vector<Mat> sm;
      ok = fast_smoother(S, ap, &sm);
      if (ok) signal_from_states(h, S, sm, &out->mode); else out->mode = h.initial_mode;
      out->iterations = 0;
    } else {
      const auto tl0 = std::chrono::steady_clock::now();
      out->iterations = run_laplace(h, S, &ap, &out->mode, &ok);
      if (bssm::opt().trace) std::fprintf(stderr, "[bssm_amd] psi set-up of one model (m = %d, n = %d): mode search %d iterations %.1f ms\n", m, n, out->iterations,
                                          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl0).count());
    }
    const auto tg0 = std::chrono::steady_clock::now();
    const double gll = gaussian_loglik(S, ap);
    if (bssm::opt().trace) std::fprintf(stderr, "[bssm_amd]   approximating log-likelihood %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tg0).count());
    update_scales(h, ap, out->mode, &out->scales);
    double ssc = 0;
    for (double v : out->scales) ssc += v;
    out->approx_loglik = gll + approx_const_term(h, ap) + ssc;
  }
  out->approx_y = ap.y;
  out->approx_H = ap.H;
  if (ok && p == 1 && m >= 9 && device_factors_hook()) {   // (smoother + conditional factors of 9 ... 64 states on the device: host_model.h)
    bool fok = true;
    if (device_factors_hook()(h, ap.y, ap.H, out, &fok)) return fok;
  }
  {
    bool fok = true;
    if (ok && lg_fast_path(h, ap, &out->alphahat, &out->Vt, &out->Ct, &fok)) return fok;
  }
  std::vector<Mat> at, Vt, Ct;
  const auto ts0 = std::chrono::steady_clock::now();
  ok = ok && smoother_ccov(S, ap, &at, &Vt, &Ct);  // :533
  const auto ts1 = std::chrono::steady_clock::now();
  out->alphahat.assign((size_t)m * (n + 1), 0.0);
  out->Vt.assign((size_t)m * m * (n + 1), 0.0);
  out->Ct.assign((size_t)m * m * (n + 1), 0.0);
  if (!ok) return false;
  conditional_cov(&Vt, &Ct);                       // :534
  if (bssm::opt().trace) std::fprintf(stderr, "[bssm_amd]   smoother %.1f ms, conditional factors %.1f ms\n", std::chrono::duration<double, std::milli>(ts1 - ts0).count(),
                                      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts1).count());
  for (int t = 0; t <= n; t++) {
    for (int i = 0; i < m; i++) { out->alphahat[i + (size_t)m * t] = at[t](i, 0); ok &= std::isfinite(at[t](i, 0)); }
    for (int q = 0; q < m * m; q++) {
      out->Vt[q + (size_t)m * m * t] = Vt[t].a[q];
      out->Ct[q + (size_t)m * m * t] = Ct[t].a[q];
      ok &= std::isfinite(Vt[t].a[q]) && std::isfinite(Ct[t].a[q]);
    }
  }
  return ok;
}

namespace {
void lg_pack(const HostModel& h, const Approx& ap, LgPacked* lg) {
  const LgHostModel lm(h, ap);
  const int n = h.n, m = h.m;
  lg->n = n; lg->m = m;
  lg->Zs = lm.g.Zs; lg->Ts = lm.g.Ts; lg->Rs = lm.g.Rs; lg->Ds = lm.g.Ds; lg->Cs = lm.g.Cs; lg->Hs = 1;
  const size_t nZ = (size_t)(lg->Zs ? n : 1) * m, nT = (size_t)(lg->Ts ? n : 1) * m * m, nR = (size_t)(lg->Rs ? n : 1) * m * m;
  const size_t nD = lg->Ds ? n : 1, nC = (size_t)(lg->Cs ? n : 1) * m;
  size_t o = 0;
  lg->oZ = o; o += nZ; lg->oT = o; o += nT; lg->oR = o; o += nR; lg->oD = o; o += nD; lg->oC = o; o += nC;
  lg->oy = o; o += n; lg->oH = o; o += n; lg->oa = o; o += m; lg->oP = o; o += (size_t)m * m;
  lg->buf.resize(o);
  double* b = lg->buf.data();
  std::memcpy(b + lg->oZ, lm.g.Z, nZ * sizeof(double));
  std::memcpy(b + lg->oT, lm.g.T, nT * sizeof(double));
  std::memcpy(b + lg->oR, lm.g.RR, nR * sizeof(double));
  std::memcpy(b + lg->oD, lm.g.D, nD * sizeof(double));
  std::memcpy(b + lg->oC, lm.g.C, nC * sizeof(double));
  std::memcpy(b + lg->oy, lm.g.y, (size_t)n * sizeof(double));
  std::memcpy(b + lg->oH, lm.g.H, (size_t)n * sizeof(double));
  std::memcpy(b + lg->oa, lm.g.a1, (size_t)m * sizeof(double));
  std::memcpy(b + lg->oP, lm.g.P1, (size_t)m * m * sizeof(double));
}
}  // namespace

namespace {
// ssm_mng's approximating model (independent series) packed for the series-at-a-time smoother: Z rows per (time point,
// series), D / y / H per (time point, series), T / R R' / C per time point
bool lg_pack_mv(const HostModel& h, const Approx& ap, LgPacked* lg) {
  const int n = h.n, m = h.m, k = h.k, p = h.p;
  lg->n = n * p; lg->m = m; lg->sub = p;
  lg->Zs = h.Ztv ? p * m : 0; lg->Ts = h.Ttv ? m * m : 0; lg->Rs = h.Rtv ? m * m : 0; lg->Ds = h.Dtv ? p : 0; lg->Cs = h.Ctv ? m : 0; lg->Hs = 1;
  const int nZt = h.Ztv ? n : 1, nR = h.Rtv ? n : 1;
  const size_t nZ = (size_t)nZt * p * m, nT = (size_t)(h.Ttv ? n : 1) * m * m, nRR = (size_t)nR * m * m;
  const size_t nD = (size_t)(h.Dtv ? n : 1) * p, nC = (size_t)(h.Ctv ? n : 1) * m, nv = (size_t)n * p;
  size_t o = 0;
  lg->oZ = o; o += nZ; lg->oT = o; o += nT; lg->oR = o; o += nRR; lg->oD = o; o += nD; lg->oC = o; o += nC;
  lg->oy = o; o += nv; lg->oH = o; o += nv; lg->oa = o; o += m; lg->oP = o; o += (size_t)m * m;
  lg->buf.resize(o);
  double* b = lg->buf.data();
  for (int t = 0; t < nZt; t++)  // h.Z is p x m column-major per time point: one contiguous row per series here
    for (int j = 0; j < p; j++) for (int c = 0; c < m; c++) b[lg->oZ + ((size_t)t * p + j) * m + c] = h.Z[j + (size_t)p * c + (size_t)p * m * t];
  std::memcpy(b + lg->oT, h.T.data(), nT * sizeof(double));
  for (int t = 0; t < nR; t++) {
    const double* R = h.R.data() + (size_t)m * k * t;
    double* RR = b + lg->oR + (size_t)m * m * t;
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) {
      double sacc = 0.0;
      for (int q = 0; q < k; q++) sacc += R[i + m * q] * R[j + m * q];
      RR[i + m * j] = sacc;
    }
  }
  std::memcpy(b + lg->oD, h.D.data(), nD * sizeof(double));
  std::memcpy(b + lg->oC, h.C.data(), nC * sizeof(double));
  std::memcpy(b + lg->oy, ap.y.data(), nv * sizeof(double));
  std::memcpy(b + lg->oH, ap.H.data(), nv * sizeof(double));
  std::memcpy(b + lg->oa, h.a1.data(), (size_t)m * sizeof(double));
  std::memcpy(b + lg->oP, h.P1.data(), (size_t)m * m * sizeof(double));
  return true;
}
}  // namespace

// the smoother ran over n * sub virtual steps: keep the entries of the time points (index s * sub)
void psi_setup_compact(const LgPacked& lg, PsiSetup* su) {
  if (lg.sub <= 1) return;
  const int m = lg.m, sub = lg.sub, n = lg.n / sub;
  const size_t mm = (size_t)m * m;
  for (int s = 1; s <= n; s++) {
    std::memmove(su->alphahat.data() + (size_t)m * s, su->alphahat.data() + (size_t)m * s * sub, (size_t)m * sizeof(double));
    std::memmove(su->Vt.data() + mm * s, su->Vt.data() + mm * s * sub, mm * sizeof(double));
    std::memmove(su->Ct.data() + mm * s, su->Ct.data() + mm * s * sub, mm * sizeof(double));
  }
  su->alphahat.resize((size_t)m * (n + 1));
  su->Vt.resize(mm * (n + 1));
  su->Ct.resize(mm * (n + 1));
}

bool psi_setup_begin(const HostModel& h, const double* is_mode, PsiSetup* out, LgPacked* lg) {
  const bool mv = h.p > 1 && h.m >= 1 && h.m <= 8 && !bssm::opt().generic_setup;
  if (!is_mode || !(mv || LgHostModel::applies(h))) return false;
  const int n = h.n, m = h.m, p = h.p;
  if (mv) {
    Approx ap;
    ap.y.assign((size_t)p * n, 0.0);
    ap.H.assign((size_t)p * n, 0.0);
    out->mode.assign(is_mode, is_mode + (size_t)p * n);  // approximate_for_is: src/model_ssm_mng.cpp:117-133
    laplace_iter(h, out->mode, &ap);
    update_scales(h, ap, out->mode, &out->scales);
    out->approx_loglik = 0.0;
    out->iterations = 0;
    out->approx_y = ap.y;
    out->approx_H = ap.H;
    const size_t nv1 = (size_t)n * p + 1;  // virtual steps + 1 until psi_setup_compact
    out->alphahat.assign((size_t)m * nv1, 0.0);
    out->Vt.assign((size_t)m * m * nv1, 0.0);
    out->Ct.assign((size_t)m * m * nv1, 0.0);
    return lg_pack_mv(h, ap, lg);
  }
  Approx ap;
  ap.y.assign((size_t)p * n, 0.0);
  ap.H.assign((size_t)p * n, 0.0);
  out->mode.assign(is_mode, is_mode + (size_t)p * n);  // approximate_for_is: src/model_ssm_ung.cpp:143-161
  laplace_iter(h, out->mode, &ap);
  update_scales(h, ap, out->mode, &out->scales);
  out->approx_loglik = 0.0;
  out->iterations = 0;
  out->approx_y = ap.y;
  out->approx_H = ap.H;
  out->alphahat.assign((size_t)m * (n + 1), 0.0);
  out->Vt.assign((size_t)m * m * (n + 1), 0.0);
  out->Ct.assign((size_t)m * m * (n + 1), 0.0);
  lg_pack(h, ap, lg);
  return true;
}

bool ung_pack(const HostModel& h, const double* is_mode, LgPacked* lg) {
  // states of 9 ... 16 dimensions (a monthly seasonal: m = 13) are packed PADDED to 16 (round 5): zero rows / columns of Z, T, R R', C, a1,
  // P1 -- the padded components stay exactly zero through the filter and the smoother, the products only gain exact zero terms, and the
  // Jacobi rotations of the factors skip zero off-diagonals -- so that ONE device instantiation (ung_setup_kernel<16>) carries them
  // (17 ... 64 states: padded to 32 / 64 for the LDS-resident smoother, lg_smoother_wave.h -- late round 5)
  if (!is_mode || h.gaussian() || h.p != 1 || h.m < 1 || h.m > 64 || bssm::opt().generic_setup) return false;
  const int n = h.n, mr = h.m, m = mr <= 8 ? mr : (mr <= 16 ? 16 : (mr <= 32 ? 32 : 64)), k = h.k;
  lg->n = n; lg->m = m;
  lg->Zs = h.Ztv ? m : 0; lg->Ts = h.Ttv ? m * m : 0; lg->Rs = h.Rtv ? m * m : 0; lg->Ds = h.Dtv ? 1 : 0; lg->Cs = h.Ctv ? m : 0; lg->Hs = 1;
  const int nR = h.Rtv ? n : 1, nZt = h.Ztv ? n : 1, nTt = h.Ttv ? n : 1, nCt = h.Ctv ? n : 1;
  const size_t nZ = (size_t)nZt * m, nT = (size_t)nTt * m * m, nRR = (size_t)nR * m * m;
  const size_t nD = h.Dtv ? n : 1, nC = (size_t)nCt * m;
  size_t o = 0;
  lg->oZ = o; o += nZ; lg->oT = o; o += nT; lg->oR = o; o += nRR; lg->oD = o; o += nD; lg->oC = o; o += nC;
  lg->oy = o; o += n; lg->oH = lg->oy; lg->oa = o; o += m; lg->oP = o; o += (size_t)m * m;
  lg->ou = o; o += n; lg->ox = o; o += n; lg->om = o; o += n; lg->oh = o; o += 2;
  lg->buf.assign(o, 0.0);
  double* b = lg->buf.data();
  for (int t = 0; t < nZt; t++) for (int c = 0; c < mr; c++) b[lg->oZ + (size_t)t * m + c] = h.Z[(size_t)t * mr + c];
  for (int t = 0; t < nTt; t++) for (int j = 0; j < mr; j++) for (int i = 0; i < mr; i++) b[lg->oT + (size_t)t * m * m + i + m * j] = h.T[(size_t)t * mr * mr + i + mr * j];
  for (int t = 0; t < nR; t++) {  // R R' in LgHostModel's operation order
    const double* R = h.R.data() + (size_t)mr * k * t;
    double* RR = b + lg->oR + (size_t)m * m * t;
    for (int j = 0; j < mr; j++) for (int i = 0; i < mr; i++) {
      double sacc = 0.0;
      for (int q = 0; q < k; q++) sacc += R[i + mr * q] * R[j + mr * q];
      RR[i + m * j] = sacc;
    }
  }
  std::memcpy(b + lg->oD, h.D.data(), nD * sizeof(double));
  for (int t = 0; t < nCt; t++) for (int c = 0; c < mr; c++) b[lg->oC + (size_t)t * m + c] = h.C[(size_t)t * mr + c];
  std::memcpy(b + lg->oy, h.y.data(), (size_t)n * sizeof(double));
  for (int c = 0; c < mr; c++) b[lg->oa + c] = h.a1[c];
  for (int j = 0; j < mr; j++) for (int i = 0; i < mr; i++) b[lg->oP + i + m * j] = h.P1[i + mr * j];
  std::memcpy(b + lg->ou, h.u.data(), (size_t)n * sizeof(double));
  std::memcpy(b + lg->ox, h.xbeta.data(), (size_t)n * sizeof(double));
  std::memcpy(b + lg->om, is_mode, (size_t)n * sizeof(double));
  b[lg->oh] = (double)h.dist[0];
  b[lg->oh + 1] = h.phis[0];
  return true;
}

// ssm_mng draw for mng_setup_kernel: [Z rows per (time point, series) | T | R R' | D | C | y (raw, p per time point) |
// a1 | P1 | u | mode | distributions (p), phis (p)]; lg->n = time points, lg->sub = p
bool mng_pack(const HostModel& h, const double* is_mode, LgPacked* lg) {
  // (round 6: 9 ... 64 states go PADDED to 16 / 32 / 64 as ung_pack does -- zero rows / columns of Z, T, R R', C, a1, P1; the padded
  // components stay exactly zero through the filter and the smoother)
  if (!is_mode || h.p <= 1 || h.m < 1 || h.m > 64 || h.gaussian() || bssm::opt().generic_setup) return false;
  const int n = h.n, mr = h.m, m = mr <= 8 ? mr : (mr <= 16 ? 16 : (mr <= 32 ? 32 : 64)), k = h.k, p = h.p;
  lg->n = n; lg->m = m; lg->sub = p;
  lg->Zs = h.Ztv ? p * m : 0; lg->Ts = h.Ttv ? m * m : 0; lg->Rs = h.Rtv ? m * m : 0; lg->Ds = h.Dtv ? p : 0; lg->Cs = h.Ctv ? m : 0; lg->Hs = 1;
  const int nZt = h.Ztv ? n : 1, nR = h.Rtv ? n : 1, nTt = h.Ttv ? n : 1, nCt = h.Ctv ? n : 1;
  const size_t nZ = (size_t)nZt * p * m, nT = (size_t)nTt * m * m, nRR = (size_t)nR * m * m;
  const size_t nD = (size_t)(h.Dtv ? n : 1) * p, nC = (size_t)nCt * m, nv = (size_t)n * p;
  size_t o = 0;
  lg->oZ = o; o += nZ; lg->oT = o; o += nT; lg->oR = o; o += nRR; lg->oD = o; o += nD; lg->oC = o; o += nC;
  lg->oy = o; o += nv; lg->oH = lg->oy; lg->oa = o; o += m; lg->oP = o; o += (size_t)m * m;
  lg->ou = o; o += nv; lg->ox = 0; lg->om = o; o += nv; lg->oh = o; o += 2 * (size_t)p;
  lg->buf.assign(o, 0.0);
  double* b = lg->buf.data();
  for (int t = 0; t < nZt; t++)  // h.Z is p x m column-major per time point: one contiguous (padded) row per series here
    for (int j = 0; j < p; j++) for (int c = 0; c < mr; c++) b[lg->oZ + ((size_t)t * p + j) * m + c] = h.Z[j + (size_t)p * c + (size_t)p * mr * t];
  for (int t = 0; t < nTt; t++) for (int j = 0; j < mr; j++) for (int i = 0; i < mr; i++) b[lg->oT + (size_t)t * m * m + i + m * j] = h.T[(size_t)t * mr * mr + i + mr * j];
  for (int t = 0; t < nR; t++) {  // R R' in lg_pack_mv's operation order
    const double* R = h.R.data() + (size_t)mr * k * t;
    double* RR = b + lg->oR + (size_t)m * m * t;
    for (int j = 0; j < mr; j++) for (int i = 0; i < mr; i++) {
      double sacc = 0.0;
      for (int q = 0; q < k; q++) sacc += R[i + mr * q] * R[j + mr * q];
      RR[i + m * j] = sacc;
    }
  }
  std::memcpy(b + lg->oD, h.D.data(), nD * sizeof(double));
  for (int t = 0; t < nCt; t++) for (int c = 0; c < mr; c++) b[lg->oC + (size_t)t * m + c] = h.C[(size_t)t * mr + c];
  std::memcpy(b + lg->oy, h.y.data(), nv * sizeof(double));
  for (int c = 0; c < mr; c++) b[lg->oa + c] = h.a1[c];
  for (int j = 0; j < mr; j++) for (int i = 0; i < mr; i++) b[lg->oP + i + m * j] = h.P1[i + mr * j];
  std::memcpy(b + lg->ou, h.u.data(), nv * sizeof(double));
  std::memcpy(b + lg->om, is_mode, nv * sizeof(double));
  for (int j = 0; j < p; j++) { b[lg->oh + j] = (double)h.dist[j]; b[lg->oh + p + j] = h.phis[j]; }
  return true;
}

bool lg_packed_run(const LgPacked& lg, double* alphahat, double* Vt, double* Ct) {
  const double* b = lg.buf.data();
  LgModel g;
  g.n = lg.n; g.sub = lg.sub;
  g.Z = b + lg.oZ; g.Zs = lg.Zs; g.T = b + lg.oT; g.Ts = lg.Ts; g.RR = b + lg.oR; g.Rs = lg.Rs;
  g.D = b + lg.oD; g.Ds = lg.Ds; g.C = b + lg.oC; g.Cs = lg.Cs; g.y = b + lg.oy; g.H = b + lg.oH; g.Hs = lg.Hs;
  g.a1 = b + lg.oa; g.P1 = b + lg.oP;
  std::vector<double> K((size_t)lg.n * lg.m), v(lg.n), Fs(lg.n);
  return lg_smoother_run(lg.m, g, alphahat, Vt, Ct, K.data(), v.data(), Fs.data());
}

// smoothed means and covariances of a linear-Gaussian model (ssm_ulg::smoother, src/model_ssm_ulg.cpp:307-417)
bool gaussian_smoother(const HostModel& h, std::vector<double>* alphahat, std::vector<double>* Vt_out) {
  if (!h.gaussian()) return false;
  const int n = h.n, m = h.m;
  const Sys S(h);
  Approx ap;
  ap.y = h.y;
  ap.H.resize(n);
  for (int t = 0; t < n; t++) ap.H[t] = h.H[t * h.Htv];
  std::vector<Mat> at, Vt, Ct;
  if (!smoother_ccov(S, ap, &at, &Vt, &Ct)) return false;
  alphahat->assign((size_t)m * (n + 1), 0.0);
  Vt_out->assign((size_t)m * m * (n + 1), 0.0);
  bool ok = true;
  for (int t = 0; t <= n; t++) {
    for (int i = 0; i < m; i++) { (*alphahat)[i + (size_t)m * t] = at[t](i, 0); ok &= std::isfinite(at[t](i, 0)); }
    for (int q = 0; q < m * m; q++) { (*Vt_out)[q + (size_t)m * m * t] = Vt[t].a[q]; ok &= std::isfinite(Vt[t].a[q]); }
  }
  return ok;
}

bool gaussian_smoothing_setup(const HostModel& h, PsiSetup* out) {
  if (!h.gaussian()) return false;
  const int n = h.n, m = h.m;
  const Sys S(h);
  Approx ap;
  ap.y = h.y;
  ap.H.resize(n);
  for (int t = 0; t < n; t++) ap.H[t] = h.H[t * h.Htv];
  out->scales.assign(n, 0.0);
  out->approx_y = ap.y; out->approx_H = ap.H; out->mode.assign(n, 0.0);
  {
    bool fok = true;
    if (lg_fast_path(h, ap, &out->alphahat, &out->Vt, &out->Ct, &fok)) return fok;
  }
  std::vector<Mat> at, Vt, Ct;
  if (!smoother_ccov(S, ap, &at, &Vt, &Ct)) return false;
  conditional_cov(&Vt, &Ct);
  out->alphahat.assign((size_t)m * (n + 1), 0.0);
  out->Vt.assign((size_t)m * m * (n + 1), 0.0);
  out->Ct.assign((size_t)m * m * (n + 1), 0.0);
  bool ok = true;
  for (int t = 0; t <= n; t++) {
    for (int i = 0; i < m; i++) { out->alphahat[i + (size_t)m * t] = at[t](i, 0); ok &= std::isfinite(at[t](i, 0)); }
    for (int q = 0; q < m * m; q++) { out->Vt[q + (size_t)m * m * t] = Vt[t].a[q]; out->Ct[q + (size_t)m * m * t] = Ct[t].a[q]; ok &= std::isfinite(Vt[t].a[q]) && std::isfinite(Ct[t].a[q]); }
  }
  return ok;
}

// ssm_mlg (gaussian_psi_smoother case 0, src/R_psi.cpp:20-25): smoother_ccov (src/model_ssm_mlg.cpp:310-396) +
// conditional_cov on the model itself.  `s` carries y n x p, Z p x m x (1|n), D p x (1|n), H p x p x (1|n) lower factor.
std::string mlg_smoothing_setup(const bssm_ung_model* s, PsiSetup* out) {
  if (!s) return "model is NULL";
  if (s->n < 1 || s->m < 1 || s->k < 1 || s->p < 1) return "n, m, k, p must be positive";
  if (s->m > 64 || s->p > 16) return "ssm_mlg: m <= 64 and p <= 16 are supported";
  if (!s->y || !s->Z || !s->T || !s->R || !s->a1 || !s->P1 || !s->H) return "y, Z, H, T, R, a1, P1 must be non-NULL";
  HostModel h;
  const int n = s->n, m = s->m, k = s->k, p = s->p;
  h.n = n; h.m = m; h.k = k; h.p = p;
  h.Ztv = s->Ztv != 0; h.Ttv = s->Ttv != 0; h.Rtv = s->Rtv != 0; h.Dtv = s->Dtv != 0; h.Ctv = s->Ctv != 0; h.Htv = s->Htv != 0;
  h.dist.assign(p, BSSM_GAUSSIAN); h.phis.assign(p, 1.0);
  h.y.resize((size_t)p * n);
  for (int t = 0; t < n; t++) for (int j = 0; j < p; j++) h.y[j + (size_t)p * t] = s->y[t + (size_t)n * j];  // y.t(), :15
  h.Z.assign(s->Z, s->Z + (size_t)p * m * (h.Ztv ? n : 1));
  h.T.assign(s->T, s->T + (size_t)m * m * (h.Ttv ? n : 1));
  h.R.assign(s->R, s->R + (size_t)m * k * (h.Rtv ? n : 1));
  h.a1.assign(s->a1, s->a1 + m);
  h.P1.assign(s->P1, s->P1 + (size_t)m * m);
  if (s->D) h.D.assign(s->D, s->D + (size_t)p * (h.Dtv ? n : 1)); else { h.D.assign(p, 0.0); h.Dtv = 0; }
  if (s->C) h.C.assign(s->C, s->C + (size_t)m * (h.Ctv ? n : 1)); else { h.C.assign(m, 0.0); h.Ctv = 0; }
  h.xbeta.assign(n, 0.0);
  Approx ap;
  ap.y = h.y;
  ap.HHtv = h.Htv;
  const int ns = h.Htv ? n : 1;
  ap.HHfull.assign((size_t)p * p * ns, 0.0);
  for (int q = 0; q < ns; q++) {
    const Mat Hm(p, p, s->H + (size_t)p * p * q);
    const Mat HHm = Hm * tr(Hm);
    std::copy(HHm.a, HHm.a + p * p, ap.HHfull.begin() + (size_t)p * p * q);
  }
  const Sys S(h);
  std::vector<Mat> at, Vt, Ct;
  if (!smoother_ccov(S, ap, &at, &Vt, &Ct)) return "smoother produced non-finite values";
  conditional_cov(&Vt, &Ct);
  out->alphahat.assign((size_t)m * (n + 1), 0.0);
  out->Vt.assign((size_t)m * m * (n + 1), 0.0);
  out->Ct.assign((size_t)m * m * (n + 1), 0.0);
  bool ok = true;
  for (int t = 0; t <= n; t++) {
    for (int i = 0; i < m; i++) { out->alphahat[i + (size_t)m * t] = at[t](i, 0); ok &= std::isfinite(at[t](i, 0)); }
    for (int q = 0; q < m * m; q++) { out->Vt[q + (size_t)m * m * t] = Vt[t].a[q]; out->Ct[q + (size_t)m * m * t] = Ct[t].a[q]; ok &= std::isfinite(Vt[t].a[q]) && std::isfinite(Ct[t].a[q]); }
  }
  return ok ? "" : "smoother produced non-finite values";
}

double approx_loglik_only(const HostModel& h) { return approx_loglik_mode(h, nullptr, nullptr); }

double approx_loglik_mode(const HostModel& h, std::vector<double>* mode_out, GlobalApprox* global) {
  const Sys S(h);
  Approx ap;
  ap.y.assign((size_t)h.p * h.n, 0.0);
  ap.H.assign((size_t)h.p * h.n, 0.0);
  std::vector<double> mode, scales;
  bool ok = true;
  double gll_pass = std::nan("");
  if (global && global->set) {  // the pseudo-observations of the first approximation are kept (src/model_ssm_ung.cpp:104-116)
    ap.y = global->y; ap.H = global->H;
    std::vector<Mat> sm;
    ok = fast_smoother(S, ap, &sm);
    if (ok) signal_from_states(h, S, sm, &mode);
  } else {
    run_laplace(h, S, &ap, &mode, &ok, &gll_pass);
    if (global && ok) { global->y = ap.y; global->H = ap.H; global->set = true; }
  }
  if (!ok) return -INFINITY;
  if (mode_out) *mode_out = mode;
  const double gll = std::isnan(gll_pass) ? gaussian_loglik(S, ap) : gll_pass;
  update_scales(h, ap, mode, &scales);
  double ssc = 0;
  for (double v : scales) ssc += v;
  return gll + approx_const_term(h, ap) + ssc;
}

namespace {
void fill_obs_rows(const HostModel& h, int MP, const PsiSetup* su, std::vector<double>* obs) {
  const int n = h.n, m = h.m, p = h.p, sstride = SER_HDR + MP, stride = ROW_HDR + p * sstride;
  obs->assign((size_t)n * stride, 0.0);
  for (int t = 0; t < n; t++) {
    double* row = obs->data() + (size_t)t * stride;
    bool any = false;
    for (int j = 0; j < p; j++) {
      double* ob = row + ROW_HDR + j * sstride;
      const size_t q = j + (size_t)p * t;
      const double y = h.y[q], u = h.u[q], phi = h.phis[j];
      const int dist = h.dist[j];
      ob[OB_DIST] = (double)dist;
      ob[OB_PHI] = phi;
      if (!std::isfinite(y)) continue;  // flag stays 0: the series is skipped (src/model_ssm_mng.cpp:414)
      any = true;
      ob[OB_FLAG] = 1.0;
      ob[OB_Y] = y;
      ob[OB_U] = u;
      const double* Zt = h.Z.data() + (size_t)p * m * (t * h.Ztv);
      switch (dist) {
        case BSSM_SVM:  // signal is the first state; Z, D, xbeta are ignored (src/model_ssm_ung.cpp:456)
          ob[OB_D] = 0.0;
          ob[SER_HDR] = 1.0;
          ob[OB_AUX] = (y / phi) * (y / phi);
          break;
        case BSSM_GAUSSIAN:
          // ssm_ulg: mu = D + Z'alpha (no xbeta, src/model_ssm_ulg.cpp:583-584), variance H_t^2;
          // ssm_mng series: -0.5 ((y - s) / phi)^2 (src/model_ssm_mng.cpp:435-437)
          ob[OB_D] = h.D[j + (size_t)p * (t * h.Dtv)];
          for (int c = 0; c < m; c++) ob[SER_HDR + c] = Zt[j + p * c];
          ob[OB_AUX] = h.gaussian() ? 1.0 / (h.H[t * h.Htv] * h.H[t * h.Htv]) : 1.0 / (phi * phi);
          break;
        default:
          ob[OB_D] = h.D[j + (size_t)p * (t * h.Dtv)] + (p == 1 ? h.xbeta[t] : 0.0);
          for (int c = 0; c < m; c++) ob[SER_HDR + c] = Zt[j + p * c];
          ob[OB_AUX] = dist == BSSM_NEGBIN ? y + phi : (dist == BSSM_GAMMA ? y / u : 0.0);
      }
      if (su) {
        ob[OB_PSIY] = su->approx_y[q];
        ob[OB_PSIIH] = 1.0 / su->approx_H[q];
      }
    }
    row[RH_FLAG] = any ? 1.0 : 0.0;
    row[RH_SCALE] = (su && any) ? su->scales[t] : 0.0;
  }
}
}  // namespace

void build_bsf_tables(const HostModel& h, int MP, int KP, FilterTables* out) {
  const int n = h.n, m = h.m, k = h.k;
  const int MS = MP * MP + MP * KP + MP;
  const bool tv = h.Ttv || h.Rtv || h.Ctv;
  out->MP = MP; out->KP = KP;
  out->mats_tstride = tv ? MS : 0;
  const int rows = tv ? n : 1;
  out->mats.assign((size_t)rows * MS, 0.0);
  for (int t = 0; t < rows; t++) {
    double* mt = out->mats.data() + (size_t)t * MS;
    const double* T = h.T.data() + (size_t)m * m * (t * h.Ttv);
    const double* R = h.R.data() + (size_t)m * k * (t * h.Rtv);
    const double* C = h.C.data() + (size_t)m * (t * h.Ctv);
    for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) mt[j + MP * c] = T[j + m * c];
    for (int c = 0; c < k; c++) for (int j = 0; j < m; j++) mt[MP * MP + j + MP * c] = R[j + m * c];
    for (int j = 0; j < m; j++) mt[MP * MP + MP * KP + j] = C[j];
  }
  fill_obs_rows(h, MP, nullptr, &out->obs);
  // initial factor: ung chol_lower(P1[nz,nz]) with nz = diag > 0 (src/model_ssm_ung.cpp:614-619);
  // gaussian: psd_chol threshold (src/psd_chol.cpp:7-15)
  const Mat P1(m, m, h.P1.data());
  std::vector<int> nz;
  double dmax = P1(0, 0);
  for (int j = 1; j < m; j++) dmax = std::max(dmax, P1(j, j));
  const double thr = h.gaussian() ? std::max(kEps, kEps * m * dmax) : 0.0;
  for (int j = 0; j < m; j++) if (P1(j, j) > thr) nz.push_back(j);
  Mat L(m, m);
  if (!nz.empty()) chol_lower_on(P1, nz, &L);
  out->init.assign((size_t)MP + (size_t)MP * MP, 0.0);
  for (int j = 0; j < m; j++) out->init[j] = h.a1[j];
  for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) out->init[MP + j + MP * c] = L(j, c);
}

void build_psi_tables(const HostModel& h, const PsiSetup& su, int MP, FilterTables* out) {
  const int n = h.n, m = h.m, KP = MP;
  const int MS = MP * MP + MP * KP + MP;
  out->MP = MP; out->KP = KP;
  out->mats_tstride = MS;
  out->mats.assign((size_t)n * MS, 0.0);
  for (int t = 0; t < n; t++) {  // src/model_ssm_ung.cpp:585-586
    double* mt = out->mats.data() + (size_t)t * MS;
    const double* Ct1 = su.Ct.data() + (size_t)m * m * (t + 1);
    const double* Vt1 = su.Vt.data() + (size_t)m * m * (t + 1);
    const double* ah0 = su.alphahat.data() + (size_t)m * t;
    const double* ah1 = su.alphahat.data() + (size_t)m * (t + 1);
    for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) {
      mt[j + MP * c] = Ct1[j + m * c];
      mt[MP * MP + j + MP * c] = Vt1[j + m * c];
    }
    for (int j = 0; j < m; j++) {
      double s = 0.0;
      for (int c = 0; c < m; c++) s += Ct1[j + m * c] * ah0[c];
      mt[MP * MP + MP * KP + j] = ah1[j] - s;
    }
  }
  fill_obs_rows(h, MP, &su, &out->obs);
  out->init.assign((size_t)MP + (size_t)MP * MP, 0.0);  // alphahat_0 + V_0 u  (:543)
  for (int j = 0; j < m; j++) out->init[j] = su.alphahat[j];
  for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) out->init[MP + j + MP * c] = su.Vt[j + m * c];
}

// ------------------------------------------------------------------ built-in ssm_nlg functors (host side)
namespace {
// user models: up to 16 states in the psi-APF set-up since round 6 (8 before; the EKF particle filter keeps 8: its per-particle covariance is a register array)
constexpr int kFnM = 16, kFnMM = kFnM * kFnM;
double fnH(const FnModel& f) { if (f.id == 7) return f.uH; return f.id == 1 ? std::exp(f.theta[0]) : (f.id == 2 ? std::exp(f.theta[3]) : std::exp(f.theta[0])); }
void fnRdiag(const FnModel& f, double* r) {
  for (int j = 0; j < kFnM; j++) r[j] = 0;  // (callers hold kFnM entries)
  if (f.id == 7) { for (int j = 0; j < f.m; j++) r[j] = f.uR[j]; return; }
  switch (f.id) { case 1: r[0] = std::exp(f.theta[1]); r[1] = std::exp(f.theta[2]); break; case 2: r[0] = std::exp(f.theta[2]); break;
                  case 3: r[0] = std::exp(f.theta[1]); break; default: r[0] = 1.0; }
}
// (t: the time index the reference passes; only user models compiled with BSSM_USER_TV read it, with known_tv_params)
void fnT(const FnModel& f, const double* a, double* x, int t) {  // T_fn, R/cpp_example_models.R:415-428 etc.
  if (f.id == 7) { f.user->T(t, f.known_tv, f.tv_rows, f.tv_cols, a, f.theta, f.known, x); return; }
  switch (f.id) {
    case 1: { const double dT = f.known[0], K = f.known[1]; x[0] = a[0]; const double e0 = std::exp(a[0]), r = e0 / (1.0 + e0);
              const double er = std::exp(r * dT);  // (each exponential evaluated once; same values as the snippet's repeated calls)
              x[1] = K * a[1] * er / (K + a[1] * (er - 1)); } break;
    case 2: x[0] = f.theta[0] * (1 - f.theta[1]) + f.theta[1] * a[0]; break;
    case 3: x[0] = std::sin(a[0]); break;
    default: x[0] = a[0];
  }
}
void fnTg(const FnModel& f, const double* a, double* Tg, int t) {  // T_gn, :432-450
  if (f.id == 7) { f.user->Tg(t, f.known_tv, f.tv_rows, f.tv_cols, a, f.theta, f.known, Tg); return; }
  switch (f.id) {
    case 1: { const double dT = f.known[0], K = f.known[1]; const double e0 = std::exp(a[0]), r = e0 / (1 + e0);
              const double er = std::exp(r * dT);
              const double tmp = er / std::pow(K + a[1] * (er - 1), 2);
              Tg[0] = 1.0; Tg[2] = 0; Tg[1] = dT * K * a[1] * (K - a[1]) * tmp * r / (1 + e0); Tg[3] = K * K * tmp; } break;
    case 2: Tg[0] = f.theta[1]; break;
    case 3: Tg[0] = std::cos(a[0]); break;
    default: Tg[0] = 1.0;
  }
}
double fnZ(const FnModel& f, const double* x, int t) { if (f.id == 7) return f.user->Z(t, f.known_tv, f.tv_rows, f.tv_cols, x, f.theta, f.known); return f.id == 1 ? x[1] : ((f.id == 2 || f.id == 3) ? std::exp(x[0]) : x[0]); }
void fnZg(const FnModel& f, const double* a, double* Zg, int t) {
  if (f.id == 7) { f.user->Zg(t, f.known_tv, f.tv_rows, f.tv_cols, a, f.theta, f.known, Zg); return; }
  switch (f.id) { case 1: Zg[0] = 0.0; Zg[1] = 1.0; break; case 2: case 3: Zg[0] = std::exp(a[0]); break; default: Zg[0] = 1.0; }
}
// dmvnorm(x, mean, cov, lwr = false, logd = true) for a DIAGONAL covariance (all shipped P1 / RR are diagonal)
double dmvnorm_diag(const double* x, const double* mean, const double* var, int d) {
  if (d == 1) return -0.5 * (std::log(2.0 * kPi) + std::log(var[0]) + std::pow(x[0] - mean[0], 2) / var[0]);
  double out = 0.0; int cnt = 0;
  for (int j = 0; j < d; j++) {
    if (!(var[j] > kEps)) continue;
    cnt++;
    const double l = std::sqrt(var[j]), z = (x[j] - mean[j]) / l;
    out += -std::log(l) - 0.5 * z * z;
  }
  return out - 0.5 * cnt * std::log(2.0 * kPi);
}
double dnorm_sd(double y, double mean, double sd) { return -0.5 * (std::log(2.0 * kPi) + 2.0 * std::log(sd) + std::pow((y - mean) / sd, 2)); }
// dmvnorm(x, mean, S, lwr = false, logd = true) for any covariance, d <= 8 (src/dmvnorm.cpp:4-58): the components with a
// zero diagonal are dropped, the rest goes through the Cholesky factor
double dmvnorm_cov4(const double* x, const double* mean, const double* S, int d) {
  int idx[kFnM], cnt = 0;
  for (int j = 0; j < d; j++) if (S[j + d * j] > kEps) idx[cnt++] = j;
  double L[kFnMM] = {0}, z2[kFnM], out = -0.5 * cnt * std::log(2.0 * kPi);
  for (int a = 0; a < cnt; a++) {
    for (int b = 0; b <= a; b++) {
      double v = S[idx[a] + d * idx[b]];
      for (int q = 0; q < b; q++) v -= L[a + kFnM * q] * L[b + kFnM * q];
      L[a + kFnM * b] = a == b ? std::sqrt(v) : v / L[b + kFnM * b];
    }
    double sacc = x[idx[a]] - mean[idx[a]];
    for (int b = 0; b < a; b++) sacc -= L[a + kFnM * b] * z2[b];
    z2[a] = sacc / L[a + kFnM * a];
    out += -std::log(L[a + kFnM * a]) - 0.5 * z2[a] * z2[a];
  }
  return out;
}
// the model's constant R as an m x m matrix (shipped models: diagonal) and R R'
void fnRmat(const FnModel& f, double* Rm) {
  const int m = f.m;
  for (int q = 0; q < m * m; q++) Rm[q] = 0.0;
  if (f.id == 7) { for (int q = 0; q < m * m; q++) Rm[q] = f.uRm[q]; return; }
  double rd[kFnM]; fnRdiag(f, rd);
  for (int j = 0; j < m; j++) Rm[j + m * j] = rd[j];
}
void fnRR(const FnModel& f, double* RR) {
  const int m = f.m;
  double Rm[kFnMM]; fnRmat(f, Rm);
  for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) { double sacc = 0; for (int q = 0; q < m; q++) sacc += Rm[i + m * q] * Rm[j + m * q]; RR[i + m * j] = sacc; }
}
// a user model whose R or P1 has off-diagonal entries: the densities below go through the general routine (the shipped
// models and diagonal user models keep the diagonal forms they have always used)
bool fn_full_cov(const FnModel& f) {
  if (f.id != 7) return false;
  const int m = f.m;
  for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) if (j != c && (f.uRm[j + m * c] != 0.0 || f.uP1[j + m * c] != 0.0)) return true;
  return false;
}

// H_fn / R_fn at a state (src/model_ssm_nlg.h:12-13): the snippet's functions where it defines them, the model's numbers otherwise
double fnH_at(const FnModel& f, const double* x, int t) {
  if (f.id == 7 && f.user && f.user->H) return f.user->H(t, f.known_tv, f.tv_rows, f.tv_cols, x, f.theta, f.known);
  return fnH(f);
}
void fnRmat_at(const FnModel& f, const double* x, int t, double* Rm) {  // m x m, zero columns behind k
  if (f.id == 7 && f.user && f.user->R) {
    const int m = f.m;
    double Ru[kFnMM] = {0};
    f.user->R(t, f.known_tv, f.tv_rows, f.tv_cols, x, f.theta, f.known, Ru);  // m x k, leading dimension m
    for (int q = 0; q < m * m; q++) Rm[q] = 0.0;
    for (int c = 0; c < f.k && c < m; c++) for (int j = 0; j < m; j++) Rm[j + m * c] = Ru[j + m * c];
    return;
  }
  fnRmat(f, Rm);
}
void small_rrt(int m, const double* Rm, double* RR) {
  for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) { double sacc = 0; for (int q = 0; q < m; q++) sacc += Rm[i + m * q] * Rm[j + m * q]; RR[i + m * j] = sacc; }
}
void fnRR_at(const FnModel& f, const double* x, int t, double* RR) { double Rm[kFnMM]; fnRmat_at(f, x, t, Rm); small_rrt(f.m, Rm, RR); }
// the approximating model's observation sd per time point
void fn_approx_H(const FnModel& f, const FnSetup& su, std::vector<double>* H) {
  if (f.state_noise()) *H = su.Ht; else H->assign(f.n, su.H);
}

void fn_linearise(const FnModel& f, const double* zpts, const double* tpts, FnSetup* su) {
  const int n = f.n, m = f.m;
  for (int t = 0; t < n; t++) {
    const double* zp = zpts + (size_t)m * t; const double* tp = tpts + (size_t)m * t;
    double* Zt = su->Z.data() + (size_t)m * t; double* Tt = su->T.data() + (size_t)m * m * t; double Tf[kFnM];
    fnZg(f, zp, Zt, t); fnTg(f, tp, Tt, t);
    double za = 0; for (int j = 0; j < m; j++) za += Zt[j] * zp[j];
    su->D[t] = fnZ(f, zp, t) - za;
    fnT(f, tp, Tf, t);
    for (int i = 0; i < m; i++) { double sacc = 0; for (int j = 0; j < m; j++) sacc += Tt[i + m * j] * tp[j]; su->C[i + (size_t)m * t] = Tf[i] - sacc; }
  }
  su->H = fnH(f);
  std::memset(su->R, 0, sizeof(su->R));
  fnRmat(f, su->R);
  if (f.state_noise()) {  // H at the points Z is linearised at, R at the points T is (src/model_ssm_nlg.cpp:100-107, 162-169, 194-203)
    su->Ht.assign(n, 0.0); su->Rt.assign((size_t)m * m * n, 0.0);
    for (int t = 0; t < n; t++) {
      su->Ht[t] = fnH_at(f, zpts + (size_t)m * t, t);
      fnRmat_at(f, tpts + (size_t)m * t, t, su->Rt.data() + (size_t)m * m * t);
    }
  }
}
// the approximating Gaussian model as a HostModel the Kalman routines understand
void fn_gaussian_model(const FnModel& f, const FnSetup& su, HostModel* g) {
  const int n = f.n, m = f.m;
  g->n = n; g->m = m; g->k = m; g->p = 1;
  g->dist.assign(1, BSSM_GAUSSIAN); g->phis.assign(1, 1.0); g->distribution = BSSM_GAUSSIAN;
  g->Ztv = 1; g->Ttv = 1; g->Rtv = 0; g->Dtv = 1; g->Ctv = 1; g->Htv = 0;
  g->y = f.y; g->Z = su.Z; g->T = su.T; g->R.assign(su.R, su.R + m * m); g->D = su.D; g->C = su.C;
  g->a1.assign(m, 0.0); g->P1.assign((size_t)m * m, 0.0);
  fn_a1P1(f, g->a1.data(), g->P1.data());
  g->xbeta.assign(n, 0.0); g->u.assign(n, 1.0); g->H.assign(1, su.H);
  if (f.state_noise()) { g->Rtv = 1; g->R = su.Rt; }  // (the observation sd per time point travels in Approx::H: fn_approx_H)
}
double fn_log_signal_pdf(const FnModel& f, const double* al) {  // src/model_ssm_nlg.cpp:1295-1319
  const int n = f.n, m = f.m;
  double a1[kFnM], P1[kFnMM], rd[kFnM], pv[kFnM], rv[kFnM], RR[kFnMM];
  fn_a1P1(f, a1, P1); fnRdiag(f, rd);
  const bool full = fn_full_cov(f);
  if (full) fnRR(f, RR);
  for (int j = 0; j < m; j++) { pv[j] = P1[j + m * j]; rv[j] = rd[j] * rd[j]; }
  const bool sn = f.state_noise();
  double ll = (full || sn) ? dmvnorm_cov4(al, a1, P1, m) : dmvnorm_diag(al, a1, pv, m);
  if (std::isfinite(f.y[0])) ll += dnorm_sd(f.y[0], fnZ(f, al, 0), fnH_at(f, al, 0));
  for (int t = 0; t < n - 1; t++) {
    double mean[kFnM];
    fnT(f, al + (size_t)m * t, mean, t);
    if (sn) { fnRR_at(f, al + (size_t)m * t, t, RR); ll += dmvnorm_cov4(al + (size_t)m * (t + 1), mean, RR, m); }
    else ll += full ? dmvnorm_cov4(al + (size_t)m * (t + 1), mean, RR, m) : dmvnorm_diag(al + (size_t)m * (t + 1), mean, rv, m);
    if (std::isfinite(f.y[t + 1])) ll += dnorm_sd(f.y[t + 1], fnZ(f, al + (size_t)m * (t + 1), t + 1), fnH_at(f, al + (size_t)m * (t + 1), t + 1));
  }
  return ll;
}
// out = A B A' for m x m column-major operands, m <= 8: the products are formed in the order the generic Mat operators
// use ((A B) first, every entry summed over the inner index from zero), without their temporaries
static inline void small_abat(int m, const double* A, const double* B, double* out) {
  double X[kFnMM];
  for (int j = 0; j < m; j++)
    for (int i = 0; i < m; i++) { double c = 0.0; for (int q = 0; q < m; q++) c += A[i + m * q] * B[q + m * j]; X[i + m * j] = c; }
  for (int j = 0; j < m; j++)
    for (int i = 0; i < m; i++) { double c = 0.0; for (int q = 0; q < m; q++) c += X[i + m * q] * A[j + m * q]; out[i + m * j] = c; }
}
bool fn_ekf(const FnModel& f, std::vector<double>* at, std::vector<double>* att) {  // :284-381 with iekf_iter = 0
  const int n = f.n, m = f.m, mm = m * m;
  std::vector<double> Pt((size_t)mm * (n + 1), 0.0), Ptt((size_t)mm * n, 0.0);
  at->assign((size_t)m * (n + 1), 0.0); att->assign((size_t)m * n, 0.0);
  double RRf[kFnMM]; double H = fnH(f);
  const bool sn = f.state_noise();
  fnRR(f, RRf);  // (diagonal R: exactly diag(r_i * r_i))
  fn_a1P1(f, at->data(), Pt.data());
  for (int t = 0; t < n; t++) {
    const double* a = at->data() + (size_t)m * t; const double* P = Pt.data() + (size_t)mm * t;
    double* af = att->data() + (size_t)m * t; double* Pf = Ptt.data() + (size_t)mm * t;
    if (std::isfinite(f.y[t])) {
      if (sn) H = fnH_at(f, a, t);  // H_fn at the predicted state (:311)
      double Zg[kFnM], PZ[kFnM] = {0}, K[kFnM], F = H * H;
      fnZg(f, a, Zg, t);
      for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) PZ[i] += P[i + m * j] * Zg[j];
      for (int j = 0; j < m; j++) F += Zg[j] * PZ[j];
      if (!std::isfinite(F) || !(F > 0)) return false;
      const double cF = std::sqrt(F), v = f.y[t] - fnZ(f, a, t);
      for (int i = 0; i < m; i++) K[i] = PZ[i] * (1.0 / cF) * (1.0 / cF);
      for (int i = 0; i < m; i++) af[i] = a[i] + K[i] * v;
      double IKZ[kFnMM], J[kFnMM];
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) IKZ[i + m * j] = (i == j ? 1.0 : 0.0) - K[i] * Zg[j];
      small_abat(m, IKZ, P, J);
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) Pf[i + m * j] = J[i + m * j] + K[i] * H * H * K[j];
    } else { std::memcpy(af, a, sizeof(double) * m); std::memcpy(Pf, P, sizeof(double) * mm); }
    double Tg[kFnMM];
    fnT(f, af, at->data() + (size_t)m * (t + 1), t);
    fnTg(f, af, Tg, t);
    double Pn[kFnMM];
    if (sn) fnRR_at(f, af, t, RRf);  // R_fn at the filtered state (:366-370)
    small_abat(m, Tg, Pf, Pn);
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) Pt[i + m * j + (size_t)mm * (t + 1)] = Pn[i + m * j] + RRf[i + m * j];
  }
  return true;
}
}  // namespace

void fn_a1P1(const FnModel& f, double* a1, double* P1) {
  const int m = f.m;
  for (int j = 0; j < m; j++) a1[j] = 0;
  for (int q = 0; q < m * m; q++) P1[q] = 0;
  if (f.id == 7) { for (int j = 0; j < m; j++) a1[j] = f.ua1[j]; for (int q = 0; q < m * m; q++) P1[q] = f.uP1[q]; return; }
  switch (f.id) {
    case 1: a1[0] = f.known[2]; a1[1] = f.known[3]; P1[0] = f.known[4]; P1[3] = f.known[5]; break;
    case 2: a1[0] = f.theta[0]; P1[0] = std::pow(std::exp(f.theta[2]), 2) / (1 - std::pow(f.theta[1], 2)); break;
    default: a1[0] = 0; P1[0] = 1;
  }
}

// initial proposal of the EKF particle filter (src/model_ssm_nlg.cpp:1154-1161): EKF update of (a1, P1) with y_0, then
// psd_chol of the updated covariance.  Everything m x m column-major, m <= 8.
void fn_ekpf_init(const FnModel& f, double* a1, double* P1, double* att1, double* L1) {
  const int m = f.m;
  fn_a1P1(f, a1, P1);
  double Ptt[kFnMM] = {0};
  const double y = f.y[0];
  if (std::isfinite(y)) {  // ekf_update_step (:1264-1293)
    double Zg[kFnM] = {0}, PZ[kFnM] = {0}, K[kFnM];
    fnZg(f, a1, Zg, 0);
    const double H = fnH_at(f, a1, 0);
    double F = H * H;
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) PZ[i] += P1[i + m * j] * Zg[j];
    for (int j = 0; j < m; j++) F += Zg[j] * PZ[j];
    const double cF = std::sqrt(F), v = y - fnZ(f, a1, 0);
    for (int i = 0; i < m; i++) { K[i] = PZ[i] * (1.0 / cF) * (1.0 / cF); att1[i] = a1[i] + K[i] * v; }
    double IKZ[kFnMM], tmp[kFnMM];
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) IKZ[i + m * j] = (i == j ? 1.0 : 0.0) - K[i] * Zg[j];
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) { double acc = 0; for (int q = 0; q < m; q++) acc += IKZ[i + m * q] * P1[q + m * j]; tmp[i + m * j] = acc; }
    for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) { double acc = 0; for (int q = 0; q < m; q++) acc += tmp[i + m * q] * IKZ[j + m * q]; Ptt[i + m * j] = acc + K[i] * H * H * K[j]; }
  } else {
    for (int i = 0; i < m; i++) att1[i] = a1[i];
    for (int q = 0; q < m * m; q++) Ptt[q] = P1[q];
  }
  // psd_chol (src/psd_chol.cpp:5-18)
  for (int q = 0; q < m * m; q++) L1[q] = 0.0;
  double dmax = Ptt[0];
  for (int j = 1; j < m; j++) dmax = std::max(dmax, Ptt[j + m * j]);
  const double thr = std::max(kEps, kEps * m * dmax);
  std::vector<int> idx;
  for (int j = 0; j < m; j++) if (Ptt[j + m * j] > thr) idx.push_back(j);
  if (!idx.empty()) {
    Mat Lm;
    chol_lower_on(Mat(m, m, Ptt), idx, &Lm);
    for (int q = 0; q < m * m; q++) L1[q] = Lm.a[q];
  }
}

bool fn_psi_setup(const FnModel& f, const double* is_mode, int max_iter, double conv_tol, FnSetup* su, bool loglik_only) {
  const int n = f.n, m = f.m, mm = m * m;
  su->Z.assign((size_t)m * n, 0.0); su->T.assign((size_t)mm * n, 0.0); su->D.assign(n, 0.0); su->C.assign((size_t)m * n, 0.0);
  su->mode.assign((size_t)m * n, 0.0); su->scales.assign(n, 0.0);
  su->alphahat.assign((size_t)m * (n + 1), 0.0); su->Vt.assign((size_t)mm * (n + 1), 0.0); su->Ct.assign((size_t)mm * (n + 1), 0.0);
  su->approx_loglik = 0; su->iters = 0; su->ok = true;
  HostModel g;
  Approx ap;
  ap.y = f.y;
  std::vector<Mat> sm;
  double gll_last = 0.0; bool have_gll = false;  // Gaussian log-likelihood of the model the LAST smoother pass ran on
  auto smooth_mode = [&](std::vector<double>* out) {
    fn_gaussian_model(f, *su, &g);
    fn_approx_H(f, *su, &ap.H);
    const Sys S(g);
    double gl = std::nan("");
    if (!fast_smoother(S, ap, &sm, &gl)) return false;
    have_gll = !std::isnan(gl); gll_last = gl;
    out->resize((size_t)m * n);
    for (int t = 0; t < n; t++) for (int j = 0; j < m; j++) (*out)[j + (size_t)m * t] = sm[t](j, 0);
    for (double v : *out) if (!std::isfinite(v)) return false;
    return true;
  };
  if (is_mode) {  // approximate_for_is (:146-176): scales stay 0, approx_loglik 0
    su->mode.assign(is_mode, is_mode + (size_t)m * n);
    fn_linearise(f, su->mode.data(), su->mode.data(), su);
  } else {        // approximate() (:62-143)
    std::vector<double> at, att, nm;
    if (!fn_ekf(f, &at, &att)) { su->ok = false; return false; }
    fn_linearise(f, at.data(), att.data(), su);  // approximate_by_ekf (:179-215)
    if (!smooth_mode(&su->mode)) { su->ok = false; return false; }
    if (max_iter > 0) {
      double ll = fn_log_signal_pdf(f, su->mode.data()), rel = 1.0e300, ad = 1;
      int i = 0;
      while (i < max_iter && rel > conv_tol && ad > 1e-4) {
        i++;
        fn_linearise(f, su->mode.data(), su->mode.data(), su);
        if (!smooth_mode(&nm)) { su->ok = false; return false; }
        const double lln = fn_log_signal_pdf(f, nm.data());
        ad = lln - ll; rel = ad / std::fabs(ll);
        if (!std::isfinite(lln)) { su->ok = false; return false; }
        // the reference's back-tracking branch (:118-138) needs rel < -conv_tol and ad > 1e-4 at once: unreachable
        su->mode = nm; ll = lln;
      }
      su->iters = i;
    }
    // the approximating model is the one the last smoother pass was run on: its forward pass already gave the likelihood
    double gll;
    if (have_gll) {
      gll = gll_last;
    } else {
      fn_gaussian_model(f, *su, &g);
      fn_approx_H(f, *su, &ap.H);
      const Sys S(g);
      gll = gaussian_loglik(S, ap);
    }
    // update_scales (:854-879)
    double rd[kFnM], rv[kFnM], RRs[kFnMM];
    fnRdiag(f, rd);
    const bool full = fn_full_cov(f);
    if (full) fnRR(f, RRs);
    for (int j = 0; j < m; j++) rv[j] = rd[j] * rd[j];
    const bool sn = f.state_noise();
    for (int t = 0; t < n; t++) {
      const double* md = su->mode.data() + (size_t)m * t;
      if (std::isfinite(f.y[t])) {
        double am = su->D[t];
        for (int j = 0; j < m; j++) am += su->Z[j + (size_t)m * t] * md[j];
        const double Ha = sn ? su->Ht[t] : su->H, HH = Ha * Ha;
        su->scales[t] = dnorm_sd(f.y[t], fnZ(f, md, t), fnH_at(f, md, t)) - dmvnorm_diag(&f.y[t], &am, &HH, 1);
      }
    }
    for (int t = 1; t < n; t++) {
      const double* mp = su->mode.data() + (size_t)m * (t - 1); const double* md = su->mode.data() + (size_t)m * t;
      double mean[kFnM], amean[kFnM];
      fnT(f, mp, mean, t - 1);
      for (int i = 0; i < m; i++) { double sacc = 0; for (int j = 0; j < m; j++) sacc += su->T[i + m * j + (size_t)mm * (t - 1)] * mp[j]; amean[i] = su->C[i + (size_t)m * (t - 1)] + sacc; }
      if (sn) {  // the model's R at the previous mode point against the approximating R_{t-1} (:869-876)
        double RRm[kFnMM], RRa[kFnMM];
        fnRR_at(f, mp, t - 1, RRm);
        small_rrt(m, su->Rt.data() + (size_t)mm * (t - 1), RRa);
        su->scales[t] += dmvnorm_cov4(md, mean, RRm, m) - dmvnorm_cov4(md, amean, RRa, m);
      } else
      su->scales[t] += full ? dmvnorm_cov4(md, mean, RRs, m) - dmvnorm_cov4(md, amean, RRs, m) : dmvnorm_diag(md, mean, rv, m) - dmvnorm_diag(md, amean, rv, m);
    }
    double ssc = 0; for (double v : su->scales) ssc += v;
    su->approx_loglik = gll + ssc;
  }
  if (loglik_only) return true;  // the approximate chain needs the mode and the log-likelihood only
  fn_gaussian_model(f, *su, &g);
  fn_approx_H(f, *su, &ap.H);
  {
    bool fok = true;
    if (lg_fast_path(g, ap, &su->alphahat, &su->Vt, &su->Ct, &fok)) { su->ok = fok; return fok; }
  }
  const Sys S(g);
  std::vector<Mat> at, Vt, Ct;
  if (!smoother_ccov(S, ap, &at, &Vt, &Ct)) { su->ok = false; return false; }
  for (int t = 0; t <= n; t++) for (int q = 0; q < mm; q++) if (!std::isfinite(Vt[t].a[q]) || !std::isfinite(Ct[t].a[q])) { su->ok = false; return false; }
  conditional_cov(&Vt, &Ct);
  for (int t = 0; t <= n; t++) {
    for (int i = 0; i < m; i++) su->alphahat[i + (size_t)m * t] = at[t](i, 0);
    for (int q = 0; q < mm; q++) { su->Vt[q + (size_t)mm * t] = Vt[t].a[q]; su->Ct[q + (size_t)mm * t] = Ct[t].a[q]; }
  }
  return true;
}

bool fn_psi_setup_begin(const FnModel& f, const double* is_mode, FnSetup* su, LgPacked* lg) {
  if (!is_mode || bssm::opt().generic_setup) return false;
  const int n = f.n, m = f.m, mm = m * m;
  su->Z.assign((size_t)m * n, 0.0); su->T.assign((size_t)mm * n, 0.0); su->D.assign(n, 0.0); su->C.assign((size_t)m * n, 0.0);
  su->mode.assign(is_mode, is_mode + (size_t)m * n); su->scales.assign(n, 0.0);
  su->alphahat.assign((size_t)m * (n + 1), 0.0); su->Vt.assign((size_t)mm * (n + 1), 0.0); su->Ct.assign((size_t)mm * (n + 1), 0.0);
  su->approx_loglik = 0; su->iters = 0; su->ok = true;
  fn_linearise(f, su->mode.data(), su->mode.data(), su);  // approximate_for_is (:146-176): scales stay 0
  HostModel g;
  fn_gaussian_model(f, *su, &g);
  Approx ap;
  ap.y = f.y;
  fn_approx_H(f, *su, &ap.H);
  lg_pack(g, ap, lg);
  return true;
}

void build_fn_psi_tables(const FnModel& f, const FnSetup& su, FilterTables* ft, std::vector<double>* amats) {
  const int n = f.n, m = f.m, MP = pad_dim(m), KP = MP, mm = m * m;  // (a 3-state user model runs in the 4-state kernels)
  const int MS = MP * MP + MP * KP + MP;
  ft->MP = MP; ft->KP = KP; ft->mats_tstride = MS;
  ft->mats.assign((size_t)n * MS, 0.0);
  for (int t = 0; t < n; t++) {  // proposal alphahat(t+1) + Ct(t+1)(a - alphahat(t)) + Vt(t+1) u  (:1032-1033)
    double* mt = ft->mats.data() + (size_t)t * MS;
    const double* Ct1 = su.Ct.data() + (size_t)mm * (t + 1); const double* Vt1 = su.Vt.data() + (size_t)mm * (t + 1);
    const double* ah0 = su.alphahat.data() + (size_t)m * t; const double* ah1 = su.alphahat.data() + (size_t)m * (t + 1);
    for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) { mt[j + MP * c] = Ct1[j + m * c]; mt[MP * MP + j + MP * c] = Vt1[j + m * c]; }
    for (int j = 0; j < m; j++) { double sacc = 0; for (int c = 0; c < m; c++) sacc += Ct1[j + m * c] * ah0[c]; mt[MP * MP + MP * KP + j] = ah1[j] - sacc; }
  }
  const int stride = ROW_HDR + SER_HDR + MP;
  ft->obs.assign((size_t)n * stride, 0.0);
  for (int t = 0; t < n; t++) {
    double* row = ft->obs.data() + (size_t)t * stride;
    if (!std::isfinite(f.y[t])) continue;
    row[RH_FLAG] = 1.0; row[RH_SCALE] = su.scales[t];
    double* ob = row + ROW_HDR;
    ob[OB_FLAG] = 1.0; ob[OB_Y] = f.y[t]; ob[OB_D] = su.D[t]; ob[OB_PSIIH] = 1.0 / (f.state_noise() ? su.Ht[t] : su.H);
    for (int c = 0; c < m; c++) ob[SER_HDR + c] = su.Z[c + (size_t)m * t];
  }
  ft->init.assign((size_t)MP + (size_t)MP * MP, 0.0);
  for (int j = 0; j < m; j++) ft->init[j] = su.alphahat[j];
  for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) ft->init[MP + j + MP * c] = su.Vt[j + m * c];
  // a snippet with an R_fn carries the approximating R_t R_t' behind (T_t, C_t): the row the kernels of ITS library step by
  // (pf_functors.h: fn_amat_row)
  const bool rfn = f.id == 7 && f.user && f.user->R;
  const int AR = fn_amats_row(MP, rfn);
  amats->assign((size_t)n * AR, 0.0);
  for (int t = 0; t < n; t++) {
    double* am = amats->data() + (size_t)t * AR;
    for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) am[j + MP * c] = su.T[j + m * c + (size_t)mm * t];
    for (int j = 0; j < m; j++) am[MP * MP + j] = su.C[j + (size_t)m * t];
    if (rfn) {
      double RRa[kFnMM];
      small_rrt(m, su.Rt.data() + (size_t)mm * t, RRa);
      for (int c = 0; c < m; c++) for (int j = 0; j < m; j++) am[MP * MP + MP + j + MP * c] = RRa[j + m * c];
    }
  }
}

std::string update_model_builtin(int model_type, const int32_t* fl, const double* th, int n_par, bssm_ung_model* mod,
                                 double* T, double* R, double* P1, double* a1, double* C, double* beta) {
  const int m = mod->m, k = mod->k;
  switch (model_type) {
    case BSSM_BSM_NG: {  // src/model_bsm_ng.cpp:17-51 ; flags {slope, seasonal, noise, fixed0, fixed1, fixed2, phi_est}
      const int slope = fl[0], seasonal = fl[1], noise = fl[2];
      const int level_est = fl[3] == 0, slope_est = slope && fl[4] == 0, seasonal_est = seasonal && fl[5] == 0;
      const int phi_est = fl[6];
      if (fl[3] + fl[4] + fl[5] < 3 || noise) {
        if (level_est) R[0] = std::exp(th[0]);
        if (slope_est) R[1 + m * 1] = std::exp(th[level_est]);
        if (seasonal_est) R[(1 + slope) + m * (1 + slope)] = std::exp(th[level_est + slope_est]);
        if (noise) {
          const int col = 1 + slope + seasonal;
          if (col >= k) return "bsm_ng noise column outside R";
          R[(m - 1) + m * col] = std::exp(th[level_est + slope_est + seasonal_est]);
          P1[(m - 1) + m * (m - 1)] = R[(m - 1) + m * col] * R[(m - 1) + m * col];
        }
      }
      if (phi_est) mod->phi = std::exp(th[level_est + slope_est + seasonal_est + noise]);
      if (mod->n_xreg > 0) for (int q = 0; q < mod->n_xreg; q++) beta[q] = th[n_par - mod->n_xreg + q];
    } break;
    case BSSM_SVM_MODEL: {  // src/model_svm.cpp:12-29 ; flags {svm_type}
      if (fl[0] == 0) mod->phi = th[2];
      else { a1[0] = th[2]; for (int t = 0; t < (mod->Ctv ? mod->n : 1); t++) C[t] = th[2] * (1.0 - th[0]); }
      T[0] = th[0];
      R[0] = th[1];
      P1[0] = th[1] * th[1] / (1 - th[0] * th[0]);
    } break;
    case BSSM_AR1_NG: {  // src/model_ar1_ng.cpp:12-36 ; flags {mu_est, phi_est}
      const double rho = th[0], sigma = std::exp(th[1]);
      T[0] = rho;
      R[0] = sigma;
      if (fl[0]) { a1[0] = th[2]; for (int t = 0; t < (mod->Ctv ? mod->n : 1); t++) C[t] = th[2] * (1.0 - rho); }
      P1[0] = sigma * sigma / (1.0 - rho * rho);
      if (fl[1]) mod->phi = std::exp(th[2 + fl[0]]);
      if (mod->n_xreg > 0) for (int q = 0; q < mod->n_xreg; q++) beta[q] = th[n_par - mod->n_xreg + q];
    } break;
    default: return "no built-in update_model for this model_type (ssm_ung/ssm_mng take pre-expanded models)";
  }
  return "";
}

}  // namespace bssm
