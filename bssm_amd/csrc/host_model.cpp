// host_model.cpp -- host-side model handling for the MI355X particle-filter engine.
// The O(n m^3) Gaussian-approximation set-up of the psi-APF stays on the host (SURVEY.md
// section 8a row a12: serial per theta, small); it is threaded over theta-draws by the caller.
#include "host_model.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "pf_device.h"

namespace bssm {

namespace {

constexpr double kZeroTol = 1e-12;  // src/model_ssm_ung.h:21, src/model_ssm_ulg.h:19
constexpr double kEps = std::numeric_limits<double>::epsilon();
constexpr double kPi = 3.14159265358979323846;

// ------------------------------------------------------------------ tiny dense matrix
struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
  Mat(int r_, int c_, const double* src) : r(r_), c(c_), a(src, src + (size_t)r_ * c_) {}
  double& operator()(int i, int j) { return a[i + (size_t)r * j]; }
  double operator()(int i, int j) const { return a[i + (size_t)r * j]; }
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
Mat operator+(const Mat& A, const Mat& B) { Mat C = A; for (size_t i = 0; i < C.a.size(); i++) C.a[i] += B.a[i]; return C; }
Mat operator-(const Mat& A, const Mat& B) { Mat C = A; for (size_t i = 0; i < C.a.size(); i++) C.a[i] -= B.a[i]; return C; }
Mat operator*(const Mat& A, double s) { Mat C = A; for (auto& v : C.a) v *= s; return C; }
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
    double off = 0.0;
    for (int j = 0; j < m; j++) for (int i = 0; i < j; i++) off += A(i, j) * A(i, j);
    if (off < 1e-300) break;
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

// views into a HostModel at time t
struct Sys {
  const HostModel& h;
  explicit Sys(const HostModel& hm) : h(hm) {}
  Mat Zc(int t) const { return Mat(h.m, 1, h.Z.data() + (size_t)h.m * (t * h.Ztv)); }
  Mat Tm(int t) const { return Mat(h.m, h.m, h.T.data() + (size_t)h.m * h.m * (t * h.Ttv)); }
  Mat Rm(int t) const { return Mat(h.m, h.k, h.R.data() + (size_t)h.m * h.k * (t * h.Rtv)); }
  Mat Cc(int t) const { return Mat(h.m, 1, h.C.data() + (size_t)h.m * (t * h.Ctv)); }
  double Dv(int t) const { return h.D[t * h.Dtv]; }
};

// ------------------------------------------------------------------ Kalman machinery of the approximating model
struct Approx {  // pseudo-observations of the Gaussian approximation
  std::vector<double> y, H;  // n each (H = sd)
};

struct KfStep { double F = 0, v = 0; bool obs = false; Mat K; };

// prediction step; joseph = numerically better covariance form (src/model_ssm_ulg.cpp:276-277)
KfStep kf_predict(const Sys& S, const Approx& ap, int t, const Mat& at, const Mat& Pt, bool joseph, Mat* at1, Mat* Pt1) {
  const int m = S.h.m;
  const Mat Z = S.Zc(t), T = S.Tm(t), R = S.Rm(t);
  const Mat RR = R * tr(R);
  const double HH = ap.H[t] * ap.H[t];
  KfStep st;
  st.K = Mat(m, 1);
  st.F = (tr(Z) * Pt * Z)(0, 0) + HH;
  const double ytmp = ap.y[t] - S.h.xbeta[t];
  if (std::isfinite(ytmp) && st.F > kZeroTol) {
    st.obs = true;
    st.v = ytmp - S.Dv(t) - (tr(Z) * at)(0, 0);
    const Mat PZ = Pt * Z;
    for (int i = 0; i < m; i++) st.K(i, 0) = PZ(i, 0) / st.F;
    *at1 = S.Cc(t) + T * (at + st.K * st.v);
    Mat inner;
    if (joseph) {
      const Mat IKZ = Mat::eye(m) - st.K * tr(Z);
      inner = IKZ * Pt * tr(IKZ) + (st.K * HH) * tr(st.K);
    } else {
      inner = Pt - (st.K * tr(st.K)) * st.F;
    }
    *Pt1 = T * inner * tr(T) + RR;
  } else {
    *at1 = S.Cc(t) + T * at;
    *Pt1 = T * Pt * tr(T) + RR;
  }
  symmatu(*Pt1);
  return st;
}

// src/model_ssm_ulg.cpp:106-138
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
    const KfStep st = kf_predict(S, ap, t, at, Pt, false, &at1, &Pt1);
    if (st.obs) ll -= 0.5 * (LOG2PI + std::log(st.F) + st.v * st.v / st.F);
    at = at1; Pt = Pt1;
  }
  return ll;
}

// src/model_ssm_ulg.cpp:252-305 -> smoothed state means, m x (n+1)
std::vector<Mat> fast_smoother(const Sys& S, const Approx& ap) {
  const HostModel& h = S.h;
  const int n = h.n, m = h.m;
  std::vector<Mat> at(n + 1);
  std::vector<KfStep> st(n);
  at[0] = Mat(m, 1, h.a1.data());
  Mat Pt(m, m, h.P1.data()), Pt1;
  for (int t = 0; t < n; t++) { st[t] = kf_predict(S, ap, t, at[t], Pt, true, &at[t + 1], &Pt1); Pt = Pt1; }
  std::vector<Mat> rt(n, Mat(m, 1));
  for (int t = n - 1; t > 0; t--) {
    if (st[t].obs) {
      const Mat Z = S.Zc(t);
      const Mat L = S.Tm(t) * (Mat::eye(m) - st[t].K * tr(Z));
      const Mat Lr = tr(L) * rt[t];
      for (int i = 0; i < m; i++) rt[t - 1](i, 0) = Z(i, 0) / st[t].F * st[t].v + Lr(i, 0);
    } else {
      rt[t - 1] = tr(S.Tm(t)) * rt[t];
    }
  }
  const Mat a1(m, 1, h.a1.data()), P1(m, m, h.P1.data());
  if (std::isfinite(h.y[0]) && st[0].F > kZeroTol) {  // :292 tests the ORIGINAL y(0)
    const Mat Z = S.Zc(0);
    const Mat L = S.Tm(0) * (Mat::eye(m) - st[0].K * tr(Z));
    Mat tmp = tr(L) * rt[0];
    for (int i = 0; i < m; i++) tmp(i, 0) = Z(i, 0) / st[0].F * st[0].v + tmp(i, 0);
    at[0] = a1 + P1 * tmp;
  } else {
    at[0] = a1 + P1 * (tr(S.Tm(0)) * rt[0]);
  }
  for (int t = 0; t < n - 1; t++) {
    const Mat R = S.Rm(t);
    at[t + 1] = S.Cc(t) + S.Tm(t) * at[t] + (R * tr(R)) * rt[t];
  }
  return at;
}

// src/model_ssm_ulg.cpp:419-473
void smoother_ccov(const Sys& S, const Approx& ap, std::vector<Mat>* at, std::vector<Mat>* Pt, std::vector<Mat>* cc) {
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
    (*cc)[t] = (*Pt)[t + 1];
  }
  Mat rt(m, 1), Nt(m, m);
  const Mat I = Mat::eye(m);
  for (int t = n - 1; t >= 0; t--) {
    const Mat T = S.Tm(t);
    if (st[t].obs) {
      const Mat Z = S.Zc(t);
      const Mat L = T * (I - st[t].K * tr(Z));
      (*cc)[t] = (*Pt)[t] * tr(L) * (I - Nt * (*cc)[t]);
      Mat r2 = tr(L) * rt;
      for (int i = 0; i < m; i++) r2(i, 0) = Z(i, 0) / st[t].F * st[t].v + r2(i, 0);
      rt = r2;
      Mat N2 = tr(L) * Nt * L;
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) N2(i, j) = Z(i, 0) * Z(j, 0) / st[t].F + N2(i, j);
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
}

// src/conditional_dist.cpp:4-31 (use_svd = true)
void conditional_cov(std::vector<Mat>* Vt, std::vector<Mat>* Ct) {
  const int n1 = (int)Vt->size(), m = (*Vt)[0].r;
  Mat U;
  std::vector<double> s;
  for (int t = n1 - 1; t > 0; t--) {
    sym_singular((*Vt)[t - 1], &U, &s);
    Mat W(m, m);
    for (int c = 0; c < m; c++) {
      if (!(s[c] > kEps * m * s[0])) continue;
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) W(i, j) += U(i, c) * U(j, c) / s[c];
    }
    const Mat tmp = tr((*Ct)[t - 1]) * W;
    (*Vt)[t] = (*Vt)[t] - tmp * (*Ct)[t - 1];
    (*Ct)[t] = tmp;
    sym_singular((*Vt)[t], &U, &s);
    for (int c = 0; c < m; c++) for (int i = 0; i < m; i++) (*Vt)[t](i, c) = U(i, c) * std::sqrt(s[c]);
  }
  sym_singular((*Vt)[0], &U, &s);
  for (int c = 0; c < m; c++) for (int i = 0; i < m; i++) (*Vt)[0](i, c) = U(i, c) * std::sqrt(s[c]);
}

// src/model_ssm_ung.cpp:302-342
void laplace_iter(const HostModel& h, const std::vector<double>& signal, Approx* ap) {
  const double phi = h.phi;
  for (int t = 0; t < h.n; t++) {
    const double y = h.y[t], sg = signal[t];
    double HH = 0, yy = 0;
    switch (h.distribution) {
      case BSSM_SVM: {
        double tmp = y;
        if (std::fabs(tmp) < 1e-4) tmp = 1e-4;
        HH = 2.0 * std::exp(sg) / ((tmp / phi) * (tmp / phi));
        yy = sg + 1.0 - 0.5 * HH;
      } break;
      case BSSM_POISSON: HH = 1.0 / (std::exp(sg) * h.u[t]); yy = y * HH + sg - 1.0; break;
      case BSSM_BINOMIAL: { const double e = std::exp(sg); HH = (1.0 + e) * (1.0 + e) / (h.u[t] * e); yy = y * HH + sg - 1.0 - e; } break;
      case BSSM_NEGBIN: { const double e = std::exp(sg) * h.u[t]; HH = (phi + e) * (phi + e) / (phi * e * (y + phi)); yy = sg + (phi + e) * (y - e) / ((y + phi) * e); } break;
      case BSSM_GAMMA: { const double e = std::exp(sg) * h.u[t]; HH = e / (y * phi); yy = sg - e / y + 1; } break;
      default: break;
    }
    ap->y[t] = yy;
    ap->H[t] = std::sqrt(HH);
  }
}

// src/model_ssm_ung.cpp:241-292
void update_scales(const HostModel& h, const Approx& ap, const std::vector<double>& mode, std::vector<double>* scales) {
  const double phi = h.phi;
  scales->assign(h.n, 0.0);
  for (int t = 0; t < h.n; t++) {
    const double y = h.y[t];
    if (!std::isfinite(y)) continue;
    const double md = mode[t];
    const double g = 0.5 * std::pow((ap.y[t] - md) / ap.H[t], 2.0);
    double v = 0;
    switch (h.distribution) {
      case BSSM_SVM: v = -0.5 * (md + std::pow(y / phi, 2.0) * std::exp(-md)); break;
      case BSSM_POISSON: v = y * md - h.u[t] * std::exp(md); break;
      case BSSM_BINOMIAL: v = y * md - h.u[t] * std::log1p(std::exp(md)); break;
      case BSSM_NEGBIN: v = y * md - (y + phi) * std::log(phi + h.u[t] * std::exp(md)); break;
      case BSSM_GAMMA: v = -phi * (md + (y * std::exp(-md) / h.u[t])); break;
      default: break;
    }
    (*scales)[t] = v + g;
  }
}

// approximate(): src/model_ssm_ung.cpp:87-140 (first call, approx_state == -1)
int run_laplace(const HostModel& h, const Sys& S, Approx* ap, std::vector<double>* mode) {
  *mode = h.initial_mode;
  int it = 0;
  double diff = h.conv_tol + 1;
  std::vector<double> nm(h.n);
  while (it < h.max_iter && diff > h.conv_tol) {
    it++;
    laplace_iter(h, *mode, ap);
    const std::vector<Mat> sm = fast_smoother(S, *ap);
    for (int t = 0; t < h.n; t++) {
      if (h.distribution == BSSM_SVM) nm[t] = sm[t](0, 0);
      else nm[t] = h.xbeta[t] + S.Dv(t) + (tr(S.Zc(t)) * sm[t])(0, 0);
    }
    diff = 0.0;
    for (int t = 0; t < h.n; t++) diff += (nm[t] - (*mode)[t]) * (nm[t] - (*mode)[t]);
    diff /= h.n;
    *mode = nm;
  }
  return it;
}

double approx_const_term(const HostModel& h, const Approx& ap) {  // src/model_ssm_ung.cpp:347-370
  double c = const_term(h);
  int nf = 0;
  double sl = 0;
  for (int t = 0; t < h.n; t++) if (std::isfinite(h.y[t])) { nf++; sl += std::log(ap.H[t]); }
  return c - (-0.5 * nf * std::log(2 * kPi) - sl);
}

}  // namespace

// ------------------------------------------------------------------ public pieces
int pad_dim(int d) { int p = 1; while (p < d) p <<= 1; return p; }

std::string ingest_model(const bssm_ung_model* s, bool need_mode, HostModel* o) {
  if (!s) return "model is NULL";
  if (s->n < 1 || s->m < 1 || s->k < 1) return "n, m, k must be positive";
  if (s->m > 16) return "state dimension m > 16 is not supported by the kernels";
  if (s->k > s->m) return "k (columns of R) must not exceed m";
  if (s->distribution < 0 || s->distribution > 5) return "unknown distribution code";
  if (!s->y || !s->Z || !s->T || !s->R || !s->a1 || !s->P1) return "y, Z, T, R, a1, P1 must be non-NULL";
  const int n = s->n, m = s->m, k = s->k;
  o->n = n; o->m = m; o->k = k;
  o->Ztv = s->Ztv != 0; o->Ttv = s->Ttv != 0; o->Rtv = s->Rtv != 0; o->Dtv = s->Dtv != 0; o->Ctv = s->Ctv != 0; o->Htv = s->Htv != 0;
  o->distribution = s->distribution;
  o->max_iter = s->max_iter > 0 ? s->max_iter : 100;
  o->local_approx = s->local_approx;
  o->conv_tol = s->conv_tol > 0 ? s->conv_tol : 1e-8;
  o->phi = s->phi;
  o->y.assign(s->y, s->y + n);
  o->Z.assign(s->Z, s->Z + (size_t)m * (o->Ztv ? n : 1));
  o->T.assign(s->T, s->T + (size_t)m * m * (o->Ttv ? n : 1));
  o->R.assign(s->R, s->R + (size_t)m * k * (o->Rtv ? n : 1));
  o->a1.assign(s->a1, s->a1 + m);
  o->P1.assign(s->P1, s->P1 + (size_t)m * m);
  if (s->D) o->D.assign(s->D, s->D + (o->Dtv ? n : 1)); else { o->D.assign(1, 0.0); o->Dtv = 0; }
  if (s->C) o->C.assign(s->C, s->C + (size_t)m * (o->Ctv ? n : 1)); else { o->C.assign(m, 0.0); o->Ctv = 0; }
  if (s->u) o->u.assign(s->u, s->u + n); else o->u.assign(n, 1.0);
  o->xbeta.assign(n, 0.0);
  if (s->n_xreg > 0) {  // xbeta = xreg * beta (src/model_ssm_ung.h:84)
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
    o->initial_mode.assign(s->initial_mode, s->initial_mode + n);
  }
  if ((o->distribution == BSSM_NEGBIN || o->distribution == BSSM_GAMMA || o->distribution == BSSM_SVM) && !(o->phi > 0))
    return "phi must be positive";
  return "";
}

double const_term(const HostModel& h) {
  double res = 0.0;
  int nfin = 0;
  for (int t = 0; t < h.n; t++) {
    const double y = h.y[t];
    if (!std::isfinite(y)) continue;
    nfin++;
    const double u = h.u[t], phi = h.phi;
    switch (h.distribution) {
      case BSSM_POISSON: res += y * std::log(u) - std::lgamma(y + 1); break;
      case BSSM_BINOMIAL: res += lchoose_ts(u, y); break;
      case BSSM_NEGBIN: res += lchoose_ts(y + phi - 1, y) + phi * std::log(phi) + y * std::log(u); break;
      case BSSM_GAMMA: res += phi * std::log(phi) - std::lgamma(phi) + (phi - 1) * std::log(y) - phi * std::log(u); break;
      case BSSM_GAUSSIAN: res += norm_log_const(h.H[t * h.Htv]); break;
      default: break;
    }
  }
  if (h.distribution == BSSM_SVM) res = nfin * norm_log_const(h.phi);
  return res;
}

bool psi_setup(const HostModel& h, const double* is_mode, PsiSetup* out) {
  const int n = h.n, m = h.m;
  const Sys S(h);
  Approx ap;
  ap.y.assign(n, 0.0);
  ap.H.assign(n, 0.0);
  if (is_mode) {  // approximate_for_is: src/model_ssm_ung.cpp:143-161
    out->mode.assign(is_mode, is_mode + n);
    laplace_iter(h, out->mode, &ap);
    update_scales(h, ap, out->mode, &out->scales);
    out->approx_loglik = 0.0;
    out->iterations = 0;
  } else {  // psi_filter set-up: src/model_ssm_ung.cpp:514-528
    out->iterations = run_laplace(h, S, &ap, &out->mode);
    const double gll = gaussian_loglik(S, ap);
    update_scales(h, ap, out->mode, &out->scales);
    double ssc = 0;
    for (double v : out->scales) ssc += v;
    out->approx_loglik = gll + approx_const_term(h, ap) + ssc;
  }
  out->approx_y = ap.y;
  out->approx_H = ap.H;
  std::vector<Mat> at, Vt, Ct;
  smoother_ccov(S, ap, &at, &Vt, &Ct);  // :533
  conditional_cov(&Vt, &Ct);            // :534
  out->alphahat.resize((size_t)m * (n + 1));
  out->Vt.resize((size_t)m * m * (n + 1));
  out->Ct.resize((size_t)m * m * (n + 1));
  bool ok = true;
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

double approx_loglik_only(const HostModel& h) {
  const Sys S(h);
  Approx ap;
  ap.y.assign(h.n, 0.0);
  ap.H.assign(h.n, 0.0);
  std::vector<double> mode, scales;
  run_laplace(h, S, &ap, &mode);
  const double gll = gaussian_loglik(S, ap);
  update_scales(h, ap, mode, &scales);
  double ssc = 0;
  for (double v : scales) ssc += v;
  return gll + approx_const_term(h, ap) + ssc;
}

namespace {
void fill_obs_rows(const HostModel& h, int MP, const PsiSetup* su, std::vector<double>* obs) {
  const int n = h.n, m = h.m, stride = OBS_HDR + MP;
  obs->assign((size_t)n * stride, 0.0);
  for (int t = 0; t < n; t++) {
    double* ob = obs->data() + (size_t)t * stride;
    const double y = h.y[t], u = h.u[t];
    ob[OB_Y] = y;
    ob[OB_U] = u;
    ob[OB_FLAG] = std::isfinite(y) ? 1.0 : 0.0;
    switch (h.distribution) {
      case BSSM_SVM:  // signal is the first state; Z, D, xbeta are ignored (src/model_ssm_ung.cpp:456)
        ob[OB_D] = 0.0;
        ob[OBS_HDR] = 1.0;
        ob[OB_AUX] = (y / h.phi) * (y / h.phi);
        break;
      case BSSM_GAUSSIAN:  // mu = D + Z'alpha, no xbeta (src/model_ssm_ulg.cpp:583-584)
        ob[OB_D] = h.D[t * h.Dtv];
        for (int c = 0; c < m; c++) ob[OBS_HDR + c] = h.Z[c + (size_t)m * (t * h.Ztv)];
        ob[OB_AUX] = 1.0 / (h.H[t * h.Htv] * h.H[t * h.Htv]);
        break;
      default:
        ob[OB_D] = h.D[t * h.Dtv] + h.xbeta[t];
        for (int c = 0; c < m; c++) ob[OBS_HDR + c] = h.Z[c + (size_t)m * (t * h.Ztv)];
        ob[OB_AUX] = h.distribution == BSSM_NEGBIN ? y + h.phi : (h.distribution == BSSM_GAMMA ? y / u : 0.0);
    }
    if (su) {
      ob[OB_PSIY] = su->approx_y[t];
      ob[OB_PSIIH] = 1.0 / su->approx_H[t];
      ob[OB_SCALE] = su->scales[t];
    }
    if (!std::isfinite(y)) {  // keep NaN out of the arithmetic of skipped rows
      ob[OB_Y] = 0.0; ob[OB_AUX] = 0.0; ob[OB_PSIY] = 0.0; ob[OB_PSIIH] = 0.0; ob[OB_SCALE] = 0.0;
    }
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
