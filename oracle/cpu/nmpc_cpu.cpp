// CPU baseline of the NMPC hot path: multiple-shooting transcription + structure-exploiting (Riccati) primal-dual interior
// point, C++17 + OpenMP over the instances of a batch, behind a C ABI that takes the product's own problem description
// (`hilo_nmpc_desc`, include/hilo_hip.h).
//
// TEST INFRASTRUCTURE / BASELINE ONLY: loaded by bench.py's `cpu_baseline` leg and by tests/ (through oracle/cpu/__init__.py),
// never by the product package.  BASELINE.md section 3 plans exactly this: "the build's own CPU restatement of the same
// algorithms behind the same C ABI (C++17, -O3, OpenMP over instances)".
//
// What it restates (each function cites what it follows):
//   * the transcription of `NMPC._setup` for a pre-discretised model with integration_method='discrete'
//     (hilo_mpc/modules/controller/mpc.py:1455-1787): v = [x_0..x_N | u_0..u_{N-1}] in scaled variables (:1462-1485),
//     rows x_{k+1} - Phi(x_k, u_k) (:1667), objective sum_k l(x_k, u_k) + V(x_N) (:1676-1682), x_0 pinned (:797-802),
//     QuadraticCost (hilo_mpc/util/modeling.py:243-283), explicit Runge-Kutta of order 1..4 (modeling.py:1239-1250);
//   * the interior-point algorithm of oracle/nmpc.py::DenseIpm statement by statement (Waechter & Biegler 2006 with IPOPT's
//     default constants: monotone barrier update, fraction to the boundary, filter line search with second-order correction,
//     inertia correction, barrier-augmented feasibility restoration), with the dense KKT solve replaced by the Riccati
//     recursion over the stages - the inertia of the KKT matrix is correct exactly when every stage's reduced input Hessian
//     F_k = R_k + B_k' P_{k+1} B_k has a Cholesky factor;
//   * derivatives: second-order forward mode (value, gradient, Hessian with respect to the interval's z = (x_k, u_k)) pushed
//     through the Runge-Kutta stages - what CasADi's SX graph of the discretised model provides to IPOPT (exact Hessian).
// Scope: tracking NMPC with box bounds and scaling on the models chemostat4, chemostat4 with a learned growth rate (C4) and
// pendulum4 (configs C2, C4 and the pendulum tests);
// anything else in the descriptor is refused.  Validated against oracle/nmpc.py in tests/test_cpu_baseline.py before it is timed.
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/hilo_hip.h"
#include "models_cpu.h"

namespace {

constexpr double INF = std::numeric_limits<double>::infinity();
constexpr double EPS = std::numeric_limits<double>::epsilon();
enum { SOLVED = 1, ACCEPTABLE = 2, INFEASIBLE = 3, RESTORATION_FAILED = 4, MAXITER = 5, OTHER = -1 };   // optimizer.py:1085-1104

char g_err[512] = "";

// ---- second-order forward mode: value, gradient [N], packed symmetric Hessian [N (N+1) / 2] ---------------------------------
template <int N>
struct H2 {
  static constexpr int NH = N * (N + 1) / 2;
  double v, g[N], h[NH];
  H2() {}
  H2(double c) : v(c) {
    for (int i = 0; i < N; ++i) g[i] = 0.0;
    for (int i = 0; i < NH; ++i) h[i] = 0.0;
  }
  static H2 seed(double c, int i) {
    H2 r(c);
    r.g[i] = 1.0;
    return r;
  }
  double hess(int i, int j) const { return i >= j ? h[i * (i + 1) / 2 + j] : h[j * (j + 1) / 2 + i]; }
};
template <int N> H2<N> operator+(const H2<N>& a, const H2<N>& b) {
  H2<N> r;
  r.v = a.v + b.v;
  for (int i = 0; i < N; ++i) r.g[i] = a.g[i] + b.g[i];
  for (int i = 0; i < H2<N>::NH; ++i) r.h[i] = a.h[i] + b.h[i];
  return r;
}
template <int N> H2<N> operator-(const H2<N>& a, const H2<N>& b) {
  H2<N> r;
  r.v = a.v - b.v;
  for (int i = 0; i < N; ++i) r.g[i] = a.g[i] - b.g[i];
  for (int i = 0; i < H2<N>::NH; ++i) r.h[i] = a.h[i] - b.h[i];
  return r;
}
template <int N> H2<N> operator-(const H2<N>& a) {
  H2<N> r;
  r.v = -a.v;
  for (int i = 0; i < N; ++i) r.g[i] = -a.g[i];
  for (int i = 0; i < H2<N>::NH; ++i) r.h[i] = -a.h[i];
  return r;
}
template <int N> H2<N> operator*(const H2<N>& a, const H2<N>& b) {
  H2<N> r;
  r.v = a.v * b.v;
  for (int i = 0; i < N; ++i) r.g[i] = a.v * b.g[i] + b.v * a.g[i];
  int q = 0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j, ++q) r.h[q] = a.v * b.h[q] + b.v * a.h[q] + a.g[i] * b.g[j] + a.g[j] * b.g[i];
  return r;
}
// composition with a scalar function: f(a) given f, f', f'' at a.v
template <int N> H2<N> chain(const H2<N>& a, double f, double f1, double f2) {
  H2<N> r;
  r.v = f;
  for (int i = 0; i < N; ++i) r.g[i] = f1 * a.g[i];
  int q = 0;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j <= i; ++j, ++q) r.h[q] = f1 * a.h[q] + f2 * a.g[i] * a.g[j];
  return r;
}
template <int N> H2<N> inv(const H2<N>& a) {
  const double i1 = 1.0 / a.v;
  return chain(a, i1, -i1 * i1, 2.0 * i1 * i1 * i1);
}
template <int N> H2<N> operator/(const H2<N>& a, const H2<N>& b) { return a * inv(b); }
template <int N> H2<N> operator+(const H2<N>& a, double c) { H2<N> r = a; r.v += c; return r; }
template <int N> H2<N> operator+(double c, const H2<N>& a) { H2<N> r = a; r.v += c; return r; }
template <int N> H2<N> operator-(const H2<N>& a, double c) { H2<N> r = a; r.v -= c; return r; }
template <int N> H2<N> operator-(double c, const H2<N>& a) { H2<N> r = -a; r.v += c; return r; }
template <int N> H2<N> operator*(const H2<N>& a, double c) {
  H2<N> r;
  r.v = a.v * c;
  for (int i = 0; i < N; ++i) r.g[i] = a.g[i] * c;
  for (int i = 0; i < H2<N>::NH; ++i) r.h[i] = a.h[i] * c;
  return r;
}
template <int N> H2<N> operator*(double c, const H2<N>& a) { return a * c; }
template <int N> H2<N> operator/(const H2<N>& a, double c) { return a * (1.0 / c); }
template <int N> H2<N> operator/(double c, const H2<N>& a) { return inv(a) * c; }
template <int N> H2<N> sin(const H2<N>& a) { const double s = std::sin(a.v), c = std::cos(a.v); return chain(a, s, c, -s); }
template <int N> H2<N> cos(const H2<N>& a) { const double s = std::sin(a.v), c = std::cos(a.v); return chain(a, c, -s, -c); }
template <int N> H2<N> exp(const H2<N>& a) { const double e = std::exp(a.v); return chain(a, e, e, e); }
using std::cos;
using namespace hilo_cpu;   // models, tableaux (models_cpu.h)

struct Problem {
  int model_id, N, order, n_sub, max_iter, acceptable_iter, nx, nu, np;
  double dt, tol, acceptable_tol, mu_init, relax;
  std::vector<double> Wz, zref, WN, xrefN, xlb, xub, ulb, uub, sx, su, xg, ug;
};

// scaled shooting map xs+ = Phi(xs * sx, us * su, p) / sx (hilo_mpc/modules/base.py:1562-1591)
template <class M, class T>
void phi_scaled(const Problem& pb, const T* xs, const T* us, const double* p, T* out) {
  constexpr int NX = M::NX, NU = M::NU;
  T x[NX], u[NU], k[4][NX], xi[NX];
  for (int i = 0; i < NX; ++i) x[i] = xs[i] * pb.sx[i];
  for (int i = 0; i < NU; ++i) u[i] = us[i] * pb.su[i];
  const Tableau& t = tableau(pb.order);
  const double h = pb.dt / pb.n_sub;
  for (int sub = 0; sub < pb.n_sub; ++sub) {
    for (int i = 0; i < t.s; ++i) {
      for (int c = 0; c < NX; ++c) xi[c] = x[c];
      for (int j = 0; j < i; ++j)
        if (t.A[i][j] != 0.0)
          for (int c = 0; c < NX; ++c) xi[c] = xi[c] + (h * t.A[i][j]) * k[j][c];
      M::ode(xi, u, p, k[i]);
    }
    for (int i = 0; i < t.s; ++i)
      if (t.b[i] != 0.0)
        for (int c = 0; c < NX; ++c) x[c] = x[c] + (h * t.b[i]) * k[i][c];
  }
  for (int i = 0; i < NX; ++i) out[i] = x[i] * (1.0 / pb.sx[i]);
}

// ---- one instance ---------------------------------------------------------------------------------------------------------------
template <class M>
struct Solver {
  static constexpr int NX = M::NX, NU = M::NU, NZ = NX + NU;
  using HD = H2<NZ>;
  const Problem& pb;
  const int N;
  const double* p;
  double x0[NX];
  // iterate: X[k] k = 0..N (X[0] = x0), U[k]; multipliers of the bounds for x_1..x_N and u_0..u_{N-1}
  std::vector<double> X, U, lam, zlx, zux, zlu, zuu;
  // bounds (relaxed), shared by all stages
  double lbx[NX], ubx[NX], lbu[NU], ubu[NU];
  bool hlx[NX], hux[NX], hlu[NU], huu[NU];
  int nb, m, nw;
  // derivative buffers
  std::vector<double> gx, gu, c, A, Bm, Hz, HN;    // gx [N+1][NX] (index k = 1..N used), gu [N][NU], c [N][NX], A [N][NX][NX], B [N][NX][NU]
  // Riccati
  std::vector<double> P, pv, K, kff, dX, dU, lamn;
  std::vector<std::pair<double, double>> filt;
  // work vectors of solve() / restore(): allocated once per solver object (one per thread), not per instance - with many
  // threads the allocator otherwise serialises the batch
  std::vector<double> ct, Xt, Ut, qx, qu, sgx, sgu, dgx, dgu, dzlx, dzux, dzlu, dzuu, csoc, dXs, dUs, lams, lam_step, Xs, Us, cs,
      r_lam0, r_qx, r_qu, r_sgx, r_sgu, r_Xt, r_Ut, r_ct;

  Solver(const Problem& pb_, const double* p_) : pb(pb_), N(pb_.N), p(p_) {
    X.resize((N + 1) * NX); U.resize(N * NU); lam.assign(N * NX, 0.0);
    zlx.resize(N * NX); zux.resize(N * NX); zlu.resize(N * NU); zuu.resize(N * NU);
    gx.resize((N + 1) * NX); gu.resize(N * NU); c.resize(N * NX); A.resize(N * NX * NX); Bm.resize(N * NX * NU);
    Hz.resize(N * NZ * NZ); HN.resize(NX * NX);
    P.resize((N + 1) * NX * NX); pv.resize((N + 1) * NX); K.resize(N * NU * NX); kff.resize(N * NU);
    dX.resize((N + 1) * NX); dU.resize(N * NU); lamn.resize(N * NX);
    ct.resize(c.size()); Xt.resize(X.size()); Ut.resize(U.size()); qx.resize((N + 1) * NX); qu.resize(N * NU);
    sgx.assign((N + 1) * NX, 0.0); sgu.resize(N * NU); dgx.resize((N + 1) * NX); dgu.resize(N * NU);
    dzlx.resize(N * NX); dzux.resize(N * NX); dzlu.resize(N * NU); dzuu.resize(N * NU); csoc.resize(c.size());
    dXs.resize(dX.size()); dUs.resize(dU.size()); lams.resize(lamn.size()); lam_step.resize(lamn.size());
    Xs.resize(X.size()); Us.resize(U.size()); cs.resize(c.size());
    r_lam0.assign(N * NX, 0.0); r_qx.resize((N + 1) * NX); r_qu.resize(N * NU); r_sgx.assign((N + 1) * NX, 0.0); r_sgu.resize(N * NU);
    r_Xt.resize(X.size()); r_Ut.resize(U.size()); r_ct.resize(c.size());
    const double r = pb.relax;
    nb = 0;
    for (int i = 0; i < NX; ++i) {
      hlx[i] = std::isfinite(pb.xlb[i]); hux[i] = std::isfinite(pb.xub[i]);
      lbx[i] = hlx[i] ? pb.xlb[i] - r * std::max(1.0, std::fabs(pb.xlb[i])) : -INF;
      ubx[i] = hux[i] ? pb.xub[i] + r * std::max(1.0, std::fabs(pb.xub[i])) : INF;
      nb += N * ((int)hlx[i] + (int)hux[i]);
    }
    for (int i = 0; i < NU; ++i) {
      hlu[i] = std::isfinite(pb.ulb[i]); huu[i] = std::isfinite(pb.uub[i]);
      lbu[i] = hlu[i] ? pb.ulb[i] - r * std::max(1.0, std::fabs(pb.ulb[i])) : -INF;
      ubu[i] = huu[i] ? pb.uub[i] + r * std::max(1.0, std::fabs(pb.uub[i])) : INF;
      nb += N * ((int)hlu[i] + (int)huu[i]);
    }
    nb = std::max(1, nb);
    m = N * NX;
    nw = N * NZ;
  }

  // IPOPT initialisation (W&B sec. 3.6): x <- P[x] with kappa_1 = kappa_2 = 1e-2
  static double push(double w, double lb, double ub, bool hl, bool hu) {
    const double bp = 1e-2, bf = 1e-2;
    double pl = bp * std::max(1.0, std::fabs(lb)), pu = bp * std::max(1.0, std::fabs(ub));
    if (hl && hu) { pl = std::min(pl, bf * (ub - lb)); pu = std::min(pu, bf * (ub - lb)); }
    if (hl) w = std::max(w, lb + pl);
    if (hu) w = std::min(w, ub - pu);
    return w;
  }

  // objective and defects at (Xc, Uc) (values only; DenseIpm.eval_fc)
  double eval_fc(const double* Xc, const double* Uc, double* cc) const {
    double f = 0.0, z[NZ], ph[NX];
    for (int k = 0; k < N; ++k) {
      for (int i = 0; i < NX; ++i) z[i] = Xc[k * NX + i] - pb.zref[i];
      for (int i = 0; i < NU; ++i) z[NX + i] = Uc[k * NU + i] - pb.zref[NX + i];
      for (int i = 0; i < NZ; ++i)
        for (int j = 0; j < NZ; ++j) f += z[i] * pb.Wz[i * NZ + j] * z[j];
      phi_scaled<M, double>(pb, Xc + k * NX, Uc + k * NU, p, ph);
      for (int i = 0; i < NX; ++i) cc[k * NX + i] = Xc[(k + 1) * NX + i] - ph[i];
    }
    double d[NX];
    for (int i = 0; i < NX; ++i) d[i] = Xc[N * NX + i] - pb.xrefN[i];
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NX; ++j) f += d[i] * pb.WN[i * NX + j] * d[j];
    return f;
  }

  // f, gradient, defects, stage Jacobians and the Hessian of the Lagrangian by stages (DenseIpm.eval_all)
  double eval_all(const double* lm) {
    double f = 0.0, z[NZ];
    HD xs[NX], us[NU], ph[NX];
    std::fill(gx.begin(), gx.end(), 0.0);
    for (int k = 0; k < N; ++k) {
      for (int i = 0; i < NX; ++i) z[i] = X[k * NX + i] - pb.zref[i];
      for (int i = 0; i < NU; ++i) z[NX + i] = U[k * NU + i] - pb.zref[NX + i];
      double gz[NZ];
      for (int i = 0; i < NZ; ++i) {
        double s = 0.0;
        for (int j = 0; j < NZ; ++j) { s += pb.Wz[i * NZ + j] * z[j]; f += z[i] * pb.Wz[i * NZ + j] * z[j]; }
        gz[i] = 2.0 * s;   // Wz symmetric (QuadraticCost builds it from diagonal / symmetric blocks)
      }
      for (int i = 0; i < NX; ++i) gx[k * NX + i] += gz[i];
      for (int i = 0; i < NU; ++i) gu[k * NU + i] = gz[NX + i];
      for (int i = 0; i < NX; ++i) xs[i] = HD::seed(X[k * NX + i], i);
      for (int i = 0; i < NU; ++i) us[i] = HD::seed(U[k * NU + i], NX + i);
      phi_scaled<M, HD>(pb, xs, us, p, ph);
      double* Hk = &Hz[k * NZ * NZ];
      for (int i = 0; i < NZ; ++i)
        for (int j = 0; j < NZ; ++j) Hk[i * NZ + j] = 2.0 * pb.Wz[i * NZ + j];
      for (int r = 0; r < NX; ++r) {
        c[k * NX + r] = X[(k + 1) * NX + r] - ph[r].v;
        for (int j = 0; j < NX; ++j) A[(k * NX + r) * NX + j] = ph[r].g[j];
        for (int j = 0; j < NU; ++j) Bm[(k * NX + r) * NU + j] = ph[r].g[NX + j];
        const double l = lm[k * NX + r];
        for (int i = 0; i < NZ; ++i)
          for (int j = 0; j < NZ; ++j) Hk[i * NZ + j] -= l * ph[r].hess(i, j);
      }
    }
    double d[NX];
    for (int i = 0; i < NX; ++i) d[i] = X[N * NX + i] - pb.xrefN[i];
    for (int i = 0; i < NX; ++i) {
      double s = 0.0;
      for (int j = 0; j < NX; ++j) { s += pb.WN[i * NX + j] * d[j]; f += d[i] * pb.WN[i * NX + j] * d[j]; HN[i * NX + j] = 2.0 * pb.WN[i * NX + j]; }
      gx[N * NX + i] += 2.0 * s;
    }
    return f;
  }

  // slacks of variable (x_k component i, k >= 1) / (u_k component i)
  double slx(const double* Xc, int k, int i) const { return hlx[i] ? Xc[k * NX + i] - lbx[i] : 1.0; }
  double sux(const double* Xc, int k, int i) const { return hux[i] ? ubx[i] - Xc[k * NX + i] : 1.0; }
  double slu(const double* Uc, int k, int i) const { return hlu[i] ? Uc[k * NU + i] - lbu[i] : 1.0; }
  double suu(const double* Uc, int k, int i) const { return huu[i] ? ubu[i] - Uc[k * NU + i] : 1.0; }

  double barrier(double f, const double* Xc, const double* Uc, double mu) const {
    double s = 0.0;
    for (int k = 1; k <= N; ++k)
      for (int i = 0; i < NX; ++i) {
        if (hlx[i]) s += std::log(Xc[k * NX + i] - lbx[i]);
        if (hux[i]) s += std::log(ubx[i] - Xc[k * NX + i]);
      }
    for (int k = 0; k < N; ++k)
      for (int i = 0; i < NU; ++i) {
        if (hlu[i]) s += std::log(Uc[k * NU + i] - lbu[i]);
        if (huu[i]) s += std::log(ubu[i] - Uc[k * NU + i]);
      }
    return f - mu * s;
  }

  static double l1(const std::vector<double>& v) { double s = 0; for (double a : v) s += std::fabs(a); return s; }
  static double linf(const std::vector<double>& v) { double s = 0; for (double a : v) s = std::max(s, std::fabs(a)); return s; }

  // scaled optimality error E_mu (W&B eq. 5, 6; DenseIpm.errors) at the current iterate and derivative buffers
  double errors(double mu, double* dual_o = nullptr, double* prim_o = nullptr, double* compl_o = nullptr) const {
    double dual = 0.0, cmp = 0.0, zsum = 0.0;
    for (int k = 1; k <= N; ++k)
      for (int i = 0; i < NX; ++i) {
        double r = gx[k * NX + i] + lam[(k - 1) * NX + i];
        if (k < N)
          for (int q = 0; q < NX; ++q) r -= A[(k * NX + q) * NX + i] * lam[k * NX + q];
        const int j = (k - 1) * NX + i;
        r += -zlx[j] + zux[j];
        dual = std::max(dual, std::fabs(r));
        if (hlx[i]) cmp = std::max(cmp, std::fabs(slx(X.data(), k, i) * zlx[j] - mu));
        if (hux[i]) cmp = std::max(cmp, std::fabs(sux(X.data(), k, i) * zux[j] - mu));
        zsum += std::fabs(zlx[j]) + std::fabs(zux[j]);
      }
    for (int k = 0; k < N; ++k)
      for (int i = 0; i < NU; ++i) {
        double r = gu[k * NU + i];
        for (int q = 0; q < NX; ++q) r -= Bm[(k * NX + q) * NU + i] * lam[k * NX + q];
        const int j = k * NU + i;
        r += -zlu[j] + zuu[j];
        dual = std::max(dual, std::fabs(r));
        if (hlu[i]) cmp = std::max(cmp, std::fabs(slu(U.data(), k, i) * zlu[j] - mu));
        if (huu[i]) cmp = std::max(cmp, std::fabs(suu(U.data(), k, i) * zuu[j] - mu));
        zsum += std::fabs(zlu[j]) + std::fabs(zuu[j]);
      }
    const double prim = linf(c), smax = 100.0;
    const double s_d = std::max(smax, (l1(lam) + zsum) / (m + nb)) / smax, s_c = std::max(smax, zsum / nb) / smax;
    if (dual_o) *dual_o = dual;
    if (prim_o) *prim_o = prim;
    if (compl_o) *compl_o = cmp;
    return std::max(std::max(dual / s_d, prim), cmp / s_c);
  }

  // Riccati solve of  [H + D, J'; J, 0] [d; lam+] = [-q; -cc]  with H = blkdiag(Hs_k) (+ HNs), D = diag(dgx, dgu), q = (qx, qu).
  // Returns false when a stage's reduced input Hessian is not positive definite (wrong inertia).  Hs == nullptr: identity.
  bool riccati(const double* Hs, const double* HNs, const double* dgx, const double* dgu, const double* qx, const double* qu,
               const double* cc) {
    double* PN = &P[N * NX * NX];
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NX; ++j) PN[i * NX + j] = (HNs ? HNs[i * NX + j] : (Hs ? 0.0 : (i == j ? 1.0 : 0.0))) + (i == j ? dgx[N * NX + i] : 0.0);
    for (int i = 0; i < NX; ++i) pv[N * NX + i] = qx[N * NX + i];
    for (int k = N - 1; k >= 0; --k) {
      const double* Pn = &P[(k + 1) * NX * NX];
      const double* Ak = &A[k * NX * NX];
      const double* Bk = &Bm[k * NX * NU];
      double pc[NX], PA[NX][NX], PB[NX][NU];
      for (int i = 0; i < NX; ++i) {
        double s = pv[(k + 1) * NX + i];
        for (int j = 0; j < NX; ++j) s -= Pn[i * NX + j] * cc[k * NX + j];
        pc[i] = s;
        for (int j = 0; j < NX; ++j) { double t = 0; for (int q = 0; q < NX; ++q) t += Pn[i * NX + q] * Ak[q * NX + j]; PA[i][j] = t; }
        for (int j = 0; j < NU; ++j) { double t = 0; for (int q = 0; q < NX; ++q) t += Pn[i * NX + q] * Bk[q * NU + j]; PB[i][j] = t; }
      }
      auto Hs_at = [&](int i, int j) { return Hs ? Hs[k * NZ * NZ + i * NZ + j] : (i == j ? 1.0 : 0.0); };
      double F[NU][NU], G[NU][NX], fu[NU];
      for (int i = 0; i < NU; ++i) {
        for (int j = 0; j < NU; ++j) {
          double t = Hs_at(NX + i, NX + j) + (i == j ? dgu[k * NU + i] : 0.0);
          for (int q = 0; q < NX; ++q) t += Bk[q * NU + i] * PB[q][j];
          F[i][j] = t;
        }
        for (int j = 0; j < NX; ++j) {
          double t = Hs_at(NX + i, j);
          for (int q = 0; q < NX; ++q) t += Bk[q * NU + i] * PA[q][j];
          G[i][j] = t;
        }
        double t = qu[k * NU + i];
        for (int q = 0; q < NX; ++q) t += Bk[q * NU + i] * pc[q];
        fu[i] = t;
      }
      // Cholesky F = L L'
      double L[NU][NU];
      for (int i = 0; i < NU; ++i)
        for (int j = 0; j <= i; ++j) {
          double t = F[i][j];
          for (int q = 0; q < j; ++q) t -= L[i][q] * L[j][q];
          if (i == j) {
            if (!(t > 0.0) || !std::isfinite(t)) return false;
            L[i][i] = std::sqrt(t);
          } else {
            L[i][j] = t / L[j][j];
          }
        }
      auto solveF = [&](double* v) {   // v <- F^-1 v
        for (int i = 0; i < NU; ++i) { double t = v[i]; for (int q = 0; q < i; ++q) t -= L[i][q] * v[q]; v[i] = t / L[i][i]; }
        for (int i = NU - 1; i >= 0; --i) { double t = v[i]; for (int q = i + 1; q < NU; ++q) t -= L[q][i] * v[q]; v[i] = t / L[i][i]; }
      };
      double col[NU];
      for (int i = 0; i < NU; ++i) col[i] = -fu[i];
      solveF(col);
      for (int i = 0; i < NU; ++i) kff[k * NU + i] = col[i];
      if (k == 0) break;   // dx_0 = 0: no gain, no cost-to-go needed
      for (int j = 0; j < NX; ++j) {
        for (int i = 0; i < NU; ++i) col[i] = -G[i][j];
        solveF(col);
        for (int i = 0; i < NU; ++i) K[(k * NU + i) * NX + j] = col[i];
      }
      double* Pk = &P[k * NX * NX];
      for (int i = 0; i < NX; ++i) {
        for (int j = 0; j < NX; ++j) {
          double t = Hs_at(i, j) + (i == j ? dgx[k * NX + i] : 0.0);
          for (int q = 0; q < NX; ++q) t += Ak[q * NX + i] * PA[q][j];
          for (int q = 0; q < NU; ++q) t += G[q][i] * K[(k * NU + q) * NX + j];
          Pk[i * NX + j] = t;
        }
        double t = qx[k * NX + i];
        for (int q = 0; q < NX; ++q) t += Ak[q * NX + i] * pc[q];
        for (int q = 0; q < NU; ++q) t += G[q][i] * kff[k * NU + q];
        pv[k * NX + i] = t;
      }
      for (int i = 0; i < NX; ++i)      // symmetrise against round-off drift
        for (int j = 0; j < i; ++j) Pk[i * NX + j] = Pk[j * NX + i] = 0.5 * (Pk[i * NX + j] + Pk[j * NX + i]);
    }
    for (int i = 0; i < NX; ++i) dX[i] = 0.0;
    for (int k = 0; k < N; ++k) {
      for (int i = 0; i < NU; ++i) {
        double t = kff[k * NU + i];
        if (k > 0)
          for (int j = 0; j < NX; ++j) t += K[(k * NU + i) * NX + j] * dX[k * NX + j];
        dU[k * NU + i] = t;
      }
      for (int i = 0; i < NX; ++i) {
        double t = -cc[k * NX + i];
        for (int j = 0; j < NX; ++j) t += A[(k * NX + i) * NX + j] * dX[k * NX + j];
        for (int j = 0; j < NU; ++j) t += Bm[(k * NX + i) * NU + j] * dU[k * NU + j];
        dX[(k + 1) * NX + i] = t;
      }
      const double* Pn = &P[(k + 1) * NX * NX];
      for (int i = 0; i < NX; ++i) {
        double t = pv[(k + 1) * NX + i];
        for (int j = 0; j < NX; ++j) t += Pn[i * NX + j] * dX[(k + 1) * NX + j];
        lamn[k * NX + i] = -t;
      }
    }
    return true;
  }

  // largest step keeping the bounded variables inside the fraction-to-the-boundary rule (W&B eq. 8)
  double alpha_primal(const double* dXc, const double* dUc, double tau) const {
    double a = 1.0;
    for (int k = 1; k <= N; ++k)
      for (int i = 0; i < NX; ++i) {
        const double d = dXc[k * NX + i];
        if (hlx[i] && d < 0) a = std::min(a, -tau * (X[k * NX + i] - lbx[i]) / d);
        if (hux[i] && d > 0) a = std::min(a, tau * (ubx[i] - X[k * NX + i]) / d);
      }
    for (int k = 0; k < N; ++k)
      for (int i = 0; i < NU; ++i) {
        const double d = dUc[k * NU + i];
        if (hlu[i] && d < 0) a = std::min(a, -tau * (U[k * NU + i] - lbu[i]) / d);
        if (huu[i] && d > 0) a = std::min(a, tau * (ubu[i] - U[k * NU + i]) / d);
      }
    return a;
  }

  bool filter_ok(double th, double ph) const {
    for (const auto& e : filt)
      if (th >= e.first && ph - 10 * EPS * std::fabs(e.second) >= e.second) return false;
    return true;
  }

  // feasibility restoration (DenseIpm._restore): 0 = new point in X/U, 1 = failed, 2 = locally infeasible
  int restore(double mu, double tau, double theta_max) {
    std::vector<double>&lam0 = r_lam0, &qx = r_qx, &qu = r_qu, &sgx = r_sgx, &sgu = r_sgu, &Xt = r_Xt, &Ut = r_Ut, &ct = r_ct;
    std::fill(lam0.begin(), lam0.end(), 0.0);
    std::fill(sgx.begin(), sgx.end(), 0.0);
    eval_all(lam0.data());
    const double th_start = l1(c);
    double th = th_start, th_ref = th;
    for (int it = 0; it < 50; ++it) {
      if (it % 10 == 9) {
        if (th > (1 - 1e-4) * th_ref && th > 1e-6) return 2;
        th_ref = th;
      }
      const double mu_r = std::max(mu, linf(c));
      for (int k = 1; k <= N; ++k)
        for (int i = 0; i < NX; ++i) {
          const double sl = slx(X.data(), k, i), su = sux(X.data(), k, i);
          sgx[k * NX + i] = (hlx[i] ? mu_r / (sl * sl) : 0.0) + (hux[i] ? mu_r / (su * su) : 0.0);
          qx[k * NX + i] = -(hlx[i] ? mu_r / sl : 0.0) + (hux[i] ? mu_r / su : 0.0);
        }
      for (int k = 0; k < N; ++k)
        for (int i = 0; i < NU; ++i) {
          const double sl = slu(U.data(), k, i), su = suu(U.data(), k, i);
          sgu[k * NU + i] = (hlu[i] ? mu_r / (sl * sl) : 0.0) + (huu[i] ? mu_r / (su * su) : 0.0);
          qu[k * NU + i] = -(hlu[i] ? mu_r / sl : 0.0) + (huu[i] ? mu_r / su : 0.0);
        }
      if (!riccati(nullptr, nullptr, sgx.data(), sgu.data(), qx.data(), qu.data(), c.data())) return 1;
      double dmax = 0.0;
      for (int i = NX; i < (N + 1) * NX; ++i) dmax = std::max(dmax, std::fabs(dX[i]));
      for (double d : dU) dmax = std::max(dmax, std::fabs(d));
      if (dmax <= 1e-9 && th > 1e-6) return 2;
      double alpha = alpha_primal(dX.data(), dU.data(), tau), ft = 0.0, tht = 0.0;
      bool ok = false;
      while (alpha > 1e-10) {
        for (size_t i = 0; i < X.size(); ++i) Xt[i] = X[i] + alpha * dX[i];
        for (size_t i = 0; i < U.size(); ++i) Ut[i] = U[i] + alpha * dU[i];
        ft = eval_fc(Xt.data(), Ut.data(), ct.data());
        tht = l1(ct);
        if (std::isfinite(tht) && tht <= (1 - 1e-4 * alpha) * th) { ok = true; break; }
        alpha *= 0.5;
      }
      if (!ok) return th > 1e-6 ? 2 : 1;
      X = Xt; U = Ut; th = tht;
      if (th <= 0.9 * th_start && th <= theta_max) {
        const double ph = barrier(ft, X.data(), U.data(), mu);
        bool acc = true;
        for (const auto& e : filt)
          if (th >= e.first && ph >= e.second) { acc = false; break; }
        if (acc) return 0;
      }
      eval_all(lam0.data());
    }
    return 1;
  }

  // DenseIpm.solve_data for one instance.  v0: warm start in the reference layout (primal only, mpc.py:725-726) or NULL.
  void solve(const double* x0_orig, const double* v0, double* v_opt, double* f_opt, double* u0, int* status_o, int* iters_o,
             double* kkt_o) {
    const double kappa_eps = 10., kappa_mu = 0.2, theta_mu = 1.5, tau_min = 0.99, kappa_sigma = 1e10;
    const double gamma_theta = 1e-5, gamma_phi = 1e-8, delta_ls = 1., s_theta = 1.1, s_phi = 2.3, eta_phi = 1e-8;
    const double alpha_red = 0.5, alpha_min_frac = 0.05, kappa_soc = 0.99;
    const double dw_min = 1e-20, dw_0 = 1e-4, dw_max = 1e40, kw_minus = 1. / 3, kw_plus = 8., kw_plus_bar = 100.;
    const int max_filter = 16, max_soc = 4;
    const double mu_floor = std::min(pb.tol, 1e-4) / (kappa_eps + 1.);
    for (int i = 0; i < NX; ++i) X[i] = x0_orig[i] / pb.sx[i];
    for (int k = 1; k <= N; ++k)
      for (int i = 0; i < NX; ++i) X[k * NX + i] = push(v0 ? v0[k * NX + i] : pb.xg[i], lbx[i], ubx[i], hlx[i], hux[i]);
    for (int k = 0; k < N; ++k)
      for (int i = 0; i < NU; ++i) U[k * NU + i] = push(v0 ? v0[(N + 1) * NX + k * NU + i] : pb.ug[i], lbu[i], ubu[i], hlu[i], huu[i]);
    std::fill(lam.begin(), lam.end(), 0.0);
    for (int j = 0; j < N * NX; ++j) { zlx[j] = hlx[j % NX] ? 1.0 : 0.0; zux[j] = hux[j % NX] ? 1.0 : 0.0; }
    for (int j = 0; j < N * NU; ++j) { zlu[j] = hlu[j % NU] ? 1.0 : 0.0; zuu[j] = huu[j % NU] ? 1.0 : 0.0; }
    double mu = pb.mu_init, tau = std::max(tau_min, 1 - mu), delta_last = 0.0;
    int status = 0, iters = 0, acc_count = 0;
    filt.clear();
    std::fill(sgx.begin(), sgx.end(), 0.0);
    double f = eval_fc(X.data(), U.data(), ct.data());
    const double theta0 = l1(ct);
    const double theta_min = 1e-4 * std::max(1.0, theta0), theta_max = 1e4 * std::max(1.0, theta0);

    for (int it = 0; it <= pb.max_iter; ++it) {
      f = eval_all(lam.data());
      const double E0 = errors(0.0);
      if (!std::isfinite(E0)) { status = OTHER; break; }
      if (E0 <= pb.tol) { status = SOLVED; break; }
      acc_count = E0 <= pb.acceptable_tol ? acc_count + 1 : 0;
      if (acc_count >= pb.acceptable_iter) { status = ACCEPTABLE; break; }
      if (it == pb.max_iter) { status = MAXITER; break; }
      // ---- barrier parameter (monotone, W&B eq. 7; floor = IPOPT's min(tol, compl_inf_tol) / (kappa_eps + 1)) ----
      for (int rep = 0; rep < 20; ++rep) {
        if (!(errors(mu) <= kappa_eps * mu && mu > mu_floor * (1 + 1e-12))) break;
        mu = std::max(mu_floor, std::min(kappa_mu * mu, std::pow(mu, theta_mu)));
        tau = std::max(tau_min, 1 - mu);
        filt.clear();
      }
      // ---- search direction with inertia correction (W&B Alg. IC) ----
      for (int k = 1; k <= N; ++k)
        for (int i = 0; i < NX; ++i) {
          const int j = (k - 1) * NX + i;
          const double sl = slx(X.data(), k, i), su = sux(X.data(), k, i);
          sgx[k * NX + i] = (hlx[i] ? zlx[j] / sl : 0.0) + (hux[i] ? zux[j] / su : 0.0);
          qx[k * NX + i] = gx[k * NX + i] - (hlx[i] ? mu / sl : 0.0) + (hux[i] ? mu / su : 0.0);    // grad of the barrier function
        }
      for (int k = 0; k < N; ++k)
        for (int i = 0; i < NU; ++i) {
          const int j = k * NU + i;
          const double sl = slu(U.data(), k, i), su = suu(U.data(), k, i);
          sgu[j] = (hlu[i] ? zlu[j] / sl : 0.0) + (huu[i] ? zuu[j] / su : 0.0);
          qu[j] = gu[j] - (hlu[i] ? mu / sl : 0.0) + (huu[i] ? mu / su : 0.0);
        }
      double delta = 0.0;
      bool first_try = true, fail = false;
      for (;;) {
        for (size_t i = 0; i < dgx.size(); ++i) dgx[i] = sgx[i] + delta;
        for (size_t i = 0; i < dgu.size(); ++i) dgu[i] = sgu[i] + delta;
        if (riccati(Hz.data(), HN.data(), dgx.data(), dgu.data(), qx.data(), qu.data(), c.data())) break;
        if (first_try) {
          delta = delta_last == 0.0 ? dw_0 : std::max(dw_min, kw_minus * delta_last);
          first_try = false;
        } else {
          delta *= delta_last == 0.0 ? kw_plus_bar : kw_plus;
        }
        if (delta > dw_max) { fail = true; break; }
      }
      if (fail) { status = RESTORATION_FAILED; break; }
      if (delta > 0) delta_last = delta;
      double alpha_z = 1.0;
      for (int k = 1; k <= N; ++k)
        for (int i = 0; i < NX; ++i) {
          const int j = (k - 1) * NX + i;
          const double sl = slx(X.data(), k, i), su = sux(X.data(), k, i), d = dX[k * NX + i];
          dzlx[j] = hlx[i] ? mu / sl - zlx[j] - zlx[j] / sl * d : 0.0;
          dzux[j] = hux[i] ? mu / su - zux[j] + zux[j] / su * d : 0.0;
          if (hlx[i] && dzlx[j] < 0) alpha_z = std::min(alpha_z, -tau * zlx[j] / dzlx[j]);
          if (hux[i] && dzux[j] < 0) alpha_z = std::min(alpha_z, -tau * zux[j] / dzux[j]);
        }
      for (int k = 0; k < N; ++k)
        for (int i = 0; i < NU; ++i) {
          const int j = k * NU + i;
          const double sl = slu(U.data(), k, i), su = suu(U.data(), k, i), d = dU[j];
          dzlu[j] = hlu[i] ? mu / sl - zlu[j] - zlu[j] / sl * d : 0.0;
          dzuu[j] = huu[i] ? mu / su - zuu[j] + zuu[j] / su * d : 0.0;
          if (hlu[i] && dzlu[j] < 0) alpha_z = std::min(alpha_z, -tau * zlu[j] / dzlu[j]);
          if (huu[i] && dzuu[j] < 0) alpha_z = std::min(alpha_z, -tau * zuu[j] / dzuu[j]);
        }
      const double alpha_max = alpha_primal(dX.data(), dU.data(), tau);
      // ---- filter line search (W&B Alg. A) ----
      const double phi0 = barrier(f, X.data(), U.data(), mu), th0 = l1(c);
      double dphi = 0.0;
      for (int i = NX; i < (N + 1) * NX; ++i) dphi += qx[i] * dX[i];
      for (int i = 0; i < N * NU; ++i) dphi += qu[i] * dU[i];
      double alpha = alpha_max;
      bool accepted = false, armijo = false, resto = false;
      lam_step = lamn;
      for (int ls = 0; ls < 60 && !accepted; ++ls) {
        for (size_t i = 0; i < X.size(); ++i) Xt[i] = X[i] + alpha * dX[i];
        for (size_t i = 0; i < U.size(); ++i) Ut[i] = U[i] + alpha * dU[i];
        const double ft = eval_fc(Xt.data(), Ut.data(), ct.data());
        const double pht = barrier(ft, Xt.data(), Ut.data(), mu), tht = l1(ct);
        const double rnd = 10 * EPS * std::fabs(phi0);
        auto acceptable = [&](double th, double ph, bool* sw_o) {
          bool ok = std::isfinite(ph) && std::isfinite(th) && th <= theta_max && filter_ok(th, ph);
          bool sw = false;
          if (ok) {
            sw = th0 <= theta_min && dphi < 0 && alpha * std::pow(-dphi, s_phi) > delta_ls * std::pow(th0, s_theta);
            ok = sw ? ph - phi0 - rnd <= eta_phi * alpha * dphi
                    : (th <= (1 - gamma_theta) * th0 || ph - phi0 - rnd <= -gamma_phi * th0);
          }
          *sw_o = sw;
          return ok;
        };
        bool sw = false;
        bool ok = acceptable(tht, pht, &sw);
        if (!ok && ls == 0 && tht >= th0) {
          // second-order correction (W&B sec. 2.4)
          for (size_t i = 0; i < c.size(); ++i) csoc[i] = alpha * c[i] + ct[i];
          double th_old = tht;
          dXs = dX; dUs = dU; lams = lamn;
          for (int q = 0; q < max_soc; ++q) {
            if (!riccati(Hz.data(), HN.data(), dgx.data(), dgu.data(), qx.data(), qu.data(), csoc.data())) break;
            const double a_s = alpha_primal(dX.data(), dU.data(), tau);
            for (size_t i = 0; i < X.size(); ++i) Xs[i] = X[i] + a_s * dX[i];
            for (size_t i = 0; i < U.size(); ++i) Us[i] = U[i] + a_s * dU[i];
            const double fs = eval_fc(Xs.data(), Us.data(), cs.data());
            const double phs = barrier(fs, Xs.data(), Us.data(), mu), ths = l1(cs);
            bool sw2 = false;
            if (acceptable(ths, phs, &sw2)) {
              ok = true; sw = sw2;
              Xt = Xs; Ut = Us;
              lam_step = lamn;
              break;
            }
            if (!(ths <= kappa_soc * th_old)) break;
            th_old = ths;
            for (size_t i = 0; i < c.size(); ++i) csoc[i] = a_s * csoc[i] + cs[i];
          }
          dX = dXs; dU = dUs; lamn = lams;
        }
        if (ok) {
          accepted = true;
          armijo = sw;
        } else {
          alpha *= alpha_red;
          double amin = gamma_theta;     // W&B eq. 23
          if (dphi < 0) {
            amin = std::min(amin, gamma_phi * th0 / (-dphi));
            if (th0 <= theta_min) amin = std::min(amin, delta_ls * std::pow(th0, s_theta) / std::pow(-dphi, s_phi));
          }
          if (alpha < alpha_min_frac * amin) { resto = true; break; }
        }
      }
      if (!accepted && !resto) resto = true;
      if (resto) {
        filt.emplace_back((1 - gamma_theta) * th0, phi0 - gamma_phi * th0);
        const int rr = restore(mu, tau, theta_max);
        if (rr != 0) { status = rr == 2 ? INFEASIBLE : RESTORATION_FAILED; break; }
        std::fill(lam.begin(), lam.end(), 0.0);       // constr_mult_reset_threshold = 0
        double zmax = 0.0;
        for (double z : zlx) zmax = std::max(zmax, z);
        for (double z : zux) zmax = std::max(zmax, z);
        for (double z : zlu) zmax = std::max(zmax, z);
        for (double z : zuu) zmax = std::max(zmax, z);
        if (zmax > 1e3) {                             // bound_mult_reset_threshold
          for (int j = 0; j < N * NX; ++j) { zlx[j] = hlx[j % NX] ? 1.0 : 0.0; zux[j] = hux[j % NX] ? 1.0 : 0.0; }
          for (int j = 0; j < N * NU; ++j) { zlu[j] = hlu[j % NU] ? 1.0 : 0.0; zuu[j] = huu[j % NU] ? 1.0 : 0.0; }
        }
      } else {
        if (!armijo) {                                // augment the filter (W&B eq. 22)
          filt.emplace_back((1 - gamma_theta) * th0, phi0 - gamma_phi * th0);
          if ((int)filt.size() > max_filter) filt.erase(filt.begin());
        }
        X = Xt; U = Ut;
        for (size_t i = 0; i < lam.size(); ++i) lam[i] += alpha * (lam_step[i] - lam[i]);
        for (int j = 0; j < N * NX; ++j) { zlx[j] += alpha_z * dzlx[j]; zux[j] += alpha_z * dzux[j]; }
        for (int j = 0; j < N * NU; ++j) { zlu[j] += alpha_z * dzlu[j]; zuu[j] += alpha_z * dzuu[j]; }
      }
      // W&B eq. 16: keep z within [mu / (kappa s), kappa mu / s]
      for (int k = 1; k <= N; ++k)
        for (int i = 0; i < NX; ++i) {
          const int j = (k - 1) * NX + i;
          const double sl = slx(X.data(), k, i), su = sux(X.data(), k, i);
          zlx[j] = hlx[i] ? std::min(std::max(zlx[j], mu / (kappa_sigma * sl)), kappa_sigma * mu / sl) : 0.0;
          zux[j] = hux[i] ? std::min(std::max(zux[j], mu / (kappa_sigma * su)), kappa_sigma * mu / su) : 0.0;
        }
      for (int k = 0; k < N; ++k)
        for (int i = 0; i < NU; ++i) {
          const int j = k * NU + i;
          const double sl = slu(U.data(), k, i), su = suu(U.data(), k, i);
          zlu[j] = hlu[i] ? std::min(std::max(zlu[j], mu / (kappa_sigma * sl)), kappa_sigma * mu / sl) : 0.0;
          zuu[j] = huu[i] ? std::min(std::max(zuu[j], mu / (kappa_sigma * su)), kappa_sigma * mu / su) : 0.0;
        }
      ++iters;
    }
    f = eval_all(lam.data());
    const double E0 = errors(0.0);
    if (v_opt) {
      std::memcpy(v_opt, X.data(), sizeof(double) * (N + 1) * NX);
      std::memcpy(v_opt + (N + 1) * NX, U.data(), sizeof(double) * N * NU);
    }
    if (f_opt) *f_opt = f;
    if (u0)
      for (int i = 0; i < NU; ++i) u0[i] = U[i] * pb.su[i];
    *status_o = status;
    *iters_o = iters;
    if (kkt_o) *kkt_o = E0;
  }
};

template <class M>
void solve_batch(const Problem& pb, int64_t batch, const double* x0, const double* par, int64_t par_stride, const double* v0,
                 double* v_opt, double* f_opt, double* u0, int32_t* status, int32_t* iters, double* kkt, int n_threads) {
  const int nv = (pb.N + 1) * M::NX + pb.N * M::NU;
#pragma omp parallel num_threads(n_threads)
  {
    Solver<M> s(pb, nullptr);
#pragma omp for schedule(dynamic, 4)
    for (int64_t b = 0; b < batch; ++b) {
      s.p = par ? par + b * par_stride : nullptr;
      int st = 0, itc = 0;
      s.solve(x0 + b * M::NX, v0 ? v0 + b * nv : nullptr, v_opt ? v_opt + b * nv : nullptr, f_opt ? f_opt + b : nullptr,
              u0 ? u0 + b * M::NU : nullptr, &st, &itc, kkt ? kkt + b : nullptr);
      status[b] = st;
      iters[b] = itc;
    }
  }
}

template <class M>
void plant_batch(const Problem& pb, int64_t batch, const double* x, const double* u, const double* par, int64_t par_stride,
                 double* xn, int n_threads) {
  Problem q = pb;                        // original units: unit scaling
  std::fill(q.sx.begin(), q.sx.end(), 1.0);
  std::fill(q.su.begin(), q.su.end(), 1.0);
#pragma omp parallel for num_threads(n_threads)
  for (int64_t b = 0; b < batch; ++b)
    phi_scaled<M, double>(q, x + b * M::NX, u + b * M::NU, par ? par + b * par_stride : nullptr, xn + b * M::NX);
}

int fail(const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return -1;
}

}  // namespace

struct hilo_cpu_nmpc { Problem pb; };

extern "C" {

const char* hilo_cpu_last_error(void) { return g_err; }

int hilo_cpu_max_threads(void) { return omp_get_max_threads(); }

// the learned term of HILO_MODEL_CHEMOSTAT4_GP (host arrays, kept by the caller): see models_cpu.h::Chemostat4Gp
void hilo_cpu_set_gp(int n, double sf2, double bias, double M0, double M1, const double* X0, const double* X1, const double* alpha) {
  GpSe2& g = gp_of_chemostat4();
  g.n = n; g.sf2 = sf2; g.bias = bias; g.M[0] = M0; g.M[1] = M1; g.X0 = X0; g.X1 = X1; g.alpha = alpha;
}

// Same descriptor as hilo_nmpc_create (include/hilo_hip.h); the subset this baseline covers is checked here.
int hilo_cpu_nmpc_create(const hilo_nmpc_desc* d, hilo_cpu_nmpc** out) {
  if (!d || !out) return fail("NULL argument");
  int nx, nu, np;
  if (d->model_id == HILO_MODEL_CHEMOSTAT4 || d->model_id == HILO_MODEL_CHEMOSTAT4_GP) { nx = 4; nu = 2; np = 4; }
  else if (d->model_id == HILO_MODEL_PENDULUM4) { nx = 4; nu = 1; np = 0; }
  else return fail("the CPU baseline holds the models chemostat4 (with or without the learned growth rate) and pendulum4 only");
  if (d->model_id == HILO_MODEL_CHEMOSTAT4_GP && gp_of_chemostat4().n <= 0) return fail("hilo_cpu_set_gp first");
  if (d->N < 1 || d->dt <= 0) return fail("bad horizon / dt");
  if ((d->Nc && d->Nc != d->N) || d->n_path_var || d->n_con || d->n_tcon || d->collocation_degree || d->time_varying ||
      d->Wdu || d->user_source)     // (d->learned is a DEVICE handle: the learned term comes through hilo_cpu_set_gp)
    return fail("the CPU baseline covers tracking NMPC with box bounds only");
  hilo_cpu_nmpc* h = new hilo_cpu_nmpc();
  Problem& p = h->pb;
  p.model_id = d->model_id; p.N = d->N; p.nx = nx; p.nu = nu; p.np = np;
  p.order = d->erk_order >= 1 ? d->erk_order : 4;
  p.n_sub = d->n_sub >= 1 ? d->n_sub : 1;
  if (p.order > 4) { delete h; return fail("explicit Runge-Kutta order 1..4"); }
  p.max_iter = d->max_iter > 0 ? d->max_iter : 3000;
  p.acceptable_iter = d->acceptable_iter > 0 ? d->acceptable_iter : 15;
  p.dt = d->dt;
  p.tol = d->tol > 0 ? d->tol : 1e-8;
  p.acceptable_tol = d->acceptable_tol > 0 ? d->acceptable_tol : 1e-6;
  p.mu_init = d->mu_init > 0 ? d->mu_init : 0.1;
  p.relax = d->bound_relax_factor < 0 ? 1e-8 : d->bound_relax_factor;
  const int nz = nx + nu;
  auto cp = [](std::vector<double>& v, const double* s, int n, double dflt) { v.resize(n); for (int i = 0; i < n; ++i) v[i] = s ? s[i] : dflt; };
  cp(p.Wz, d->Wz, nz * nz, 0.0); cp(p.zref, d->zref, nz, 0.0); cp(p.WN, d->WN, nx * nx, 0.0); cp(p.xrefN, d->xrefN, nx, 0.0);
  cp(p.sx, d->x_scaling, nx, 1.0); cp(p.su, d->u_scaling, nu, 1.0);
  cp(p.xlb, d->x_lb, nx, -INF); cp(p.xub, d->x_ub, nx, INF); cp(p.ulb, d->u_lb, nu, -INF); cp(p.uub, d->u_ub, nu, INF);
  cp(p.xg, d->x_guess, nx, 0.0); cp(p.ug, d->u_guess, nu, 0.0);
  for (int i = 0; i < nx; ++i) { p.xlb[i] /= p.sx[i]; p.xub[i] /= p.sx[i]; p.xg[i] /= p.sx[i]; }     // mpc.py:248-263
  for (int i = 0; i < nu; ++i) { p.ulb[i] /= p.su[i]; p.uub[i] /= p.su[i]; p.ug[i] /= p.su[i]; }
  *out = h;
  return 0;
}

void hilo_cpu_nmpc_destroy(hilo_cpu_nmpc* h) { delete h; }

// HOST pointers throughout.  x0 [batch][nx] original units; par [batch][par_stride] or NULL; v0 [batch][n_v] scaled warm start or
// NULL (the guess of the descriptor); outputs like hilo_nmpc_solve: v_opt [batch][n_v] scaled, f_opt, first input (original
// units), status (optimizer.py:1085-1104), iteration count, scaled KKT error.  n_threads <= 0: all cores.
int hilo_cpu_nmpc_solve(hilo_cpu_nmpc* h, int64_t batch, const double* x0, const double* par, int64_t par_stride, const double* v0,
                        double* v_opt, double* f_opt, double* first_u, int32_t* status, int32_t* iters, double* kkt, int n_threads) {
  if (!h || !x0 || !status || !iters) return fail("NULL argument");
  if (h->pb.np > 0 && !par) return fail("the model has parameters: par is required");
  if (n_threads <= 0) n_threads = omp_get_max_threads();
  if (h->pb.model_id == HILO_MODEL_CHEMOSTAT4)
    solve_batch<Chemostat4>(h->pb, batch, x0, par, par_stride, v0, v_opt, f_opt, first_u, status, iters, kkt, n_threads);
  else if (h->pb.model_id == HILO_MODEL_CHEMOSTAT4_GP)
    solve_batch<Chemostat4Gp>(h->pb, batch, x0, par, par_stride, v0, v_opt, f_opt, first_u, status, iters, kkt, n_threads);
  else
    solve_batch<Pendulum4>(h->pb, batch, x0, par, par_stride, v0, v_opt, f_opt, first_u, status, iters, kkt, n_threads);
  return 0;
}

// x+ = Phi(x, u, p) in original units (the plant of the closed loop)
int hilo_cpu_plant_step(hilo_cpu_nmpc* h, int64_t batch, const double* x, const double* u, const double* par, int64_t par_stride,
                        double* xn, int n_threads) {
  if (!h || !x || !u || !xn) return fail("NULL argument");
  if (n_threads <= 0) n_threads = omp_get_max_threads();
  if (h->pb.model_id == HILO_MODEL_CHEMOSTAT4) plant_batch<Chemostat4>(h->pb, batch, x, u, par, par_stride, xn, n_threads);
  else if (h->pb.model_id == HILO_MODEL_CHEMOSTAT4_GP) plant_batch<Chemostat4Gp>(h->pb, batch, x, u, par, par_stride, xn, n_threads);
  else plant_batch<Pendulum4>(h->pb, batch, x, u, par, par_stride, xn, n_threads);
  return 0;
}

}  // extern "C"
