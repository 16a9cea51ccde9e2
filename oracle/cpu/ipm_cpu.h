// CPU baseline: second-order forward mode and the stage-structured interior-point solver shared by the optimisation legs
// (nmpc_cpu.cpp: tracking NMPC, mhe_cpu.cpp: moving-horizon estimation).
// TEST INFRASTRUCTURE / BASELINE ONLY - see nmpc_cpu.cpp.
//
// StageIpm<Pol> is oracle/nmpc.py::DenseIpm statement by statement (Waechter & Biegler 2006 with IPOPT's default constants:
// monotone barrier update, fraction to the boundary, filter line search with second-order correction, inertia correction,
// barrier-augmented feasibility restoration) for problems of the form
//     min  sum_k l_k(x_k, u_k) + V(x_N)   s.t.  x_{k+1} = F_k(x_k, u_k),  dl <= d(x_k, u_k) <= du,  box bounds on x_k and u_k,
// the entries of x_0 pinned (NMPC, mpc.py:797-802) or variables (MHE, mhe.py:614-655; the path variable and the slack state of
// the general NMPC) by the mask `pol.free0`, with the dense KKT solve replaced by the Riccati recursion over the stages - the
// inertia of the KKT matrix is correct exactly when every stage's reduced input Hessian F_k = R_k + B_k' P_{k+1} B_k (and the
// block of P_0 over the free entries of x_0) has a Cholesky factor.  The inequality rows are handled the way IPOPT does (W&B
// sec. 3.4 of the implementation paper, like oracle/nmpc_gen.py::GenIpm): a slack s with d - s = 0 and bounds dl <= s <= du,
// s_0 = d(w_0) pushed into the interior; the slacks and their multipliers are eliminated stage by stage before the recursion
// ((H + Jd' D Jd) dz = ..., D = Sigma_s + delta) and recovered behind it.
// The policy `Pol` supplies the problem functions:
//   static constexpr int NX, NU, NR (inequality rows per stage, 0 = none);  bool free0[NX];
//   double stage_fc(int k, const double* x, const double* u, double* F) const;       value of l_k, F = F_k(x, u)
//   double stage_all(int k, const double* x, const double* u, const double* lam_k,    + gradient gz [NZ] of l_k, Jacobians
//                    double* gz, double* H, double* F, double* A, double* B) const;    A [NX][NX], B [NX][NU] of F_k and
//                                                                                     H [NZ][NZ] = hess l_k - sum_r lam_kr hess F_kr
//   double term_fc(const double* xN) const;  double term_all(const double* xN, double* gN, double* HN) const;
// and with NR > 0
//   void rows_fc(int k, const double* x, const double* u, double* d) const;
//   void rows_all(int k, const double* x, const double* u, const double* lam_d, double* d, double* Jd, double* H) const;
//                                                                  Jd [NR][NZ], H += sum_r lam_dr hess d_r
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

namespace hilo_cpu {

constexpr double INF = std::numeric_limits<double>::infinity();
constexpr double EPS = std::numeric_limits<double>::epsilon();
enum { SOLVED = 1, ACCEPTABLE = 2, INFEASIBLE = 3, RESTORATION_FAILED = 4, MAXITER = 5, OTHER = -1 };   // optimizer.py:1085-1104

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

struct IpmOptions {
  int max_iter = 3000, acceptable_iter = 15;
  double tol = 1e-8, acceptable_tol = 1e-6, mu_init = 0.1, relax = 1e-8;
};

// ---- one instance ---------------------------------------------------------------------------------------------------------------
template <class Pol>
struct StageIpm {
  static constexpr int NX = Pol::NX, NU = Pol::NU, NZ = NX + NU;
  static constexpr int NR = Pol::NR, NRA = NR > 0 ? NR : 1;
  Pol& pol;
  const IpmOptions o;
  const int N;
  // iterate: X[k] k = 0..N, U[k], slacks S[k] of the rows; multipliers of the bounds (index k NX + i for x: zero for a pinned
  // entry of x_0), of the defects (lam) and of the rows (lamd)
  std::vector<double> X, U, S, lam, lamd, zlx, zux, zlu, zuu, zls, zus;
  bool fr0[NX];                                        // entry i of x_0 is a variable
  int nfree0 = 0;
  // bounds (relaxed), shared by all stages
  double lbx[NX], ubx[NX], lbu[NU], ubu[NU], lbs[NRA], ubs[NRA];
  bool hlx[NX], hux[NX], hlu[NU], huu[NU], hls[NRA], hus[NRA];
  int nb, m;
  // derivative buffers
  std::vector<double> gx, gu, c, A, Bm, Hz, HN;    // gx [N+1][NX], gu [N][NU], c [N][NX], A [N][NX][NX], B [N][NX][NU]
  std::vector<double> cd, Jd, He, HNe, qxe, que;   // rows: cd [N][NR] = d - s, Jd [N][NR][NZ]; condensed Hessian / gradient
  // Riccati
  std::vector<double> P, pv, K, kff, dX, dU, dS, lamn, lamdn;
  std::vector<std::pair<double, double>> filt;
  // work vectors of solve() / restore(): allocated once per solver object (one per thread), not per instance - with many
  // threads the allocator otherwise serialises the batch
  std::vector<double> ct, Xt, Ut, qx, qu, sgx, sgu, dgx, dgu, dzlx, dzux, dzlu, dzuu, csoc, dXs, dUs, lams, lam_step, Xs, Us, cs,
      r_lam0, r_qx, r_qu, r_sgx, r_sgu, r_Xt, r_Ut, r_ct;
  std::vector<double> cdt, St, qs, sgs, dgs, dzls, dzus, cdsoc, dSs, lamds, lamd_step, Ss, cds, r_lamd0, r_qs, r_sgs, r_St, r_cdt;

  // bounds in the scaled variables, +-INF = none
  // dlb / dub: bounds of the rows (NULL with NR = 0)
  StageIpm(Pol& pol_, const IpmOptions& o_, int N_, const double* xlb, const double* xub, const double* ulb, const double* uub,
           const double* dlb = nullptr, const double* dub = nullptr)
      : pol(pol_), o(o_), N(N_) {
    for (int i = 0; i < NX; ++i) { fr0[i] = pol.free0[i]; nfree0 += (int)fr0[i]; }
    X.resize((N + 1) * NX); U.resize(N * NU); lam.assign(N * NX, 0.0);
    zlx.assign((N + 1) * NX, 0.0); zux.assign((N + 1) * NX, 0.0); zlu.resize(N * NU); zuu.resize(N * NU);
    gx.resize((N + 1) * NX); gu.resize(N * NU); c.resize(N * NX); A.resize(N * NX * NX); Bm.resize(N * NX * NU);
    Hz.resize(N * NZ * NZ); HN.resize(NX * NX);
    P.resize((N + 1) * NX * NX); pv.resize((N + 1) * NX); K.resize(N * NU * NX); kff.resize(N * NU);
    dX.resize((N + 1) * NX); dU.resize(N * NU); lamn.resize(N * NX);
    ct.resize(c.size()); Xt.resize(X.size()); Ut.resize(U.size()); qx.assign((N + 1) * NX, 0.0); qu.resize(N * NU);
    sgx.assign((N + 1) * NX, 0.0); sgu.resize(N * NU); dgx.resize((N + 1) * NX); dgu.resize(N * NU);
    dzlx.assign((N + 1) * NX, 0.0); dzux.assign((N + 1) * NX, 0.0); dzlu.resize(N * NU); dzuu.resize(N * NU); csoc.resize(c.size());
    dXs.resize(dX.size()); dUs.resize(dU.size()); lams.resize(lamn.size()); lam_step.resize(lamn.size());
    Xs.resize(X.size()); Us.resize(U.size()); cs.resize(c.size());
    r_lam0.assign(N * NX, 0.0); r_qx.assign((N + 1) * NX, 0.0); r_qu.resize(N * NU); r_sgx.assign((N + 1) * NX, 0.0); r_sgu.resize(N * NU);
    r_Xt.resize(X.size()); r_Ut.resize(U.size()); r_ct.resize(c.size());
    const double r = o.relax;
    nb = 0;
    for (int i = 0; i < NX; ++i) {
      hlx[i] = std::isfinite(xlb[i]); hux[i] = std::isfinite(xub[i]);
      lbx[i] = hlx[i] ? xlb[i] - r * std::max(1.0, std::fabs(xlb[i])) : -INF;
      ubx[i] = hux[i] ? xub[i] + r * std::max(1.0, std::fabs(xub[i])) : INF;
      nb += (N + (int)fr0[i]) * ((int)hlx[i] + (int)hux[i]);
    }
    for (int i = 0; i < NU; ++i) {
      hlu[i] = std::isfinite(ulb[i]); huu[i] = std::isfinite(uub[i]);
      lbu[i] = hlu[i] ? ulb[i] - r * std::max(1.0, std::fabs(ulb[i])) : -INF;
      ubu[i] = huu[i] ? uub[i] + r * std::max(1.0, std::fabs(uub[i])) : INF;
      nb += N * ((int)hlu[i] + (int)huu[i]);
    }
    if constexpr (NR > 0) {
      const size_t ns = (size_t)N * NR;
      S.resize(ns); lamd.assign(ns, 0.0); zls.resize(ns); zus.resize(ns); cd.resize(ns); Jd.resize(ns * NZ);
      He.resize(N * NZ * NZ); HNe.resize(NX * NX); qxe.resize((N + 1) * NX); que.resize(N * NU);
      dS.resize(ns); lamdn.resize(ns); cdt.resize(ns); St.resize(ns); qs.resize(ns); sgs.resize(ns); dgs.resize(ns);
      dzls.resize(ns); dzus.resize(ns); cdsoc.resize(ns); dSs.resize(ns); lamds.resize(ns); lamd_step.resize(ns); Ss.resize(ns);
      cds.resize(ns); r_lamd0.assign(ns, 0.0); r_qs.resize(ns); r_sgs.resize(ns); r_St.resize(ns); r_cdt.resize(ns);
      for (int i = 0; i < NR; ++i) {
        hls[i] = std::isfinite(dlb[i]); hus[i] = std::isfinite(dub[i]);
        lbs[i] = hls[i] ? dlb[i] - r * std::max(1.0, std::fabs(dlb[i])) : -INF;
        ubs[i] = hus[i] ? dub[i] + r * std::max(1.0, std::fabs(dub[i])) : INF;
        nb += N * ((int)hls[i] + (int)hus[i]);
      }
    }
    nb = std::max(1, nb);
    m = N * NX + N * NR;
  }

  bool var(int k, int i) const { return k > 0 || fr0[i]; }

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
  double eval_fc(const double* Xc, const double* Uc, double* cc, const double* Sc = nullptr, double* ccd = nullptr) const {
    double f = 0.0, ph[NX];
    for (int k = 0; k < N; ++k) {
      f += pol.stage_fc(k, Xc + k * NX, Uc + k * NU, ph);
      for (int i = 0; i < NX; ++i) cc[k * NX + i] = Xc[(k + 1) * NX + i] - ph[i];
      if constexpr (NR > 0) {
        double d[NRA];
        pol.rows_fc(k, Xc + k * NX, Uc + k * NU, d);
        for (int r = 0; r < NR; ++r) ccd[k * NR + r] = d[r] - Sc[k * NR + r];
      }
    }
    return f + pol.term_fc(Xc + N * NX);
  }

  // f, gradient, defects, stage Jacobians and the Hessian of the Lagrangian by stages (DenseIpm.eval_all)
  double eval_all(const double* lm, const double* lmd = nullptr) {
    double f = 0.0, gz[NZ], ph[NX], gN[NX];
    std::fill(gx.begin(), gx.end(), 0.0);
    for (int k = 0; k < N; ++k) {
      f += pol.stage_all(k, &X[k * NX], &U[k * NU], lm + k * NX, gz, &Hz[k * NZ * NZ], ph, &A[k * NX * NX], &Bm[k * NX * NU]);
      for (int i = 0; i < NX; ++i) gx[k * NX + i] += gz[i];
      for (int i = 0; i < NU; ++i) gu[k * NU + i] = gz[NX + i];
      for (int r = 0; r < NX; ++r) c[k * NX + r] = X[(k + 1) * NX + r] - ph[r];
      if constexpr (NR > 0) {
        double d[NRA];
        pol.rows_all(k, &X[k * NX], &U[k * NU], lmd + k * NR, d, &Jd[(size_t)k * NR * NZ], &Hz[k * NZ * NZ]);
        for (int r = 0; r < NR; ++r) cd[k * NR + r] = d[r] - S[k * NR + r];
      }
    }
    f += pol.term_all(&X[N * NX], gN, HN.data());
    for (int i = 0; i < NX; ++i) gx[N * NX + i] += gN[i];
    return f;
  }

  // slacks of variable (x_k component i) / (u_k component i) / (row slack s_k component i)
  double slx(const double* Xc, int k, int i) const { return hlx[i] ? Xc[k * NX + i] - lbx[i] : 1.0; }
  double sux(const double* Xc, int k, int i) const { return hux[i] ? ubx[i] - Xc[k * NX + i] : 1.0; }
  double slu(const double* Uc, int k, int i) const { return hlu[i] ? Uc[k * NU + i] - lbu[i] : 1.0; }
  double suu(const double* Uc, int k, int i) const { return huu[i] ? ubu[i] - Uc[k * NU + i] : 1.0; }
  double sls(const double* Sc, int k, int i) const { return hls[i] ? Sc[k * NR + i] - lbs[i] : 1.0; }
  double sus(const double* Sc, int k, int i) const { return hus[i] ? ubs[i] - Sc[k * NR + i] : 1.0; }

  double barrier(double f, const double* Xc, const double* Uc, double mu, const double* Sc = nullptr) const {
    double s = 0.0;
    if constexpr (NR > 0)
      for (int k = 0; k < N; ++k)
        for (int i = 0; i < NR; ++i) {
          if (hls[i]) s += std::log(Sc[k * NR + i] - lbs[i]);
          if (hus[i]) s += std::log(ubs[i] - Sc[k * NR + i]);
        }
    for (int k = 0; k <= N; ++k)
      for (int i = 0; i < NX; ++i) {
        if (!var(k, i)) continue;
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
  double errors(double mu) const {
    double dual = 0.0, cmp = 0.0, zsum = 0.0, lsum = l1(lam), prim = linf(c);
    if constexpr (NR > 0) {
      for (int k = 0; k < N; ++k)
        for (int i = 0; i < NR; ++i) {
          const int j = k * NR + i;
          dual = std::max(dual, std::fabs(-lamd[j] - zls[j] + zus[j]));
          if (hls[i]) cmp = std::max(cmp, std::fabs(sls(S.data(), k, i) * zls[j] - mu));
          if (hus[i]) cmp = std::max(cmp, std::fabs(sus(S.data(), k, i) * zus[j] - mu));
          zsum += std::fabs(zls[j]) + std::fabs(zus[j]);
        }
      lsum += l1(lamd);
      prim = std::max(prim, linf(cd));
    }
    for (int k = 0; k <= N; ++k)
      for (int i = 0; i < NX; ++i) {
        if (!var(k, i)) continue;
        double r = gx[k * NX + i];
        if (k > 0) r += lam[(k - 1) * NX + i];
        if (k < N) {
          for (int q = 0; q < NX; ++q) r -= A[(k * NX + q) * NX + i] * lam[k * NX + q];
          if constexpr (NR > 0)
            for (int q = 0; q < NR; ++q) r += Jd[((size_t)k * NR + q) * NZ + i] * lamd[k * NR + q];
        }
        const int j = k * NX + i;
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
        if constexpr (NR > 0)
          for (int q = 0; q < NR; ++q) r += Jd[((size_t)k * NR + q) * NZ + NX + i] * lamd[k * NR + q];
        const int j = k * NU + i;
        r += -zlu[j] + zuu[j];
        dual = std::max(dual, std::fabs(r));
        if (hlu[i]) cmp = std::max(cmp, std::fabs(slu(U.data(), k, i) * zlu[j] - mu));
        if (huu[i]) cmp = std::max(cmp, std::fabs(suu(U.data(), k, i) * zuu[j] - mu));
        zsum += std::fabs(zlu[j]) + std::fabs(zuu[j]);
      }
    const double smax = 100.0;
    const double s_d = std::max(smax, (lsum + zsum) / (m + nb)) / smax, s_c = std::max(smax, zsum / nb) / smax;
    return std::max(std::max(dual / s_d, prim), cmp / s_c);
  }

  // Riccati solve of  [H + D, J'; J, 0] [d; lam+] = [-q; -cc]  with H = blkdiag(Hs_k) (+ HNs), D = diag(dgx, dgu), q = (qx, qu).
  // Returns false when a stage's reduced input Hessian (or, with x_0 free, the cost-to-go of the first stage) is not positive
  // definite (wrong inertia).  Hs == nullptr: identity.
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
      if (k == 0 && nfree0 == 0) break;   // dx_0 = 0: no gain, no cost-to-go needed
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
    if (nfree0 > 0) {                   // free entries of x_0: P_0[ff] dx_0[f] = -p_0[f]
      int id[NX], nf = 0;
      for (int i = 0; i < NX; ++i)
        if (fr0[i]) id[nf++] = i;
      double L[NX][NX], y[NX];
      for (int i = 0; i < nf; ++i)
        for (int j = 0; j <= i; ++j) {
          double t = P[id[i] * NX + id[j]];
          for (int q = 0; q < j; ++q) t -= L[i][q] * L[j][q];
          if (i == j) {
            if (!(t > 0.0) || !std::isfinite(t)) return false;
            L[i][i] = std::sqrt(t);
          } else {
            L[i][j] = t / L[j][j];
          }
        }
      for (int i = 0; i < nf; ++i) { double t = -pv[id[i]]; for (int q = 0; q < i; ++q) t -= L[i][q] * y[q]; y[i] = t / L[i][i]; }
      for (int i = nf - 1; i >= 0; --i) { double t = y[i]; for (int q = i + 1; q < nf; ++q) t -= L[q][i] * y[q]; y[i] = t / L[i][i]; }
      for (int i = 0; i < nf; ++i) dX[id[i]] = y[i];
    }
    for (int k = 0; k < N; ++k) {
      for (int i = 0; i < NU; ++i) {
        double t = kff[k * NU + i];
        if (k > 0 || nfree0 > 0)
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

  // search direction with the rows: their slacks and multipliers are eliminated around the Riccati recursion.  dgs = Sigma_s
  // (+ delta), qs = gradient of the barrier function in s, ccd = right-hand side of the rows; Hs == nullptr: identity Hessian
  // (restoration), on the slacks too.  Leaves dS and lamdn next to dX / dU / lamn.
  bool kkt_solve(const double* Hs, const double* HNs, const double* dgx, const double* dgu, const double* dgs_, const double* qx,
                 const double* qu, const double* qs_, const double* cc, const double* ccd) {
    if constexpr (NR == 0) {
      return riccati(Hs, HNs, dgx, dgu, qx, qu, cc);
    } else {
      const double hs = Hs ? 0.0 : 1.0;
      for (int k = 0; k < N; ++k) {
        double* H = &He[k * NZ * NZ];
        for (int i = 0; i < NZ; ++i)
          for (int j = 0; j < NZ; ++j) H[i * NZ + j] = Hs ? Hs[k * NZ * NZ + i * NZ + j] : (i == j ? 1.0 : 0.0);
        double gz[NZ];
        for (int i = 0; i < NZ; ++i) gz[i] = 0.0;
        for (int r = 0; r < NR; ++r) {
          const double* J = &Jd[((size_t)k * NR + r) * NZ];
          const double D = dgs_[k * NR + r] + hs, t = D * ccd[k * NR + r] + qs_[k * NR + r];
          for (int i = 0; i < NZ; ++i) {
            gz[i] += J[i] * t;
            const double DJ = D * J[i];
            for (int j = 0; j < NZ; ++j) H[i * NZ + j] += DJ * J[j];
          }
        }
        for (int i = 0; i < NX; ++i) qxe[k * NX + i] = qx[k * NX + i] + gz[i];
        for (int i = 0; i < NU; ++i) que[k * NU + i] = qu[k * NU + i] + gz[NX + i];
      }
      for (int i = 0; i < NX; ++i) qxe[N * NX + i] = qx[N * NX + i];
      for (int i = 0; i < NX; ++i)
        for (int j = 0; j < NX; ++j) HNe[i * NX + j] = HNs ? HNs[i * NX + j] : (Hs ? 0.0 : (i == j ? 1.0 : 0.0));
      if (!riccati(He.data(), HNe.data(), dgx, dgu, qxe.data(), que.data(), cc)) return false;
      for (int k = 0; k < N; ++k)
        for (int r = 0; r < NR; ++r) {
          const double* J = &Jd[((size_t)k * NR + r) * NZ];
          double t = ccd[k * NR + r];
          for (int i = 0; i < NX; ++i) t += J[i] * dX[k * NX + i];
          for (int i = 0; i < NU; ++i) t += J[NX + i] * dU[k * NU + i];
          dS[k * NR + r] = t;
          lamdn[k * NR + r] = (dgs_[k * NR + r] + hs) * t + qs_[k * NR + r];
        }
      return true;
    }
  }

  // largest step keeping the bounded variables inside the fraction-to-the-boundary rule (W&B eq. 8)
  double alpha_primal(const double* dXc, const double* dUc, double tau, const double* dSc = nullptr) const {
    double a = 1.0;
    if constexpr (NR > 0)
      for (int k = 0; k < N; ++k)
        for (int i = 0; i < NR; ++i) {
          const double d = dSc[k * NR + i];
          if (hls[i] && d < 0) a = std::min(a, -tau * (S[k * NR + i] - lbs[i]) / d);
          if (hus[i] && d > 0) a = std::min(a, tau * (ubs[i] - S[k * NR + i]) / d);
        }
    for (int k = 0; k <= N; ++k)
      for (int i = 0; i < NX; ++i) {
        if (!var(k, i)) continue;
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

  double theta_of(const std::vector<double>& cc, const std::vector<double>& ccd) const {
    if constexpr (NR > 0) return l1(cc) + l1(ccd);
    else return l1(cc);
  }
  double cinf() const {
    if constexpr (NR > 0) return std::max(linf(c), linf(cd));
    else return linf(c);
  }

  // feasibility restoration (DenseIpm._restore): 0 = new point in X/U/S, 1 = failed, 2 = locally infeasible
  int restore(double mu, double tau, double theta_max) {
    std::vector<double>&lam0 = r_lam0, &qx = r_qx, &qu = r_qu, &sgx = r_sgx, &sgu = r_sgu, &Xt = r_Xt, &Ut = r_Ut, &ct = r_ct;
    std::vector<double>&lamd0 = r_lamd0, &qs = r_qs, &sgs = r_sgs, &St = r_St, &cdt = r_cdt;
    std::fill(lam0.begin(), lam0.end(), 0.0);
    std::fill(lamd0.begin(), lamd0.end(), 0.0);
    std::fill(sgx.begin(), sgx.end(), 0.0);
    std::fill(qx.begin(), qx.end(), 0.0);
    eval_all(lam0.data(), lamd0.data());
    const double th_start = theta_of(c, cd);
    double th = th_start, th_ref = th;
    for (int it = 0; it < 50; ++it) {
      if (it % 10 == 9) {
        if (th > (1 - 1e-4) * th_ref && th > 1e-6) return 2;
        th_ref = th;
      }
      const double mu_r = std::max(mu, cinf());
      for (int k = 0; k <= N; ++k)
        for (int i = 0; i < NX; ++i) {
          if (!var(k, i)) continue;
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
      if constexpr (NR > 0)
        for (int k = 0; k < N; ++k)
          for (int i = 0; i < NR; ++i) {
            const double sl = sls(S.data(), k, i), su = sus(S.data(), k, i);
            sgs[k * NR + i] = (hls[i] ? mu_r / (sl * sl) : 0.0) + (hus[i] ? mu_r / (su * su) : 0.0);
            qs[k * NR + i] = -(hls[i] ? mu_r / sl : 0.0) + (hus[i] ? mu_r / su : 0.0);
          }
      if (!kkt_solve(nullptr, nullptr, sgx.data(), sgu.data(), sgs.data(), qx.data(), qu.data(), qs.data(), c.data(), cd.data())) return 1;
      double dmax = 0.0;
      for (double d : dX) dmax = std::max(dmax, std::fabs(d));
      for (double d : dU) dmax = std::max(dmax, std::fabs(d));
      for (double d : dS) dmax = std::max(dmax, std::fabs(d));
      if (dmax <= 1e-9 && th > 1e-6) return 2;
      double alpha = alpha_primal(dX.data(), dU.data(), tau, dS.data()), ft = 0.0, tht = 0.0;
      bool ok = false;
      while (alpha > 1e-10) {
        for (size_t i = 0; i < X.size(); ++i) Xt[i] = X[i] + alpha * dX[i];
        for (size_t i = 0; i < U.size(); ++i) Ut[i] = U[i] + alpha * dU[i];
        for (size_t i = 0; i < S.size(); ++i) St[i] = S[i] + alpha * dS[i];
        ft = eval_fc(Xt.data(), Ut.data(), ct.data(), St.data(), cdt.data());
        tht = theta_of(ct, cdt);
        if (std::isfinite(tht) && tht <= (1 - 1e-4 * alpha) * th) { ok = true; break; }
        alpha *= 0.5;
      }
      if (!ok) return th > 1e-6 ? 2 : 1;
      X = Xt; U = Ut; S = St; th = tht;
      if (th <= 0.9 * th_start && th <= theta_max) {
        const double ph = barrier(ft, X.data(), U.data(), mu, S.data());
        bool acc = true;
        for (const auto& e : filt)
          if (th >= e.first && ph >= e.second) { acc = false; break; }
        if (acc) return 0;
      }
      eval_all(lam0.data(), lamd0.data());
    }
    return 1;
  }

  void reset_bound_multipliers() {
    for (int k = 0; k <= N; ++k)
      for (int i = 0; i < NX; ++i) {
        zlx[k * NX + i] = var(k, i) && hlx[i] ? 1.0 : 0.0;
        zux[k * NX + i] = var(k, i) && hux[i] ? 1.0 : 0.0;
      }
    for (int j = 0; j < N * NU; ++j) { zlu[j] = hlu[j % NU] ? 1.0 : 0.0; zuu[j] = huu[j % NU] ? 1.0 : 0.0; }
    if constexpr (NR > 0)
      for (int j = 0; j < N * NR; ++j) { zls[j] = hls[j % NR] ? 1.0 : 0.0; zus[j] = hus[j % NR] ? 1.0 : 0.0; }
  }

  // DenseIpm.solve_data for one instance.  Starting point: X0 [N+1][NX] (the pinned entries of X0[0] are the given state),
  // U0 [N][NU], in the scaled variables; the variables are pushed into the bounds here, the slacks of the rows start at d(w_0)
  // pushed into theirs (like IPOPT).  The solution stays in X / U / S / lam / lamd.
  void solve(const double* X0, const double* U0, double* f_opt, int* status_o, int* iters_o, double* kkt_o) {
    const double kappa_eps = 10., kappa_mu = 0.2, theta_mu = 1.5, tau_min = 0.99, kappa_sigma = 1e10;
    const double gamma_theta = 1e-5, gamma_phi = 1e-8, delta_ls = 1., s_theta = 1.1, s_phi = 2.3, eta_phi = 1e-8;
    const double alpha_red = 0.5, alpha_min_frac = 0.05, kappa_soc = 0.99;
    const double dw_min = 1e-20, dw_0 = 1e-4, dw_max = 1e40, kw_minus = 1. / 3, kw_plus = 8., kw_plus_bar = 100.;
    const int max_filter = 16, max_soc = 4;
    const double mu_floor = std::min(o.tol, 1e-4) / (kappa_eps + 1.);
    for (int k = 0; k <= N; ++k)
      for (int i = 0; i < NX; ++i)
        X[k * NX + i] = var(k, i) ? push(X0[k * NX + i], lbx[i], ubx[i], hlx[i], hux[i]) : X0[k * NX + i];
    for (int k = 0; k < N; ++k)
      for (int i = 0; i < NU; ++i) U[k * NU + i] = push(U0[k * NU + i], lbu[i], ubu[i], hlu[i], huu[i]);
    if constexpr (NR > 0)
      for (int k = 0; k < N; ++k) {
        double d[NRA];
        pol.rows_fc(k, &X[k * NX], &U[k * NU], d);
        for (int i = 0; i < NR; ++i) S[k * NR + i] = push(d[i], lbs[i], ubs[i], hls[i], hus[i]);
      }
    std::fill(lam.begin(), lam.end(), 0.0);
    std::fill(lamd.begin(), lamd.end(), 0.0);
    reset_bound_multipliers();
    double mu = o.mu_init, tau = std::max(tau_min, 1 - mu), delta_last = 0.0;
    int status = 0, iters = 0, acc_count = 0;
    filt.clear();
    std::fill(sgx.begin(), sgx.end(), 0.0);
    std::fill(qx.begin(), qx.end(), 0.0);
    double f = eval_fc(X.data(), U.data(), ct.data(), S.data(), cdt.data());
    const double theta0 = theta_of(ct, cdt);
    const double theta_min = 1e-4 * std::max(1.0, theta0), theta_max = 1e4 * std::max(1.0, theta0);

    for (int it = 0; it <= o.max_iter; ++it) {
      f = eval_all(lam.data(), lamd.data());
      const double E0 = errors(0.0);
      if (!std::isfinite(E0)) { status = OTHER; break; }
      if (E0 <= o.tol) { status = SOLVED; break; }
      acc_count = E0 <= o.acceptable_tol ? acc_count + 1 : 0;
      if (acc_count >= o.acceptable_iter) { status = ACCEPTABLE; break; }
      if (it == o.max_iter) { status = MAXITER; break; }
      // ---- barrier parameter (monotone, W&B eq. 7; floor = IPOPT's min(tol, compl_inf_tol) / (kappa_eps + 1)) ----
      for (int rep = 0; rep < 20; ++rep) {
        if (!(errors(mu) <= kappa_eps * mu && mu > mu_floor * (1 + 1e-12))) break;
        mu = std::max(mu_floor, std::min(kappa_mu * mu, std::pow(mu, theta_mu)));
        tau = std::max(tau_min, 1 - mu);
        filt.clear();
      }
      // ---- search direction with inertia correction (W&B Alg. IC) ----
      for (int k = 0; k <= N; ++k)
        for (int i = 0; i < NX; ++i) {
          if (!var(k, i)) continue;
          const int j = k * NX + i;
          const double sl = slx(X.data(), k, i), su = sux(X.data(), k, i);
          sgx[j] = (hlx[i] ? zlx[j] / sl : 0.0) + (hux[i] ? zux[j] / su : 0.0);
          qx[j] = gx[j] - (hlx[i] ? mu / sl : 0.0) + (hux[i] ? mu / su : 0.0);    // grad of the barrier function
        }
      for (int k = 0; k < N; ++k)
        for (int i = 0; i < NU; ++i) {
          const int j = k * NU + i;
          const double sl = slu(U.data(), k, i), su = suu(U.data(), k, i);
          sgu[j] = (hlu[i] ? zlu[j] / sl : 0.0) + (huu[i] ? zuu[j] / su : 0.0);
          qu[j] = gu[j] - (hlu[i] ? mu / sl : 0.0) + (huu[i] ? mu / su : 0.0);
        }
      if constexpr (NR > 0)
        for (int k = 0; k < N; ++k)
          for (int i = 0; i < NR; ++i) {
            const int j = k * NR + i;
            const double sl = sls(S.data(), k, i), su = sus(S.data(), k, i);
            sgs[j] = (hls[i] ? zls[j] / sl : 0.0) + (hus[i] ? zus[j] / su : 0.0);
            qs[j] = -(hls[i] ? mu / sl : 0.0) + (hus[i] ? mu / su : 0.0);
          }
      double delta = 0.0;
      bool first_try = true, fail = false;
      for (;;) {
        for (size_t i = 0; i < dgx.size(); ++i) dgx[i] = sgx[i] + delta;
        for (size_t i = 0; i < dgu.size(); ++i) dgu[i] = sgu[i] + delta;
        for (size_t i = 0; i < dgs.size(); ++i) dgs[i] = sgs[i] + delta;
        if (kkt_solve(Hz.data(), HN.data(), dgx.data(), dgu.data(), dgs.data(), qx.data(), qu.data(), qs.data(), c.data(), cd.data())) break;
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
      for (int k = 0; k <= N; ++k)
        for (int i = 0; i < NX; ++i) {
          if (!var(k, i)) continue;
          const int j = k * NX + i;
          const double sl = slx(X.data(), k, i), su = sux(X.data(), k, i), d = dX[j];
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
      if constexpr (NR > 0)
        for (int k = 0; k < N; ++k)
          for (int i = 0; i < NR; ++i) {
            const int j = k * NR + i;
            const double sl = sls(S.data(), k, i), su = sus(S.data(), k, i), d = dS[j];
            dzls[j] = hls[i] ? mu / sl - zls[j] - zls[j] / sl * d : 0.0;
            dzus[j] = hus[i] ? mu / su - zus[j] + zus[j] / su * d : 0.0;
            if (hls[i] && dzls[j] < 0) alpha_z = std::min(alpha_z, -tau * zls[j] / dzls[j]);
            if (hus[i] && dzus[j] < 0) alpha_z = std::min(alpha_z, -tau * zus[j] / dzus[j]);
          }
      const double alpha_max = alpha_primal(dX.data(), dU.data(), tau, dS.data());
      // ---- filter line search (W&B Alg. A) ----
      const double phi0 = barrier(f, X.data(), U.data(), mu, S.data()), th0 = theta_of(c, cd);
      double dphi = 0.0;
      for (int k = 0; k <= N; ++k)
        for (int i = 0; i < NX; ++i)
          if (var(k, i)) dphi += qx[k * NX + i] * dX[k * NX + i];
      for (int i = 0; i < N * NU; ++i) dphi += qu[i] * dU[i];
      for (int i = 0; i < N * NR; ++i) dphi += qs[i] * dS[i];
      double alpha = alpha_max;
      bool accepted = false, armijo = false, resto = false;
      lam_step = lamn;
      lamd_step = lamdn;
      for (int ls = 0; ls < 60 && !accepted; ++ls) {
        for (size_t i = 0; i < X.size(); ++i) Xt[i] = X[i] + alpha * dX[i];
        for (size_t i = 0; i < U.size(); ++i) Ut[i] = U[i] + alpha * dU[i];
        for (size_t i = 0; i < S.size(); ++i) St[i] = S[i] + alpha * dS[i];
        const double ft = eval_fc(Xt.data(), Ut.data(), ct.data(), St.data(), cdt.data());
        const double pht = barrier(ft, Xt.data(), Ut.data(), mu, St.data()), tht = theta_of(ct, cdt);
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
          for (size_t i = 0; i < cd.size(); ++i) cdsoc[i] = alpha * cd[i] + cdt[i];
          double th_old = tht;
          dXs = dX; dUs = dU; lams = lamn; dSs = dS; lamds = lamdn;
          for (int q = 0; q < max_soc; ++q) {
            if (!kkt_solve(Hz.data(), HN.data(), dgx.data(), dgu.data(), dgs.data(), qx.data(), qu.data(), qs.data(), csoc.data(),
                           cdsoc.data()))
              break;
            const double a_s = alpha_primal(dX.data(), dU.data(), tau, dS.data());
            for (size_t i = 0; i < X.size(); ++i) Xs[i] = X[i] + a_s * dX[i];
            for (size_t i = 0; i < U.size(); ++i) Us[i] = U[i] + a_s * dU[i];
            for (size_t i = 0; i < S.size(); ++i) Ss[i] = S[i] + a_s * dS[i];
            const double fs = eval_fc(Xs.data(), Us.data(), cs.data(), Ss.data(), cds.data());
            const double phs = barrier(fs, Xs.data(), Us.data(), mu, Ss.data()), ths = theta_of(cs, cds);
            bool sw2 = false;
            if (acceptable(ths, phs, &sw2)) {
              ok = true; sw = sw2;
              Xt = Xs; Ut = Us; St = Ss;
              lam_step = lamn;
              lamd_step = lamdn;
              break;
            }
            if (!(ths <= kappa_soc * th_old)) break;
            th_old = ths;
            for (size_t i = 0; i < c.size(); ++i) csoc[i] = a_s * csoc[i] + cs[i];
            for (size_t i = 0; i < cd.size(); ++i) cdsoc[i] = a_s * cdsoc[i] + cds[i];
          }
          dX = dXs; dU = dUs; lamn = lams; dS = dSs; lamdn = lamds;
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
        std::fill(lamd.begin(), lamd.end(), 0.0);
        double zmax = 0.0;
        for (double z : zlx) zmax = std::max(zmax, z);
        for (double z : zux) zmax = std::max(zmax, z);
        for (double z : zlu) zmax = std::max(zmax, z);
        for (double z : zuu) zmax = std::max(zmax, z);
        for (double z : zls) zmax = std::max(zmax, z);
        for (double z : zus) zmax = std::max(zmax, z);
        if (zmax > 1e3) reset_bound_multipliers();    // bound_mult_reset_threshold
      } else {
        if (!armijo) {                                // augment the filter (W&B eq. 22)
          filt.emplace_back((1 - gamma_theta) * th0, phi0 - gamma_phi * th0);
          if ((int)filt.size() > max_filter) filt.erase(filt.begin());
        }
        X = Xt; U = Ut; S = St;
        for (size_t i = 0; i < lam.size(); ++i) lam[i] += alpha * (lam_step[i] - lam[i]);
        for (size_t i = 0; i < lamd.size(); ++i) lamd[i] += alpha * (lamd_step[i] - lamd[i]);
        for (size_t j = 0; j < zlx.size(); ++j) { zlx[j] += alpha_z * dzlx[j]; zux[j] += alpha_z * dzux[j]; }
        for (int j = 0; j < N * NU; ++j) { zlu[j] += alpha_z * dzlu[j]; zuu[j] += alpha_z * dzuu[j]; }
        for (int j = 0; j < N * NR; ++j) { zls[j] += alpha_z * dzls[j]; zus[j] += alpha_z * dzus[j]; }
      }
      // W&B eq. 16: keep z within [mu / (kappa s), kappa mu / s]
      for (int k = 0; k <= N; ++k)
        for (int i = 0; i < NX; ++i) {
          if (!var(k, i)) continue;
          const int j = k * NX + i;
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
      if constexpr (NR > 0)
        for (int k = 0; k < N; ++k)
          for (int i = 0; i < NR; ++i) {
            const int j = k * NR + i;
            const double sl = sls(S.data(), k, i), su = sus(S.data(), k, i);
            zls[j] = hls[i] ? std::min(std::max(zls[j], mu / (kappa_sigma * sl)), kappa_sigma * mu / sl) : 0.0;
            zus[j] = hus[i] ? std::min(std::max(zus[j], mu / (kappa_sigma * su)), kappa_sigma * mu / su) : 0.0;
          }
      ++iters;
    }
    f = eval_all(lam.data(), lamd.data());
    if (f_opt) *f_opt = f;
    *status_o = status;
    *iters_o = iters;
    if (kkt_o) *kkt_o = errors(0.0);
  }
};

}  // namespace hilo_cpu
