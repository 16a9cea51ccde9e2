// CPU baseline of BASELINE configuration 5 AS IT IS WRITTEN (tests/problems.py::C5D): path-following NMPC on the robot's DAE - the
// squared speed as algebraic state, 0 = z - (vx^2 + vy^2) - with the soft limit on the algebraic state, under the reference's
// default transcription: direct collocation (Radau points of degree 3 by default) with the continuous objective.  C++17 + OpenMP
// over the instances of a batch, on the stage-structured interior-point solver of ipm_cpu.h.
//
// TEST INFRASTRUCTURE / BASELINE ONLY: loaded by bench.py's `cpu_baseline` leg and by tests/ (through oracle/cpu/__init__.py),
// never by the product package.
//
// What it restates (hilo_mpc/modules/controller/mpc.py with integration_method='collocation'; the dense statement of the same
// NLP is oracle/nmpc_coll_gen.py, against which tests/test_cpu_baseline.py validates this leg before it is timed):
//   * the path variable theta is a state of the model with theta' = u_theta (:1173-1204): it has collocation states like the others;
//   * per interval the collocation equations dt f(x_{k,i}, u_k) - sum_j C[j,i] x_{k,j} = 0 and the algebraic equations at the
//     collocation points (hilo_mpc/util/modeling.py:1128-1211), continuity x_{k+1} = sum_j D_j x_{k,j};
//   * the stage constraint at EVERY collocation point and at the node (:1338-1356, :1700-1725): rows z - e <= ub with ONE slack e
//     shared by all stages (:1529-1537), penalty e' W e once per interval (:1708);
//   * the continuous objective: dt sum_i B_i l(x_{k,i}, u_k) (modeling.py:1195), l = the quadratic input term + the path term.
// Form for the stage solver - the form the device engine uses: the collocation states of an interval are
// eliminated by solving its square collocation system (Newton on the Runge-Kutta form), the algebraic state through its
// equation; the interval's map and its rows are differentiated exactly by two further Newton sweeps in second-order forward mode
// (the k-th sweep fixes the k-th order).  State (x, theta, e) with e+ = e, input (u, u_theta), D + 1 inequality rows per stage.
#include <omp.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ipm_cpu.h"
#include "models_cpu.h"

namespace {

using namespace hilo_cpu;

char g_err[512] = "";
int fail(const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return -1;
}

constexpr int MX = 6, MU = 2, MXA = MX + 1, MUA = MU + 1, NM = MXA + MUA;   // the robot + path variable; NM: derivative directions
constexpr int IT = 6, IE = 7;                                                 // theta and e in the engine state
constexpr int DMAX = 4;

// robot + theta' = u_theta
template <class T>
void ode_aug(const T* x, const T* u, T* dx) {
  dx[0] = x[1];
  dx[1] = u[0] * cos(x[4]);
  dx[2] = x[3];
  dx[3] = u[0] * sin(x[4]);
  dx[4] = x[5];
  dx[5] = u[1];
  dx[6] = u[2];
}
// the algebraic state through its equation 0 = z - (vx^2 + vy^2)
template <class T> T alg_z(const T* x) { return x[1] * x[1] + x[3] * x[3]; }

struct PdProblem {
  int N, D;
  double dt;
  IpmOptions opt;
  double A[DMAX * DMAX], Dc[DMAX + 1], Bq[DMAX + 1];      // Runge-Kutta matrix of the collocation method, continuity / quadrature weights
  double Wu[MU], uref[MU];                                 // quadratic input term (diagonal)
  double wps[2], wpt[2];
  double We, con_ub;
  double xlb[MXA + 1], xub[MXA + 1], ulb[MUA], uub[MUA], xg[MXA + 1], ug[MUA];
};

// dense LU with partial pivoting, n <= 28
struct Lu {
  int n, piv[28];
  double a[28 * 28];
  bool factor() {
    for (int k = 0; k < n; ++k) {
      int p = k;
      for (int i = k + 1; i < n; ++i)
        if (std::fabs(a[i * n + k]) > std::fabs(a[p * n + k])) p = i;
      piv[k] = p;
      if (p != k)
        for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[p * n + j]);
      if (a[k * n + k] == 0.0) return false;
      const double ip = 1.0 / a[k * n + k];
      for (int i = k + 1; i < n; ++i) {
        const double f = a[i * n + k] * ip;
        a[i * n + k] = f;
        for (int j = k + 1; j < n; ++j) a[i * n + j] -= f * a[k * n + j];
      }
    }
    return true;
  }
  void solve(double* b) const {
    for (int k = 0; k < n; ++k) std::swap(b[k], b[piv[k]]);
    for (int i = 1; i < n; ++i) {
      double s = b[i];
      for (int j = 0; j < i; ++j) s -= a[i * n + j] * b[j];
      b[i] = s;
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = b[i];
      for (int j = i + 1; j < n; ++j) s -= a[i * n + j] * b[j];
      b[i] = s / a[i * n + i];
    }
  }
};

// collocation states of one interval: values X [D][MXA] by Newton on  X_i = x + dt sum_j A_ij f(X_j, u);  leaves the factors
bool coll_values(const PdProblem& pb, const double* x, const double* u, double* X, Lu& lu) {
  const int D = pb.D, n = D * MXA;
  for (int i = 0; i < D; ++i)
    for (int m = 0; m < MXA; ++m) X[i * MXA + m] = x[m];
  lu.n = n;
  for (int it = 0; it < 20; ++it) {
    using H1 = H2<MXA>;     // (only values and gradients are read)
    double F[DMAX * MXA], J[DMAX][MXA * MXA];
    for (int j = 0; j < D; ++j) {
      H1 xs[MXA], us[MUA], f[MXA];
      for (int a = 0; a < MXA; ++a) xs[a] = H1::seed(X[j * MXA + a], a);
      for (int a = 0; a < MUA; ++a) us[a] = H1(u[a]);
      ode_aug(xs, us, f);
      for (int m = 0; m < MXA; ++m) {
        F[j * MXA + m] = f[m].v;
        for (int a = 0; a < MXA; ++a) J[j][m * MXA + a] = f[m].g[a];
      }
    }
    double R[DMAX * MXA], scale = 1.0;
    for (int i = 0; i < D; ++i)
      for (int m = 0; m < MXA; ++m) {
        double r = X[i * MXA + m] - x[m];
        for (int j = 0; j < D; ++j) r -= pb.dt * pb.A[i * D + j] * F[j * MXA + m];
        R[i * MXA + m] = -r;
        scale = std::max(scale, std::fabs(X[i * MXA + m]));
        for (int j = 0; j < D; ++j)
          for (int a = 0; a < MXA; ++a)
            lu.a[(i * MXA + m) * n + j * MXA + a] = ((i == j && m == a) ? 1.0 : 0.0) - pb.dt * pb.A[i * D + j] * J[j][m * MXA + a];
      }
    if (!lu.factor()) return false;
    lu.solve(R);
    double dmax = 0.0;
    for (int q = 0; q < n; ++q) { X[q] += R[q]; dmax = std::max(dmax, std::fabs(R[q])); }
    if (!(dmax > 1e-13 * scale)) return true;      // (the factors belong to a point within round-off of the solution)
  }
  return true;
}

// the interval in second-order forward mode over (x [MXA], u [MUA]): collocation states XJ, two sweeps with the value factors
using HD = H2<NM>;
void coll_taylor(const PdProblem& pb, const double* x, const double* u, const double* X, const Lu& lu, HD* XJ, HD* xs, HD* us) {
  const int D = pb.D, n = D * MXA;
  for (int a = 0; a < MXA; ++a) xs[a] = HD::seed(x[a], a);
  for (int a = 0; a < MUA; ++a) us[a] = HD::seed(u[a], MXA + a);
  for (int q = 0; q < n; ++q) XJ[q] = HD(X[q]);
  std::vector<double> rhs(n);
  for (int rep = 0; rep < 2; ++rep) {
    HD F[DMAX * MXA], R[DMAX * MXA];
    for (int j = 0; j < D; ++j) ode_aug(XJ + j * MXA, us, F + j * MXA);
    for (int i = 0; i < D; ++i)
      for (int m = 0; m < MXA; ++m) {
        HD r = XJ[i * MXA + m] - xs[m];
        for (int j = 0; j < D; ++j) r = r - (pb.dt * pb.A[i * D + j]) * F[j * MXA + m];
        R[i * MXA + m] = r;
      }
    for (int c = 0; c < NM; ++c) {                 // first-order coefficients
      for (int q = 0; q < n; ++q) rhs[q] = -R[q].g[c];
      lu.solve(rhs.data());
      for (int q = 0; q < n; ++q) XJ[q].g[c] += rhs[q];
    }
    for (int c = 0; c < HD::NH; ++c) {             // second-order coefficients
      for (int q = 0; q < n; ++q) rhs[q] = -R[q].h[c];
      lu.solve(rhs.data());
      for (int q = 0; q < n; ++q) XJ[q].h[c] += rhs[q];
    }
  }
}

template <int DD>
struct PdPolicy {
  static constexpr int NX = MXA + 1, NU = MUA, NZ = NX + NU, NR = DD + 1;   // rows: the node, the collocation points 1..D
  bool free0[NX];
  const PdProblem& pb;
  explicit PdPolicy(const PdProblem& pb_) : pb(pb_) {
    std::fill(free0, free0 + NX, false);
    free0[IT] = free0[IE] = true;
  }
  // engine index of derivative direction c (x | theta | u | u_theta)
  static int zi(int c) { return c < MXA ? c : NX + (c - MXA); }

  template <class T>
  T lagrange(const T* xs, const T* us) const {     // quadratic input term + path term at a point
    T l = T(0.0);
    for (int a = 0; a < MU; ++a) l = l + pb.Wu[a] * ((us[a] - pb.uref[a]) * (us[a] - pb.uref[a]));
    const T d0 = xs[0] - sin(xs[IT]), d1 = xs[2] - sin(2.0 * xs[IT]);
    return l + pb.wps[0] * (d0 * d0) + pb.wps[1] * (d1 * d1);
  }

  double stage_fc(int, const double* x, const double* u, double* F) const {
    double X[DMAX * MXA];
    Lu lu;
    coll_values(pb, x, u, X, lu);
    double f = 0.0;
    for (int i = 0; i < pb.D; ++i) f += pb.dt * pb.Bq[i + 1] * lagrange<double>(X + i * MXA, u);
    for (int m = 0; m < MXA; ++m) {
      double s = pb.Dc[0] * x[m];
      for (int i = 0; i < pb.D; ++i) s += pb.Dc[i + 1] * X[i * MXA + m];
      F[m] = s;
    }
    F[IE] = x[IE];
    return f + pb.We * x[IE] * x[IE];
  }

  double stage_all(int, const double* x, const double* u, const double* lamk, double* gz, double* Hk, double* F, double* Ak,
                   double* Bk) const {
    std::fill(Hk, Hk + NZ * NZ, 0.0);
    std::fill(gz, gz + NZ, 0.0);
    std::fill(Ak, Ak + NX * NX, 0.0);
    std::fill(Bk, Bk + NX * NU, 0.0);
    double X[DMAX * MXA];
    Lu lu;
    coll_values(pb, x, u, X, lu);
    HD XJ[DMAX * MXA], xs[MXA], us[MUA];
    coll_taylor(pb, x, u, X, lu, XJ, xs, us);
    HD cost(0.0);
    for (int i = 0; i < pb.D; ++i) cost = cost + (pb.dt * pb.Bq[i + 1]) * lagrange<HD>(XJ + i * MXA, us);
    for (int c = 0; c < NM; ++c) {
      gz[zi(c)] += cost.g[c];
      for (int e = 0; e < NM; ++e) Hk[zi(c) * NZ + zi(e)] += cost.hess(c, e);
    }
    for (int m = 0; m < MXA; ++m) {
      HD s = pb.Dc[0] * xs[m];
      for (int i = 0; i < pb.D; ++i) s = s + pb.Dc[i + 1] * XJ[i * MXA + m];
      F[m] = s.v;
      for (int c = 0; c < MXA; ++c) Ak[m * NX + c] = s.g[c];
      for (int c = 0; c < MUA; ++c) Bk[m * NU + c] = s.g[MXA + c];
      const double l = lamk[m];
      for (int c = 0; c < NM; ++c)
        for (int e = 0; e < NM; ++e) Hk[zi(c) * NZ + zi(e)] -= l * s.hess(c, e);
    }
    F[IE] = x[IE];
    Ak[IE * NX + IE] = 1.0;
    gz[IE] += 2.0 * pb.We * x[IE];
    Hk[IE * NZ + IE] += 2.0 * pb.We;
    return cost.v + pb.We * x[IE] * x[IE];
  }

  double term_fc(const double* xN) const {
    const double d0 = xN[0] - std::sin(xN[IT]), d1 = xN[2] - std::sin(2.0 * xN[IT]);
    return pb.wpt[0] * d0 * d0 + pb.wpt[1] * d1 * d1;
  }
  double term_all(const double* xN, double* gN, double* HN) const {
    std::fill(gN, gN + NX, 0.0);
    std::fill(HN, HN + NX * NX, 0.0);
    using H3 = H2<3>;
    const H3 px = H3::seed(xN[0], 0), py = H3::seed(xN[2], 1), th = H3::seed(xN[IT], 2);
    const H3 d0 = px - sin(th), d1 = py - sin(2.0 * th);
    const H3 c = pb.wpt[0] * (d0 * d0) + pb.wpt[1] * (d1 * d1);
    const int id[3] = {0, 2, IT};
    for (int i = 0; i < 3; ++i) {
      gN[id[i]] += c.g[i];
      for (int j = 0; j < 3; ++j) HN[id[i] * NX + id[j]] += c.hess(i, j);
    }
    return c.v;
  }

  // rows: [0] the node z(x_k) - e, [i] the collocation point i: z(x_{k,i}) - e
  void rows_fc(int, const double* x, const double* u, double* d) const {
    double X[DMAX * MXA];
    Lu lu;
    coll_values(pb, x, u, X, lu);
    d[0] = alg_z<double>(x) - x[IE];
    for (int i = 0; i < DD; ++i) d[1 + i] = alg_z<double>(X + i * MXA) - x[IE];
  }
  void rows_all(int, const double* x, const double* u, const double* lamd, double* d, double* Jd, double* Hk) const {
    double X[DMAX * MXA];
    Lu lu;
    coll_values(pb, x, u, X, lu);
    HD XJ[DMAX * MXA], xs[MXA], us[MUA];
    coll_taylor(pb, x, u, X, lu, XJ, xs, us);
    std::fill(Jd, Jd + NR * NZ, 0.0);
    for (int r = 0; r < NR; ++r) {
      const HD z = r == 0 ? alg_z<HD>(xs) : alg_z<HD>(XJ + (r - 1) * MXA);
      d[r] = z.v - x[IE];
      for (int c = 0; c < NM; ++c) {
        Jd[r * NZ + zi(c)] = z.g[c];
        for (int e = 0; e < NM; ++e) Hk[zi(c) * NZ + zi(e)] += lamd[r] * z.hess(c, e);
      }
      Jd[r * NZ + IE] = -1.0;
    }
  }
};

}  // namespace

// problem data that are not fixed by the problem functions; NULL = zero weight / no bound / zero guess
struct hilo_cpu_pfdae_desc {
  int32_t N, degree, max_iter, acceptable_iter;
  double dt, tol, acceptable_tol, mu_init, bound_relax_factor;
  const double *coll_A, *coll_D, *coll_B;          // [d][d], [d + 1], [d + 1] (hilo_mpc_amd/nmpc.py::_collocation_basis restates them too)
  const double *Wu, *uref;                         // [2] diagonal input weights / references
  const double *x_lb, *x_ub, *u_lb, *u_ub, *x_guess, *u_guess;   // [6] / [2]
  double w_path_stage[2], w_path_term[2];
  double theta_lb, theta_ub, theta_guess, u_pf_lb, u_pf_ub;
  double con_ub, con_weight, max_violation;
};

struct hilo_cpu_pfdae { PdProblem pb; };

struct Robot6P {
  static constexpr int NX = MX, NU = MU, NP = 0;
  static void ode(const double* x, const double* u, const double*, double* dx) {
    dx[0] = x[1]; dx[1] = u[0] * std::cos(x[4]); dx[2] = x[3]; dx[3] = u[0] * std::sin(x[4]); dx[4] = x[5]; dx[5] = u[1];
  }
};

template <int DD>
static int pfdae_solve_t(hilo_cpu_pfdae* h, int64_t batch, const double* x0, const double* w0, double* w_opt, double* vx, double* f_opt,
                         double* first_u, int32_t* status, int32_t* iters, double* kkt, int n_threads) {
  const PdProblem& pb = h->pb;
  using Pol = PdPolicy<DD>;
  constexpr int NX = Pol::NX, NU = Pol::NU, NR = Pol::NR;
  const int N = pb.N, nw = (N + 1) * NX + N * NU, nv = (N + 1) * MXA + N * NU + 1;
  double dlb[NR], dub[NR];
  for (int r = 0; r < NR; ++r) { dlb[r] = -INF; dub[r] = pb.con_ub; }
#pragma omp parallel num_threads(n_threads)
  {
    Pol pol(pb);
    StageIpm<Pol> ipm(pol, pb.opt, N, pb.xlb, pb.xub, pb.ulb, pb.uub, dlb, dub);
    std::vector<double> X0((N + 1) * NX), U0(N * NU);
#pragma omp for schedule(dynamic, 1)
    for (int64_t b = 0; b < batch; ++b) {
      const double* w = w0 ? w0 + b * nw : nullptr;
      for (int k = 0; k <= N; ++k)
        for (int i = 0; i < NX; ++i) X0[k * NX + i] = w ? w[k * NX + i] : pb.xg[i];
      for (int i = 0; i < MX; ++i) X0[i] = x0[b * MX + i];
      for (int k = 0; k < N; ++k)
        for (int i = 0; i < NU; ++i) U0[k * NU + i] = w ? w[(N + 1) * NX + k * NU + i] : pb.ug[i];
      int st = 0, itc = 0;
      ipm.solve(X0.data(), U0.data(), f_opt ? f_opt + b : nullptr, &st, &itc, kkt ? kkt + b : nullptr);
      status[b] = st;
      iters[b] = itc;
      if (w_opt) {
        std::memcpy(w_opt + b * nw, ipm.X.data(), sizeof(double) * (N + 1) * NX);
        std::memcpy(w_opt + b * nw + (N + 1) * NX, ipm.U.data(), sizeof(double) * N * NU);
      }
      if (vx) {
        double* v = vx + b * nv;
        for (int k = 0; k <= N; ++k)
          for (int i = 0; i < MXA; ++i) v[k * MXA + i] = ipm.X[k * NX + i];
        std::memcpy(v + (N + 1) * MXA, ipm.U.data(), sizeof(double) * N * NU);
        v[nv - 1] = ipm.X[IE];
      }
      if (first_u)
        for (int i = 0; i < MU; ++i) first_u[b * MU + i] = ipm.U[i];
    }
  }
  return 0;
}

extern "C" {

const char* hilo_cpu_pfdae_last_error(void) { return g_err; }

int hilo_cpu_pfdae_create(const hilo_cpu_pfdae_desc* d, hilo_cpu_pfdae** out) {
  if (!d || !out) return fail("NULL argument");
  if (d->N < 1 || d->dt <= 0) return fail("bad horizon / dt");
  if (d->degree < 1 || d->degree > DMAX || !d->coll_A || !d->coll_D || !d->coll_B) return fail("collocation degree 1..4 with its basis");
  hilo_cpu_pfdae* h = new hilo_cpu_pfdae();
  PdProblem& p = h->pb;
  p.N = d->N;
  p.D = d->degree;
  p.dt = d->dt;
  p.opt.max_iter = d->max_iter > 0 ? d->max_iter : 3000;
  p.opt.acceptable_iter = d->acceptable_iter > 0 ? d->acceptable_iter : 15;
  p.opt.tol = d->tol > 0 ? d->tol : 1e-8;
  p.opt.acceptable_tol = d->acceptable_tol > 0 ? d->acceptable_tol : 1e-6;
  p.opt.mu_init = d->mu_init > 0 ? d->mu_init : 0.1;
  p.opt.relax = d->bound_relax_factor < 0 ? 1e-8 : d->bound_relax_factor;
  for (int i = 0; i < p.D * p.D; ++i) p.A[i] = d->coll_A[i];
  for (int i = 0; i <= p.D; ++i) { p.Dc[i] = d->coll_D[i]; p.Bq[i] = d->coll_B[i]; }
  auto cp = [](double* v, const double* s, int n, double dflt) { for (int i = 0; i < n; ++i) v[i] = s ? s[i] : dflt; };
  cp(p.Wu, d->Wu, MU, 0.0); cp(p.uref, d->uref, MU, 0.0);
  cp(p.xlb, d->x_lb, MX, -INF); cp(p.xub, d->x_ub, MX, INF); cp(p.ulb, d->u_lb, MU, -INF); cp(p.uub, d->u_ub, MU, INF);
  cp(p.xg, d->x_guess, MX, 0.0); cp(p.ug, d->u_guess, MU, 0.0);
  p.xlb[IT] = d->theta_lb; p.xub[IT] = d->theta_ub; p.xg[IT] = d->theta_guess;
  p.xlb[IE] = 0.0; p.xub[IE] = d->max_violation; p.xg[IE] = 0.0;
  p.ulb[MU] = d->u_pf_lb; p.uub[MU] = d->u_pf_ub; p.ug[MU] = d->u_pf_lb + 1e-4;      // mpc.py:1194-1195
  for (int i = 0; i < 2; ++i) { p.wps[i] = d->w_path_stage[i]; p.wpt[i] = d->w_path_term[i]; }
  p.We = d->con_weight;
  p.con_ub = d->con_ub;
  *out = h;
  return 0;
}

void hilo_cpu_pfdae_destroy(hilo_cpu_pfdae* h) { delete h; }

// HOST pointers.  x0 [batch][6]; w0 [batch][(N + 1) 8 + N 3] warm start in the solver's own layout [X | U] or NULL (guesses);
// outputs: w_opt (same layout), f_opt, first input [batch][2], status, iteration count, scaled KKT error; vx [batch][(N + 1) 7 + N 3
// + 1] = the [x with theta | u with u_theta | e] part of the reference's decision vector (the collocation blocks are not rebuilt)
int hilo_cpu_pfdae_solve(hilo_cpu_pfdae* h, int64_t batch, const double* x0, const double* w0, double* w_opt, double* vx, double* f_opt,
                         double* first_u, int32_t* status, int32_t* iters, double* kkt, int n_threads) {
  if (!h || !x0 || !status || !iters) return fail("NULL argument");
  if (n_threads <= 0) n_threads = omp_get_max_threads();
  switch (h->pb.D) {
    case 1: return pfdae_solve_t<1>(h, batch, x0, w0, w_opt, vx, f_opt, first_u, status, iters, kkt, n_threads);
    case 2: return pfdae_solve_t<2>(h, batch, x0, w0, w_opt, vx, f_opt, first_u, status, iters, kkt, n_threads);
    case 3: return pfdae_solve_t<3>(h, batch, x0, w0, w_opt, vx, f_opt, first_u, status, iters, kkt, n_threads);
    default: return pfdae_solve_t<4>(h, batch, x0, w0, w_opt, vx, f_opt, first_u, status, iters, kkt, n_threads);
  }
}

// the plant of the closed loop: the robot over one sampling interval with the classic Runge-Kutta step (what the product's
// `plant_step` does for a continuous model)
int hilo_cpu_pfdae_plant_step(hilo_cpu_pfdae* h, int64_t batch, const double* x, const double* u, double* xn, int n_threads) {
  if (!h || !x || !u || !xn) return fail("NULL argument");
  if (n_threads <= 0) n_threads = omp_get_max_threads();
  const PdProblem& pb = h->pb;
#pragma omp parallel for num_threads(n_threads)
  for (int64_t b = 0; b < batch; ++b) erk_map<Robot6P, double>(4, 1, pb.dt, x + b * MX, u + b * MU, nullptr, xn + b * MX);
  return 0;
}

}  // extern "C"
