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
//   * the interior-point algorithm of oracle/nmpc.py::DenseIpm statement by statement (ipm_cpu.h; Waechter & Biegler 2006 with IPOPT's
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
#include "ipm_cpu.h"      // second-order forward mode H2<N>, StageIpm<Pol>
#include "models_cpu.h"

namespace {

using namespace hilo_cpu;   // models, tableaux (models_cpu.h), solver (ipm_cpu.h)

char g_err[512] = "";

struct Problem {
  int model_id, N, order, n_sub, nx, nu, np;
  double dt;
  IpmOptions opt;
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

// ---- the problem functions of the tracking NMPC for StageIpm (ipm_cpu.h): l = (z - zref)' Wz (z - zref), V = (x - xrefN)' WN (.),
// F = the scaled shooting map, x_0 pinned ---------------------------------------------------------------------------------------
template <class M>
struct TrackPolicy {
  static constexpr int NX = M::NX, NU = M::NU, NZ = NX + NU, NR = 0;
  bool free0[NX] = {};               // x_0 is the measured state
  using HD = H2<NZ>;
  const Problem& pb;
  const double* p = nullptr;
  explicit TrackPolicy(const Problem& pb_) : pb(pb_) {}

  double stage_fc(int, const double* x, const double* u, double* F) const {
    double f = 0.0, z[NZ];
    for (int i = 0; i < NX; ++i) z[i] = x[i] - pb.zref[i];
    for (int i = 0; i < NU; ++i) z[NX + i] = u[i] - pb.zref[NX + i];
    for (int i = 0; i < NZ; ++i)
      for (int j = 0; j < NZ; ++j) f += z[i] * pb.Wz[i * NZ + j] * z[j];
    phi_scaled<M, double>(pb, x, u, p, F);
    return f;
  }

  double stage_all(int, const double* x, const double* u, const double* lamk, double* gz, double* Hk, double* F, double* Ak,
                   double* Bk) const {
    double f = 0.0, z[NZ];
    HD xs[NX], us[NU], ph[NX];
    for (int i = 0; i < NX; ++i) z[i] = x[i] - pb.zref[i];
    for (int i = 0; i < NU; ++i) z[NX + i] = u[i] - pb.zref[NX + i];
    for (int i = 0; i < NZ; ++i) {
      double s = 0.0;
      for (int j = 0; j < NZ; ++j) { s += pb.Wz[i * NZ + j] * z[j]; f += z[i] * pb.Wz[i * NZ + j] * z[j]; }
      gz[i] = 2.0 * s;   // Wz symmetric (QuadraticCost builds it from diagonal / symmetric blocks)
    }
    for (int i = 0; i < NX; ++i) xs[i] = HD::seed(x[i], i);
    for (int i = 0; i < NU; ++i) us[i] = HD::seed(u[i], NX + i);
    phi_scaled<M, HD>(pb, xs, us, p, ph);
    for (int i = 0; i < NZ; ++i)
      for (int j = 0; j < NZ; ++j) Hk[i * NZ + j] = 2.0 * pb.Wz[i * NZ + j];
    for (int r = 0; r < NX; ++r) {
      F[r] = ph[r].v;
      for (int j = 0; j < NX; ++j) Ak[r * NX + j] = ph[r].g[j];
      for (int j = 0; j < NU; ++j) Bk[r * NU + j] = ph[r].g[NX + j];
      const double l = lamk[r];
      for (int i = 0; i < NZ; ++i)
        for (int j = 0; j < NZ; ++j) Hk[i * NZ + j] -= l * ph[r].hess(i, j);
    }
    return f;
  }

  double term_fc(const double* xN) const {
    double f = 0.0, d[NX];
    for (int i = 0; i < NX; ++i) d[i] = xN[i] - pb.xrefN[i];
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NX; ++j) f += d[i] * pb.WN[i * NX + j] * d[j];
    return f;
  }

  double term_all(const double* xN, double* gN, double* HN) const {
    double f = 0.0, d[NX];
    for (int i = 0; i < NX; ++i) d[i] = xN[i] - pb.xrefN[i];
    for (int i = 0; i < NX; ++i) {
      double s = 0.0;
      for (int j = 0; j < NX; ++j) { s += pb.WN[i * NX + j] * d[j]; f += d[i] * pb.WN[i * NX + j] * d[j]; HN[i * NX + j] = 2.0 * pb.WN[i * NX + j]; }
      gN[i] = 2.0 * s;
    }
    return f;
  }
};

// one instance in the layout of the reference: v = [x_0..x_N | u_0..u_{N-1}] scaled (mpc.py:1462-1485); v0: warm start (primal
// only, mpc.py:725-726) or NULL = the guesses of the descriptor
template <class M>
struct Solver {
  static constexpr int NX = M::NX, NU = M::NU;
  const Problem& pb;
  TrackPolicy<M> pol;
  StageIpm<TrackPolicy<M>> ipm;
  std::vector<double> X0, U0;
  explicit Solver(const Problem& pb_)
      : pb(pb_), pol(pb_), ipm(pol, pb_.opt, pb_.N, pb_.xlb.data(), pb_.xub.data(), pb_.ulb.data(), pb_.uub.data()),
        X0((pb_.N + 1) * NX), U0(pb_.N * NU) {}

  void solve(const double* par, const double* x0_orig, const double* v0, double* v_opt, double* f_opt, double* u0, int* status_o,
             int* iters_o, double* kkt_o) {
    const int N = pb.N;
    pol.p = par;
    for (int i = 0; i < NX; ++i) X0[i] = x0_orig[i] / pb.sx[i];
    for (int k = 1; k <= N; ++k)
      for (int i = 0; i < NX; ++i) X0[k * NX + i] = v0 ? v0[k * NX + i] : pb.xg[i];
    for (int k = 0; k < N; ++k)
      for (int i = 0; i < NU; ++i) U0[k * NU + i] = v0 ? v0[(N + 1) * NX + k * NU + i] : pb.ug[i];
    ipm.solve(X0.data(), U0.data(), f_opt, status_o, iters_o, kkt_o);
    if (v_opt) {
      std::memcpy(v_opt, ipm.X.data(), sizeof(double) * (N + 1) * NX);
      std::memcpy(v_opt + (N + 1) * NX, ipm.U.data(), sizeof(double) * N * NU);
    }
    if (u0)
      for (int i = 0; i < NU; ++i) u0[i] = ipm.U[i] * pb.su[i];
  }
};


template <class M>
void solve_batch(const Problem& pb, int64_t batch, const double* x0, const double* par, int64_t par_stride, const double* v0,
                 double* v_opt, double* f_opt, double* u0, int32_t* status, int32_t* iters, double* kkt, int n_threads) {
  const int nv = (pb.N + 1) * M::NX + pb.N * M::NU;
#pragma omp parallel num_threads(n_threads)
  {
    Solver<M> s(pb);
#pragma omp for schedule(dynamic, 4)
    for (int64_t b = 0; b < batch; ++b) {
      int st = 0, itc = 0;
      s.solve(par ? par + b * par_stride : nullptr, x0 + b * M::NX, v0 ? v0 + b * nv : nullptr, v_opt ? v_opt + b * nv : nullptr,
              f_opt ? f_opt + b : nullptr, u0 ? u0 + b * M::NU : nullptr, &st, &itc, kkt ? kkt + b : nullptr);
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
  p.opt.max_iter = d->max_iter > 0 ? d->max_iter : 3000;
  p.opt.acceptable_iter = d->acceptable_iter > 0 ? d->acceptable_iter : 15;
  p.dt = d->dt;
  p.opt.tol = d->tol > 0 ? d->tol : 1e-8;
  p.opt.acceptable_tol = d->acceptable_tol > 0 ? d->acceptable_tol : 1e-6;
  p.opt.mu_init = d->mu_init > 0 ? d->mu_init : 0.1;
  p.opt.relax = d->bound_relax_factor < 0 ? 1e-8 : d->bound_relax_factor;
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
