// CPU baseline of the path-following NMPC of configuration C5: the transcription of oracle/nmpc_gen.py::GenNmpcProblem for the
// robot with a path variable and a soft speed limit, on the stage-structured interior-point solver of ipm_cpu.h, C++17 + OpenMP
// over the instances of a batch.
//
// TEST INFRASTRUCTURE / BASELINE ONLY: loaded by bench.py's `cpu_baseline` leg and by tests/ (through oracle/cpu/__init__.py),
// never by the product package.
//
// What it restates (hilo_mpc/modules/controller/mpc.py, pre-discretised model + integration_method='discrete'):
//   * path following (:1025-1053, :1173-1204): the path variable theta is a model state with its own virtual input,
//     theta+ = theta + dt u_theta; theta_0 is a free, bounded variable (`optimize` pins the original states only, :785-789);
//     the path cost puts the expression of theta in the place of the reference, (s - r(theta))' W (s - r(theta))
//     (hilo_mpc/util/modeling.py:252-283), stage and terminal;
//   * the soft stage constraint (`GenericConstraint`, modeling.py:820-1005; mpc.py:1271-1283, :1700-1725): rows
//     c(x_k, u_k) - e <= ub, k = 0..N-1, with ONE slack e >= 0 shared by all stages (mpc.py:1529-1537) and the penalty e' W e once
//     per stage (:1708).
// The problem functions are those of C5 (tests/problems.py): robot6 (states px, vx, py, vy, psi, omega; inputs a, alpha), path
// references r(theta) = (sin theta, sin 2 theta) for (px, py), constraint vx^2 + vy^2 - e <= ub; weights, bounds and guesses come
// through the descriptor.  Form for the stage solver - the form the device engine uses: state (x, theta, e) with
// e+ = e, input (u, u_theta), theta_0 and e_0 variables, one inequality row per stage.  With the slack as a state the iteration
// path is not the dense oracle's (other multipliers, N + 1 bounds e_k >= 0 instead of one); the minimiser is - validated against
// oracle/nmpc_gen.py::GenIpm in tests/test_cpu_baseline.py before it is timed.
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

constexpr int MX = 6, MU = 2;                 // the robot
constexpr int IT = 6, IE = 7;                 // theta and e in the engine state

struct Robot6 {
  static constexpr int NX = MX, NU = MU, NP = 0;
  template <class T> static void ode(const T* x, const T* u, const double*, T* dx) {
    dx[0] = x[1];
    dx[1] = u[0] * cos(x[4]);
    dx[2] = x[3];
    dx[3] = u[0] * sin(x[4]);
    dx[4] = x[5];
    dx[5] = u[1];
  }
};

struct PfProblem {
  int N, order, n_sub;
  double dt;
  IpmOptions opt;
  double Wz[(MX + MU) * (MX + MU)], zref[MX + MU], WN[MX * MX], xrefN[MX];
  double wps[2], wpt[2];                      // path weights on (px, py), stage and terminal
  double We, con_ub;
  double xlb[MX + 2], xub[MX + 2], ulb[MU + 1], uub[MU + 1], xg[MX + 2], ug[MU + 1];
};

struct PfPolicy {
  static constexpr int NX = MX + 2, NU = MU + 1, NZ = NX + NU, NR = 1, NM = MX + MU;
  bool free0[NX];
  const PfProblem& pb;
  explicit PfPolicy(const PfProblem& pb_) : pb(pb_) {
    std::fill(free0, free0 + NX, false);
    free0[IT] = free0[IE] = true;
  }
  // engine index of entry i of the model's z = (x, u)
  static int zi(int i) { return i < MX ? i : NX + (i - MX); }

  template <class T>
  static T path_cost(const double* w, const T& px, const T& py, const T& th) {
    const T d0 = px - sin(th), d1 = py - sin(2.0 * th);
    return w[0] * (d0 * d0) + w[1] * (d1 * d1);
  }

  double stage_fc(int, const double* x, const double* u, double* F) const {
    double z[NM], f = 0.0;
    for (int i = 0; i < MX; ++i) z[i] = x[i] - pb.zref[i];
    for (int i = 0; i < MU; ++i) z[MX + i] = u[i] - pb.zref[MX + i];
    for (int i = 0; i < NM; ++i)
      for (int j = 0; j < NM; ++j) f += z[i] * pb.Wz[i * NM + j] * z[j];
    f += path_cost<double>(pb.wps, x[0], x[2], x[IT]) + pb.We * x[IE] * x[IE];
    erk_map<Robot6, double>(pb.order, pb.n_sub, pb.dt, x, u, nullptr, F);
    F[IT] = x[IT] + pb.dt * u[MU];
    F[IE] = x[IE];
    return f;
  }

  double stage_all(int, const double* x, const double* u, const double* lamk, double* gz, double* Hk, double* F, double* Ak,
                   double* Bk) const {
    std::fill(Hk, Hk + NZ * NZ, 0.0);
    std::fill(gz, gz + NZ, 0.0);
    std::fill(Ak, Ak + NX * NX, 0.0);
    std::fill(Bk, Bk + NX * NU, 0.0);
    double z[NM], f = 0.0;
    for (int i = 0; i < MX; ++i) z[i] = x[i] - pb.zref[i];
    for (int i = 0; i < MU; ++i) z[MX + i] = u[i] - pb.zref[MX + i];
    for (int i = 0; i < NM; ++i) {
      double s = 0.0;
      for (int j = 0; j < NM; ++j) {
        s += pb.Wz[i * NM + j] * z[j];
        Hk[zi(i) * NZ + zi(j)] += 2.0 * pb.Wz[i * NM + j];
      }
      f += z[i] * s;
      gz[zi(i)] += 2.0 * s;         // Wz symmetric
    }
    {                               // path term in (px, py, theta)
      using H3 = H2<3>;
      const H3 c = path_cost<H3>(pb.wps, H3::seed(x[0], 0), H3::seed(x[2], 1), H3::seed(x[IT], 2));
      const int id[3] = {0, 2, IT};
      f += c.v;
      for (int i = 0; i < 3; ++i) {
        gz[id[i]] += c.g[i];
        for (int j = 0; j < 3; ++j) Hk[id[i] * NZ + id[j]] += c.hess(i, j);
      }
    }
    f += pb.We * x[IE] * x[IE];
    gz[IE] += 2.0 * pb.We * x[IE];
    Hk[IE * NZ + IE] += 2.0 * pb.We;
    {                               // the robot's shooting map in second-order forward mode over its own (x, u)
      using HD = H2<NM>;
      HD xs[MX], us[MU], ph[MX];
      for (int i = 0; i < MX; ++i) xs[i] = HD::seed(x[i], i);
      for (int i = 0; i < MU; ++i) us[i] = HD::seed(u[i], MX + i);
      erk_map<Robot6, HD>(pb.order, pb.n_sub, pb.dt, xs, us, nullptr, ph);
      for (int r = 0; r < MX; ++r) {
        F[r] = ph[r].v;
        for (int j = 0; j < MX; ++j) Ak[r * NX + j] = ph[r].g[j];
        for (int j = 0; j < MU; ++j) Bk[r * NU + j] = ph[r].g[MX + j];
        const double l = lamk[r];
        for (int i = 0; i < NM; ++i)
          for (int j = 0; j < NM; ++j) Hk[zi(i) * NZ + zi(j)] -= l * ph[r].hess(i, j);
      }
    }
    F[IT] = x[IT] + pb.dt * u[MU];
    Ak[IT * NX + IT] = 1.0;
    Bk[IT * NU + MU] = pb.dt;
    F[IE] = x[IE];
    Ak[IE * NX + IE] = 1.0;
    return f;
  }

  double term_fc(const double* xN) const {
    double f = 0.0, d[MX];
    for (int i = 0; i < MX; ++i) d[i] = xN[i] - pb.xrefN[i];
    for (int i = 0; i < MX; ++i)
      for (int j = 0; j < MX; ++j) f += d[i] * pb.WN[i * MX + j] * d[j];
    return f + path_cost<double>(pb.wpt, xN[0], xN[2], xN[IT]);
  }

  double term_all(const double* xN, double* gN, double* HN) const {
    std::fill(gN, gN + NX, 0.0);
    std::fill(HN, HN + NX * NX, 0.0);
    double f = 0.0, d[MX];
    for (int i = 0; i < MX; ++i) d[i] = xN[i] - pb.xrefN[i];
    for (int i = 0; i < MX; ++i) {
      double s = 0.0;
      for (int j = 0; j < MX; ++j) { s += pb.WN[i * MX + j] * d[j]; HN[i * NX + j] += 2.0 * pb.WN[i * MX + j]; }
      f += d[i] * s;
      gN[i] += 2.0 * s;
    }
    using H3 = H2<3>;
    const H3 c = path_cost<H3>(pb.wpt, H3::seed(xN[0], 0), H3::seed(xN[2], 1), H3::seed(xN[IT], 2));
    const int id[3] = {0, 2, IT};
    for (int i = 0; i < 3; ++i) {
      gN[id[i]] += c.g[i];
      for (int j = 0; j < 3; ++j) HN[id[i] * NX + id[j]] += c.hess(i, j);
    }
    return f + c.v;
  }

  // the row vx^2 + vy^2 - e
  void rows_fc(int, const double* x, const double*, double* d) const { d[0] = x[1] * x[1] + x[3] * x[3] - x[IE]; }
  void rows_all(int, const double* x, const double*, const double* lamd, double* d, double* Jd, double* Hk) const {
    d[0] = x[1] * x[1] + x[3] * x[3] - x[IE];
    std::fill(Jd, Jd + NZ, 0.0);
    Jd[1] = 2.0 * x[1];
    Jd[3] = 2.0 * x[3];
    Jd[IE] = -1.0;
    Hk[1 * NZ + 1] += 2.0 * lamd[0];
    Hk[3 * NZ + 3] += 2.0 * lamd[0];
  }
};

}  // namespace

// the problem data of C5 that are not fixed by its functions; NULL = zero weight / no bound / zero guess
struct hilo_cpu_pf_desc {
  int32_t N, erk_order, n_sub, max_iter, acceptable_iter;
  double dt, tol, acceptable_tol, mu_init, bound_relax_factor;     // 0 / < 0 -> IPOPT defaults, see hilo_nmpc_desc
  const double* Wz;       // [8][8] over (x, u)
  const double* zref;     // [8]
  const double* WN;       // [6][6]
  const double* xrefN;    // [6]
  const double *x_lb, *x_ub, *u_lb, *u_ub, *x_guess, *u_guess;   // [6] / [2]
  double w_path_stage[2], w_path_term[2];
  double theta_lb, theta_ub, theta_guess, u_pf_lb, u_pf_ub;
  double con_ub, con_weight, max_violation;
};

struct hilo_cpu_pf { PfProblem pb; };

extern "C" {

const char* hilo_cpu_pf_last_error(void) { return g_err; }

int hilo_cpu_pf_create(const hilo_cpu_pf_desc* d, hilo_cpu_pf** out) {
  if (!d || !out) return fail("NULL argument");
  if (d->N < 1 || d->dt <= 0) return fail("bad horizon / dt");
  hilo_cpu_pf* h = new hilo_cpu_pf();
  PfProblem& p = h->pb;
  p.N = d->N;
  p.order = d->erk_order >= 1 ? d->erk_order : 4;
  p.n_sub = d->n_sub >= 1 ? d->n_sub : 1;
  if (p.order > 4) { delete h; return fail("explicit Runge-Kutta order 1..4"); }
  p.dt = d->dt;
  p.opt.max_iter = d->max_iter > 0 ? d->max_iter : 3000;
  p.opt.acceptable_iter = d->acceptable_iter > 0 ? d->acceptable_iter : 15;
  p.opt.tol = d->tol > 0 ? d->tol : 1e-8;
  p.opt.acceptable_tol = d->acceptable_tol > 0 ? d->acceptable_tol : 1e-6;
  p.opt.mu_init = d->mu_init > 0 ? d->mu_init : 0.1;
  p.opt.relax = d->bound_relax_factor < 0 ? 1e-8 : d->bound_relax_factor;
  auto cp = [](double* v, const double* s, int n, double dflt) { for (int i = 0; i < n; ++i) v[i] = s ? s[i] : dflt; };
  constexpr int NM = MX + MU;
  cp(p.Wz, d->Wz, NM * NM, 0.0); cp(p.zref, d->zref, NM, 0.0); cp(p.WN, d->WN, MX * MX, 0.0); cp(p.xrefN, d->xrefN, MX, 0.0);
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

void hilo_cpu_pf_destroy(hilo_cpu_pf* h) { delete h; }

// HOST pointers.  x0 [batch][6]; w0 [batch][(N + 1) 8 + N 3] warm start in the solver's own layout [X | U] or NULL (guesses);
// outputs: w_opt (same layout), v_opt [batch][(N + 1) 7 + N 3 + 1] in the reference's layout [x_0..x_N with theta | u with u_theta
// | e] (mpc.py:1462-1537), f_opt, first input [batch][2], status (optimizer.py:1085-1104), iteration count, scaled KKT error.
int hilo_cpu_pf_solve(hilo_cpu_pf* h, int64_t batch, const double* x0, const double* w0, double* w_opt, double* v_opt, double* f_opt,
                      double* first_u, int32_t* status, int32_t* iters, double* kkt, int n_threads) {
  if (!h || !x0 || !status || !iters) return fail("NULL argument");
  if (n_threads <= 0) n_threads = omp_get_max_threads();
  const PfProblem& pb = h->pb;
  constexpr int NX = PfPolicy::NX, NU = PfPolicy::NU;
  const int N = pb.N, nw = (N + 1) * NX + N * NU, nv = (N + 1) * (MX + 1) + N * NU + 1;
  const double dlb = -INF, dub = pb.con_ub;
#pragma omp parallel num_threads(n_threads)
  {
    PfPolicy pol(pb);
    StageIpm<PfPolicy> ipm(pol, pb.opt, N, pb.xlb, pb.xub, pb.ulb, pb.uub, &dlb, &dub);
    std::vector<double> X0((N + 1) * NX), U0(N * NU);
#pragma omp for schedule(dynamic, 2)
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
      if (v_opt) {
        double* v = v_opt + b * nv;
        for (int k = 0; k <= N; ++k)
          for (int i = 0; i <= MX; ++i) v[k * (MX + 1) + i] = ipm.X[k * NX + i];
        std::memcpy(v + (N + 1) * (MX + 1), ipm.U.data(), sizeof(double) * N * NU);
        v[nv - 1] = ipm.X[IE];
      }
      if (first_u)
        for (int i = 0; i < MU; ++i) first_u[b * MU + i] = ipm.U[i];
    }
  }
  return 0;
}

// x+ = Phi(x, u) of the robot (the plant of the closed loop)
int hilo_cpu_pf_plant_step(hilo_cpu_pf* h, int64_t batch, const double* x, const double* u, double* xn, int n_threads) {
  if (!h || !x || !u || !xn) return fail("NULL argument");
  if (n_threads <= 0) n_threads = omp_get_max_threads();
  const PfProblem& pb = h->pb;
#pragma omp parallel for num_threads(n_threads)
  for (int64_t b = 0; b < batch; ++b) erk_map<Robot6, double>(pb.order, pb.n_sub, pb.dt, x + b * MX, u + b * MU, nullptr, xn + b * MX);
  return 0;
}

}  // extern "C"
