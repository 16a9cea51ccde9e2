// CPU baseline of the moving-horizon estimator: the transcription of oracle/mhe.py::MheProblem / MheIpm with the stage-structured
// interior-point solver of ipm_cpu.h, C++17 + OpenMP over the instances of a batch, behind a C ABI that takes the product's own
// problem description (`hilo_mhe_desc`, include/hilo_hip.h).
//
// TEST INFRASTRUCTURE / BASELINE ONLY: loaded by bench.py's `cpu_baseline` leg and by tests/ (through oracle/cpu/__init__.py),
// never by the product package.
//
// What it restates - `MovingHorizonEstimator.setup` for a pre-discretised model with integration_method='discrete' and state
// noise, parameters pinned (hilo_mpc/modules/estimator/mhe.py:596-790):
//   v = [p | x_0..x_N | w_0..w_{N-1}] in scaled variables                     (mhe.py:614-655)
//   rows x_{k+1} - (Phi_s(x_k, u_meas_k, p) + w_k) = 0                         (mhe.py:733-740)
//   J = arrival(x_0) at k = 0, stage(w_k, x_k, y_k) for k >= 1                 (mhe.py:742-748: no stage cost at k = 0)
//   arrival = (x_0 sx - x_arr)' Wx (.), stage = (h(x_k sx) - y_k)' Wy (.) + (w_k sw)' Ww (w_k sw)   (util/modeling.py:665-777)
//   u_meas enters the scaled model un-divided: the model sees u_meas * su      (mhe.py:352 vs :242)
// As a stage problem for StageIpm: state x_k, "input" w_k (B_k = I), x_0 a variable.  Derivatives: second-order forward mode in
// the nx states of the interval (the noise enters linearly).
// Scope: chemostat4 (configuration C3); anything else in the descriptor is refused.  Validated against oracle/mhe.py in
// tests/test_cpu_baseline.py before it is timed.
#include <omp.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/hilo_hip.h"
#include "ipm_cpu.h"
#include "models_cpu.h"

namespace {

using namespace hilo_cpu;

char g_err[512] = "";
int fail(const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return -1;
}

struct MheProblem {
  int N, order, n_sub;
  double dt;
  IpmOptions opt;
  std::vector<double> Wx, Wy, Ww, xlb, xub, wlb, wub, sx, sw, su, xg, wg;
};

template <class M>
struct MhePolicy {
  static constexpr int NX = M::NX, NU = M::NX, NZ = 2 * M::NX, NY = M::NY, NUM = M::NU, NR = 0;
  bool free0[NX];                    // x_0 is a variable
  using HD = H2<NX>;
  const MheProblem& pb;
  const double *p = nullptr, *xa = nullptr, *um = nullptr, *ym = nullptr;    // per instance: [np], [nx], [N][nu], [N][ny]
  explicit MhePolicy(const MheProblem& pb_) : pb(pb_) { std::fill(free0, free0 + NX, true); }

  // Phi_s and the measurement function at the scaled state xs (T = double or HD)
  template <class T>
  void model(int k, const T* xs, T* ph, T* y) const {
    T x[NX], u[NUM];
    for (int i = 0; i < NX; ++i) x[i] = xs[i] * pb.sx[i];
    for (int i = 0; i < NUM; ++i) u[i] = T(um[k * NUM + i] * pb.su[i]);
    erk_map<M, T>(pb.order, pb.n_sub, pb.dt, x, u, p, ph);
    for (int i = 0; i < NX; ++i) ph[i] = ph[i] * (1.0 / pb.sx[i]);
    if (k >= 1) M::meas(x, u, p, y);
  }

  double stage_fc(int k, const double* x, const double* w, double* F) const {
    double y[NY], f = 0.0;
    model<double>(k, x, F, y);
    for (int i = 0; i < NX; ++i) F[i] += w[i];
    if (k == 0) {
      double d[NX];
      for (int i = 0; i < NX; ++i) d[i] = x[i] * pb.sx[i] - xa[i];
      for (int i = 0; i < NX; ++i)
        for (int j = 0; j < NX; ++j) f += d[i] * pb.Wx[i * NX + j] * d[j];
      return f;
    }
    double r[NY], ws[NX];
    for (int a = 0; a < NY; ++a) r[a] = y[a] - ym[k * NY + a];
    for (int a = 0; a < NY; ++a)
      for (int b = 0; b < NY; ++b) f += r[a] * pb.Wy[a * NY + b] * r[b];
    for (int i = 0; i < NX; ++i) ws[i] = w[i] * pb.sw[i];
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NX; ++j) f += ws[i] * pb.Ww[i * NX + j] * ws[j];
    return f;
  }

  double stage_all(int k, const double* x, const double* w, const double* lamk, double* gz, double* Hk, double* F, double* Ak,
                   double* Bk) const {
    HD xs[NX], ph[NX], y[NY];
    for (int i = 0; i < NX; ++i) xs[i] = HD::seed(x[i], i);
    model<HD>(k, xs, ph, y);
    std::fill(Hk, Hk + NZ * NZ, 0.0);
    std::fill(gz, gz + NZ, 0.0);
    for (int r = 0; r < NX; ++r) {
      F[r] = ph[r].v + w[r];
      for (int j = 0; j < NX; ++j) { Ak[r * NX + j] = ph[r].g[j]; Bk[r * NX + j] = r == j ? 1.0 : 0.0; }
      for (int i = 0; i < NX; ++i)
        for (int j = 0; j < NX; ++j) Hk[i * NZ + j] -= lamk[r] * ph[r].hess(i, j);
    }
    double f = 0.0;
    if (k == 0) {
      double d[NX];
      for (int i = 0; i < NX; ++i) d[i] = x[i] * pb.sx[i] - xa[i];
      for (int i = 0; i < NX; ++i) {
        double s = 0.0;
        for (int j = 0; j < NX; ++j) {
          s += pb.Wx[i * NX + j] * d[j];       // Wx symmetric (the weights are built from diagonals / symmetric blocks)
          Hk[i * NZ + j] += 2.0 * pb.Wx[i * NX + j] * pb.sx[i] * pb.sx[j];
        }
        f += d[i] * s;
        gz[i] = 2.0 * s * pb.sx[i];
      }
      return f;
    }
    double r[NY], rW[NY], ws[NX];
    for (int a = 0; a < NY; ++a) r[a] = y[a].v - ym[k * NY + a];
    for (int a = 0; a < NY; ++a) {
      double s = 0.0;
      for (int b = 0; b < NY; ++b) s += pb.Wy[a * NY + b] * r[b];
      rW[a] = s;
      f += r[a] * s;
    }
    for (int a = 0; a < NY; ++a) {
      for (int i = 0; i < NX; ++i) {
        gz[i] += 2.0 * rW[a] * y[a].g[i];
        double t = 0.0;                       // (Wy hx)_a over column i
        for (int b = 0; b < NY; ++b) t += pb.Wy[a * NY + b] * y[b].g[i];
        for (int j = 0; j < NX; ++j) Hk[j * NZ + i] += 2.0 * y[a].g[j] * t;
      }
      for (int i = 0; i < NX; ++i)
        for (int j = 0; j < NX; ++j) Hk[i * NZ + j] += 2.0 * rW[a] * y[a].hess(i, j);
    }
    for (int i = 0; i < NX; ++i) ws[i] = w[i] * pb.sw[i];
    for (int i = 0; i < NX; ++i) {
      double s = 0.0;
      for (int j = 0; j < NX; ++j) {
        s += pb.Ww[i * NX + j] * ws[j];
        Hk[(NX + i) * NZ + NX + j] += 2.0 * pb.Ww[i * NX + j] * pb.sw[i] * pb.sw[j];
      }
      f += ws[i] * s;
      gz[NX + i] = 2.0 * s * pb.sw[i];
    }
    return f;
  }

  double term_fc(const double*) const { return 0.0; }
  double term_all(const double*, double* gN, double* HN) const {
    std::fill(gN, gN + NX, 0.0);
    std::fill(HN, HN + NX * NX, 0.0);
    return 0.0;
  }
};

template <class M>
void estimate_batch(const MheProblem& pb, int64_t batch, const double* xa, const double* par, int64_t par_stride, const double* um,
                    const double* ym, const double* v0, double* v_opt, double* f_opt, double* x_opt, int32_t* status, int32_t* iters,
                    double* kkt, int n_threads) {
  constexpr int NX = M::NX, NP = M::NP;
  const int N = pb.N, nv = NP + (2 * N + 1) * NX;
#pragma omp parallel num_threads(n_threads)
  {
    MhePolicy<M> pol(pb);
    StageIpm<MhePolicy<M>> ipm(pol, pb.opt, N, pb.xlb.data(), pb.xub.data(), pb.wlb.data(), pb.wub.data());
    std::vector<double> X0((N + 1) * NX), U0(N * NX);
#pragma omp for schedule(dynamic, 4)
    for (int64_t b = 0; b < batch; ++b) {
      pol.p = par + b * par_stride;
      pol.xa = xa + b * NX;
      pol.um = um + b * N * M::NU;
      pol.ym = ym + b * N * M::NY;
      const double* w0 = v0 ? v0 + b * nv + NP : nullptr;
      for (int k = 0; k <= N; ++k)
        for (int i = 0; i < NX; ++i) X0[k * NX + i] = w0 ? w0[k * NX + i] : pb.xg[i];
      for (int k = 0; k < N; ++k)
        for (int i = 0; i < NX; ++i) U0[k * NX + i] = w0 ? w0[(N + 1) * NX + k * NX + i] : pb.wg[i];
      int st = 0, itc = 0;
      ipm.solve(X0.data(), U0.data(), f_opt ? f_opt + b : nullptr, &st, &itc, kkt ? kkt + b : nullptr);
      status[b] = st;
      iters[b] = itc;
      if (v_opt) {
        double* v = v_opt + b * nv;
        std::memcpy(v, pol.p, sizeof(double) * NP);
        std::memcpy(v + NP, ipm.X.data(), sizeof(double) * (N + 1) * NX);
        std::memcpy(v + NP + (N + 1) * NX, ipm.U.data(), sizeof(double) * N * NX);
      }
      if (x_opt)
        for (int i = 0; i < NX; ++i) x_opt[b * NX + i] = ipm.X[N * NX + i] * pb.sx[i];    // mhe.py:381-384
    }
  }
}

}  // namespace

struct hilo_cpu_mhe { MheProblem pb; };

extern "C" {

const char* hilo_cpu_mhe_last_error(void) { return g_err; }

// Same descriptor as hilo_mhe_create (include/hilo_hip.h); the subset this baseline covers is checked here.
int hilo_cpu_mhe_create(const hilo_mhe_desc* d, hilo_cpu_mhe** out) {
  if (!d || !out) return fail("NULL argument");
  if (d->model_id != HILO_MODEL_CHEMOSTAT4) return fail("the CPU baseline of the estimator holds the model chemostat4 only");
  if (d->N < 1 || d->dt <= 0) return fail("bad horizon / dt");
  if (d->estimate_parameters || d->user_source || d->collocation_degree)
    return fail("the CPU baseline covers the state-noise estimator with pinned parameters and a discretised model only");
  constexpr int nx = Chemostat4::NX, nu = Chemostat4::NU, ny = Chemostat4::NY;
  hilo_cpu_mhe* h = new hilo_cpu_mhe();
  MheProblem& p = h->pb;
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
  auto cp = [](std::vector<double>& v, const double* s, int n, double dflt) { v.resize(n); for (int i = 0; i < n; ++i) v[i] = s ? s[i] : dflt; };
  cp(p.Wx, d->Wx, nx * nx, 0.0); cp(p.Wy, d->Wy, ny * ny, 0.0); cp(p.Ww, d->Ww, nx * nx, 0.0);
  cp(p.sx, d->x_scaling, nx, 1.0); cp(p.sw, d->w_scaling, nx, 1.0); cp(p.su, d->u_scaling, nu, 1.0);
  cp(p.xlb, d->x_lb, nx, -INF); cp(p.xub, d->x_ub, nx, INF); cp(p.wlb, d->w_lb, nx, -INF); cp(p.wub, d->w_ub, nx, INF);
  cp(p.xg, d->x_guess, nx, 0.0); cp(p.wg, d->w_guess, nx, 0.0);
  for (int i = 0; i < nx; ++i) {     // bounds and guesses of the scaled variables (mhe.py:640-655)
    p.xlb[i] /= p.sx[i]; p.xub[i] /= p.sx[i]; p.xg[i] /= p.sx[i];
    p.wlb[i] /= p.sw[i]; p.wub[i] /= p.sw[i]; p.wg[i] /= p.sw[i];
  }
  *out = h;
  return 0;
}

void hilo_cpu_mhe_destroy(hilo_cpu_mhe* h) { delete h; }

// HOST pointers throughout, shapes like hilo_mhe_estimate: x_arrival [batch][nx] original units; p [batch][p_stride] pinned model
// parameters; u_meas [batch][N][nu]; y_meas [batch][N][ny]; v0 [batch][n_v] scaled warm start (primal) or NULL = the guesses of the
// descriptor; outputs v_opt [batch][n_v] = [p | x | w] scaled, f_opt, x_opt [batch][nx] = x_N un-scaled, status
// (optimizer.py:1085-1104), iteration count, scaled KKT error.  n_threads <= 0: all cores.
int hilo_cpu_mhe_estimate(hilo_cpu_mhe* h, int64_t batch, const double* x_arrival, const double* p, int64_t p_stride,
                          const double* u_meas, const double* y_meas, const double* v0, double* v_opt, double* f_opt, double* x_opt,
                          int32_t* status, int32_t* iters, double* kkt, int n_threads) {
  if (!h || !x_arrival || !p || !u_meas || !y_meas || !status || !iters) return fail("NULL argument");
  if (n_threads <= 0) n_threads = omp_get_max_threads();
  estimate_batch<Chemostat4>(h->pb, batch, x_arrival, p, p_stride, u_meas, y_meas, v0, v_opt, f_opt, x_opt, status, iters, kkt,
                             n_threads);
  return 0;
}

}  // extern "C"
