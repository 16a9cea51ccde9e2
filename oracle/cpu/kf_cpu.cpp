// CPU baseline of the Kalman filter hot path (BASELINE configuration 3): extended and unscented Kalman filter steps of the
// chemostat model, C++17 + OpenMP over the instances of a batch.
//
// TEST INFRASTRUCTURE / BASELINE ONLY: loaded by bench.py's `cpu_baseline` leg and by tests/ (through oracle/cpu/__init__.py),
// never by the product package.  Validated against oracle/kf.py (which is pinned by the reference's known answers) in
// tests/test_cpu_baseline.py before it is timed.
//
// What it restates (hilo_mpc/modules/estimator/kf.py, reference v1.1.0), statement by statement like oracle/kf.py:
//   * predict  :71-133   x- = f(x, u, p),  P- = F P F^T + Q with F the Jacobian of the discretised model at the PRIOR state (:91)
//   * update   :135-186  P_xy = P H^T, P_yy = H P H^T + R, K = (P_yy^T \ P_xy^T)^T, x+ = x + K (y - h(x)), P+ = P - K P_yy K^T
//   * step     :258-265  update(predict(.))
//   * UKF      :486-604  lambda / gamma / weights (:493-500), sigma points x, x +- gamma S[:, k] with S = chol(P) the UPPER factor
//                        and its columns (:503, :522-527), sums accumulated in the reference's order (:542-548, :583-592), the
//                        update re-uses the propagated points (:580-592)
// The model is `Model.discretize('rk4')` of the chemostat (models_cpu.h); the Jacobian is first-order forward mode through the
// Runge-Kutta stages - what CasADi's graph of the discretised model provides.
#include <omp.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>

#include "models_cpu.h"

namespace {

using namespace hilo_cpu;

// ---- first-order forward mode over N directions -------------------------------------------------------------------------------
template <int N>
struct D1 {
  double v, g[N];
  D1() {}
  D1(double c) : v(c) {
    for (int i = 0; i < N; ++i) g[i] = 0.0;
  }
};
template <int N> D1<N> operator+(const D1<N>& a, const D1<N>& b) { D1<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.g[i] = a.g[i] + b.g[i]; return r; }
template <int N> D1<N> operator-(const D1<N>& a, const D1<N>& b) { D1<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.g[i] = a.g[i] - b.g[i]; return r; }
template <int N> D1<N> operator-(const D1<N>& a) { D1<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.g[i] = -a.g[i]; return r; }
template <int N> D1<N> operator*(const D1<N>& a, const D1<N>& b) { D1<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i]; return r; }
template <int N> D1<N> operator/(const D1<N>& a, const D1<N>& b) {
  D1<N> r;
  const double ib = 1.0 / b.v;
  r.v = a.v * ib;
  for (int i = 0; i < N; ++i) r.g[i] = (a.g[i] - r.v * b.g[i]) * ib;
  return r;
}
template <int N> D1<N> operator+(const D1<N>& a, double c) { D1<N> r = a; r.v += c; return r; }
template <int N> D1<N> operator+(double c, const D1<N>& a) { return a + c; }
template <int N> D1<N> operator-(const D1<N>& a, double c) { D1<N> r = a; r.v -= c; return r; }
template <int N> D1<N> operator-(double c, const D1<N>& a) { return -a + c; }
template <int N> D1<N> operator*(const D1<N>& a, double c) { D1<N> r; r.v = a.v * c; for (int i = 0; i < N; ++i) r.g[i] = a.g[i] * c; return r; }
template <int N> D1<N> operator*(double c, const D1<N>& a) { return a * c; }
template <int N> D1<N> operator/(const D1<N>& a, double c) { return a * (1.0 / c); }
template <int N> D1<N> operator/(double c, const D1<N>& a) { return D1<N>(c) / a; }

constexpr int NX = Chemostat4::NX, NU = Chemostat4::NU, NY = 2;
constexpr int YI[NY] = {0, 2};   // measurements X and P (hilo_mpc/library/models.py:163-198: `y = [X, P]`)

// K = (P_yy^T \ P_xy^T)^T, x+ = x + K (y - y_pred), P+ = P - K P_yy K^T   (kf.py:177-180 / :595-598)
void gain_update(double* x, double (*P)[NX], const double (*Pxy)[NY], const double (*Pyy)[NY], const double* y, const double* yp) {
  // 2 x 2 system P_yy^T K^T = P_xy^T by Gaussian elimination with partial pivoting (numpy.linalg.solve in the oracle)
  double K[NX][NY];
  for (int i = 0; i < NX; ++i) {
    double a00 = Pyy[0][0], a01 = Pyy[1][0], a10 = Pyy[0][1], a11 = Pyy[1][1];   // P_yy^T
    double b0 = Pxy[i][0], b1 = Pxy[i][1];
    if (std::fabs(a10) > std::fabs(a00)) { std::swap(a00, a10); std::swap(a01, a11); std::swap(b0, b1); }
    const double f = a10 / a00;
    a11 -= f * a01;
    b1 -= f * b0;
    K[i][1] = b1 / a11;
    K[i][0] = (b0 - a01 * K[i][1]) / a00;
  }
  for (int i = 0; i < NX; ++i)
    for (int m = 0; m < NY; ++m) x[i] += K[i][m] * (y[m] - yp[m]);
  double KP[NX][NY];
  for (int i = 0; i < NX; ++i)
    for (int m = 0; m < NY; ++m) KP[i][m] = K[i][0] * Pyy[0][m] + K[i][1] * Pyy[1][m];
  for (int i = 0; i < NX; ++i)
    for (int j = 0; j < NX; ++j) P[i][j] -= KP[i][0] * K[j][0] + KP[i][1] * K[j][1];
}

void ekf_step(int order, double dt, double* x, double (*P)[NX], const double* y, const double* u, const double* p, double q, double r) {
  using T = D1<NX>;
  T xs[NX], us[NU], xn[NX];
  for (int i = 0; i < NX; ++i) { xs[i] = T(x[i]); xs[i].g[i] = 1.0; }
  for (int i = 0; i < NU; ++i) us[i] = T(u[i]);
  erk_map<Chemostat4, T>(order, 1, dt, xs, us, p, xn);
  double FP[NX][NX], Pn[NX][NX];
  for (int i = 0; i < NX; ++i)
    for (int j = 0; j < NX; ++j) {
      double s = 0.0;
      for (int k = 0; k < NX; ++k) s += xn[i].g[k] * P[k][j];
      FP[i][j] = s;
    }
  for (int i = 0; i < NX; ++i)
    for (int j = 0; j < NX; ++j) {
      double s = 0.0;
      for (int k = 0; k < NX; ++k) s += FP[i][k] * xn[j].g[k];
      Pn[i][j] = s + (i == j ? q : 0.0);
    }
  for (int i = 0; i < NX; ++i) x[i] = xn[i].v;
  // update: H selects the measured states
  double Pxy[NX][NY], Pyy[NY][NY], yp[NY];
  for (int i = 0; i < NX; ++i)
    for (int m = 0; m < NY; ++m) Pxy[i][m] = Pn[i][YI[m]];
  for (int a = 0; a < NY; ++a)
    for (int m = 0; m < NY; ++m) Pyy[a][m] = Pn[YI[a]][YI[m]] + (a == m ? r : 0.0);
  for (int m = 0; m < NY; ++m) yp[m] = x[YI[m]];
  std::memcpy(P, Pn, sizeof(Pn));
  gain_update(x, P, Pxy, Pyy, y, yp);
}

void ukf_step(int order, double dt, double* x, double (*P)[NX], const double* y, const double* u, const double* p, double q, double r) {
  constexpr int NS = 2 * NX + 1;
  const double alpha = 1e-3, beta = 2.0, kappa = 0.0;                      // kf.py:486-492 defaults
  const double lam = alpha * alpha * (NX + kappa) - NX, gamma = std::sqrt(NX + lam);
  double W0[NS], W1[NS];
  W0[0] = lam / (NX + lam);
  W1[0] = lam / (NX + lam) + 1 - alpha * alpha + beta;
  for (int k = 1; k < NS; ++k) W0[k] = W1[k] = 1.0 / (2 * (NX + lam));
  // lower Cholesky factor L; the reference's upper factor is S = L^T: its column k is row k of L, S[i][k] = L[k][i]
  double L[NX][NX] = {};
  for (int j = 0; j < NX; ++j) {
    double d = P[j][j];
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    L[j][j] = std::sqrt(d);
    for (int i = j + 1; i < NX; ++i) {
      double s = P[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      L[i][j] = s / L[j][j];
    }
  }
  double X[NS][NX], Xp[NS][NX];
  for (int i = 0; i < NX; ++i) X[0][i] = x[i];
  for (int k = 0; k < NX; ++k)
    for (int i = 0; i < NX; ++i) {
      X[1 + k][i] = x[i] + gamma * L[k][i];
      X[1 + NX + k][i] = x[i] - gamma * L[k][i];
    }
  for (int k = 0; k < NS; ++k) erk_map<Chemostat4, double>(order, 1, dt, X[k], u, p, Xp[k]);
  double xp[NX] = {}, Pp[NX][NX];
  for (int k = 0; k < NS; ++k)
    for (int i = 0; i < NX; ++i) xp[i] += W0[k] * Xp[k][i];
  for (int i = 0; i < NX; ++i)
    for (int j = 0; j < NX; ++j) Pp[i][j] = (i == j ? q : 0.0);
  for (int k = 0; k < NS; ++k)
    for (int i = 0; i < NX; ++i)
      for (int j = 0; j < NX; ++j) Pp[i][j] += (W1[k] * (Xp[k][i] - xp[i])) * (Xp[k][j] - xp[j]);
  // update with the propagated points
  double yp[NY] = {}, Pxy[NX][NY] = {}, Pyy[NY][NY];
  for (int k = 0; k < NS; ++k)
    for (int m = 0; m < NY; ++m) yp[m] += W0[k] * Xp[k][YI[m]];
  for (int a = 0; a < NY; ++a)
    for (int m = 0; m < NY; ++m) Pyy[a][m] = (a == m ? r : 0.0);
  for (int k = 0; k < NS; ++k) {
    double dy[NY];
    for (int m = 0; m < NY; ++m) dy[m] = Xp[k][YI[m]] - yp[m];
    for (int i = 0; i < NX; ++i)
      for (int m = 0; m < NY; ++m) Pxy[i][m] += (W1[k] * (Xp[k][i] - xp[i])) * dy[m];
    for (int a = 0; a < NY; ++a)
      for (int m = 0; m < NY; ++m) Pyy[a][m] += (W1[k] * dy[a]) * dy[m];
  }
  for (int i = 0; i < NX; ++i) x[i] = xp[i];
  std::memcpy(P, Pp, sizeof(Pp));
  gain_update(x, P, Pxy, Pyy, y, yp);
}

}  // namespace

extern "C" {

// HOST pointers.  xP [batch][4][5] packed [x | P] (kf.py:129-133), updated in place over `steps` sampling instants;
// y [steps][batch][2]; u [batch][2], p [batch][4] held over the steps; Q = q I, R = r I.  n_threads <= 0: all cores.
int hilo_cpu_kf_steps(int ukf, int erk_order, double dt, int64_t batch, int steps, double* xP, const double* y, const double* u,
                      const double* p, double q, double r, int n_threads) {
  if (!xP || !y || !u || !p || erk_order < 1 || erk_order > 4 || steps < 1) return 1;
  if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (int64_t b = 0; b < batch; ++b) {
    double x[NX], P[NX][NX];
    double* t = xP + b * NX * (NX + 1);
    for (int i = 0; i < NX; ++i) {
      x[i] = t[i * (NX + 1)];
      for (int j = 0; j < NX; ++j) P[i][j] = t[i * (NX + 1) + 1 + j];
    }
    for (int s = 0; s < steps; ++s) {
      const double* ys = y + ((int64_t)s * batch + b) * NY;
      if (ukf) ukf_step(erk_order, dt, x, P, ys, u + b * NU, p + b * Chemostat4::NP, q, r);
      else ekf_step(erk_order, dt, x, P, ys, u + b * NU, p + b * Chemostat4::NP, q, r);
    }
    for (int i = 0; i < NX; ++i) {
      t[i * (NX + 1)] = x[i];
      for (int j = 0; j < NX; ++j) t[i * (NX + 1) + 1 + j] = P[i][j];
    }
  }
  return 0;
}

}  // extern "C"
