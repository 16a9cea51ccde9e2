// CPU baseline: model right-hand sides and Runge-Kutta tableaux shared by the legs (nmpc_cpu.cpp, mhe_cpu.cpp, kf_cpu.cpp).
// TEST INFRASTRUCTURE / BASELINE ONLY - see nmpc_cpu.cpp.
#pragma once
#include <cmath>

namespace hilo_cpu {

using std::cos;
using std::exp;
using std::sin;

// ---- models (continuous right-hand sides), templated on the scalar type (double, forward-mode types) ---------------------------
// `ecoli_D1210_conti('simple')` with the closed-form rate laws (hilo_mpc/library/models.py:163-198, :143-148);
// states X, S, P, I; inputs DS, DI; parameters Sf, If, ISF, IRF
struct Chemostat4 {
  static constexpr int NX = 4, NU = 2, NP = 4;
  template <class T> static void ode(const T* x, const T* u, const double* p, T* dx) {
    const T& X = x[0]; const T& S = x[1]; const T& P = x[2]; const T& I = x[3];
    const T phi = 0.407 * S / (0.108 + S + S * S / 14814.0);
    const T mu = phi * (p[2] + 0.22 * p[3] / (0.22 + I));
    const T Rs = 2.0 * mu;
    const T Rfp = phi * (0.0005 + I) / (0.022 + I);
    const T D = u[0] + u[1];
    dx[0] = mu * X - D * X;
    dx[1] = -(Rs * X) - D * S + u[0] * p[0];
    dx[2] = Rfp * X - D * P;
    dx[3] = -(D * I) + u[1] * p[1];
  }
  static constexpr int NY = 2;      // measurements X and P (hilo_mpc/library/models.py:163-198: `y = [X, P]`)
  template <class T> static void meas(const T* x, const T*, const double*, T* y) { y[0] = x[0]; y[1] = x[2]; }
};

// Chemostat4 with the growth rate of the biomass balance given by a GP posterior mean over (S, I) - BASELINE configuration 4,
// `model.substitute_from(gp)` (hilo_mpc/modules/dynamic_model.py:3040-3125); squared-exponential kernel with one length scale per
// feature (gp/kernel.py:538-555, :696), posterior mean bias + sum_i alpha_i sf2 exp(-1/2 sum_d M_d (x_d - X_di)^2)
// (gp/inference.py:211-213), written out term by term like oracle/models.py::chemostat4_gp.  Rs and Rfp keep their closed forms.
struct GpSe2 {
  int n = 0;
  double sf2 = 1.0, bias = 0.0, M[2] = {1.0, 1.0};
  const double *X0 = nullptr, *X1 = nullptr, *alpha = nullptr;   // [n] each
};
inline GpSe2& gp_of_chemostat4() {   // one learned term per process (the baseline times one problem at a time)
  static GpSe2 g;
  return g;
}
struct Chemostat4Gp {
  static constexpr int NX = 4, NU = 2, NP = 4;
  template <class T> static void ode(const T* x, const T* u, const double* p, T* dx) {
    const T& X = x[0]; const T& S = x[1]; const T& P = x[2]; const T& I = x[3];
    const GpSe2& g = gp_of_chemostat4();
    T mu = 0.0 * S + g.bias;
    for (int i = 0; i < g.n; ++i) {
      const T d0 = S - g.X0[i], d1 = I - g.X1[i];
      mu = mu + (g.alpha[i] * g.sf2) * exp(-0.5 * (g.M[0] * (d0 * d0) + g.M[1] * (d1 * d1)));
    }
    const T phi = 0.407 * S / (0.108 + S + S * S / 14814.0);
    const T Rs = 2.0 * (phi * (p[2] + 0.22 * p[3] / (0.22 + I)));
    const T Rfp = phi * (0.0005 + I) / (0.022 + I);
    const T D = u[0] + u[1];
    dx[0] = mu * X - D * X;
    dx[1] = -(Rs * X) - D * S + u[0] * p[0];
    dx[2] = Rfp * X - D * P;
    dx[3] = -(D * I) + u[1] * p[1];
  }
};

// cart-pendulum of tests/test_NMPC.py:12-43: x, v, theta, omega; input F
struct Pendulum4 {
  static constexpr int NX = 4, NU = 1, NP = 0;
  template <class T> static void ode(const T* x, const T* u, const double*, T* dx) {
    const double M = 5.0, m = 1.0, l = 1.0, g = 9.81;
    const T s = sin(x[2]), c = cos(x[2]);
    const T dv = 1.0 / (M + m - m * c) * (m * g * s - m * l * s * x[3] * x[3] + u[0]);
    dx[0] = x[1];
    dx[1] = dv;
    dx[2] = x[3];
    dx[3] = 1.0 / l * (dv * c + g * s);
  }
};

// explicit Runge-Kutta tableaux of modeling.py:1239-1250 (order 1: Euler, 2: midpoint, 3: Kutta, 4: classic)
struct Tableau { int s; double A[4][4], b[4]; };
inline const Tableau& tableau(int order) {
  static const Tableau TAB[5] = {
      {},
      {1, {{0}}, {1.0}},
      {2, {{0}, {0.5}}, {0.0, 1.0}},
      {3, {{0}, {0.5}, {-1.0, 2.0}}, {1.0 / 6, 2.0 / 3, 1.0 / 6}},
      {4, {{0}, {0.5}, {0, 0.5}, {0, 0, 1.0}}, {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6}},
  };
  return TAB[order];
}

// x+ = Phi(x, u, p): n_sub steps of the tableau over dt (modeling.py:1213-1281)
template <class M, class T>
void erk_map(int order, int n_sub, double dt, const T* x0, const T* u, const double* p, T* out) {
  constexpr int NX = M::NX;
  T x[NX], k[4][NX], xi[NX];
  for (int i = 0; i < NX; ++i) x[i] = x0[i];
  const Tableau& t = tableau(order);
  const double h = dt / n_sub;
  for (int sub = 0; sub < n_sub; ++sub) {
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
  for (int i = 0; i < NX; ++i) out[i] = x[i];
}

}  // namespace hilo_cpu
