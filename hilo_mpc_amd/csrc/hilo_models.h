// Device model zoo + explicit Runge-Kutta shooting map.
//
// The reference holds a model as CasADi expressions (hilo_mpc/modules/dynamic_model/dynamic_model.py) and
// discretises it with `Model.discretize('erk'|'rk4', order)` -> RungeKutta._explicit
// (hilo_mpc/util/modeling.py:1213-1281, tableaux :1008-1085, order->tableau :1239-1250).  Here a model is a
// functor whose `ode`/`meas` are templated on the scalar type so that the same statement serves values
// (double), first-order sensitivities (Dual<N>) and second-order directional derivatives (Jet2).
//
// ids match HILO_MODEL_* in include/hilo_hip.h.
#pragma once
#include "hilo_ad.h"

namespace hilo {

enum ModelId : int {
  MODEL_LTI = 0,          // x+ = A x + B u, y = C x; A,B,C packed in p (row-major), dims fixed per instantiation
  MODEL_TOY1D = 1,        // reference tests/test_KFs.py:548-556
  MODEL_BIOREACTOR3 = 2,  // reference tests/test_KFs.py:691-712
  MODEL_CHEMOSTAT4 = 3,   // hilo_mpc/library/models.py:163-198 closed with the rate laws of :143-148
  MODEL_PENDULUM4 = 4,    // reference tests/test_NMPC.py:12-43
  MODEL_ROBOT6 = 5,
  MODEL_CSTR3 = 6,
  MODEL_LINEAR2 = 7,      // reference tests/test_KFs.py:247-255
};

// ---- tests/test_KFs.py:247-255: dx1 = -k1 x1 + u, dx2 = k1 x1 - k2 x2, y = x2 -------------------------
struct Linear2 {
  static constexpr int NX = 2, NU = 1, NP = 2, NY = 1;
  static constexpr bool DISCRETE = false;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P* p, double, T* dx) {
    dx[0] = -1.0 * (p[0] * x[0]) + u[0];
    dx[1] = p[0] * x[0] - p[1] * x[1];
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) { y[0] = x[1]; }
};

// ---- tests/test_KFs.py:548-556: x+ = x/2 + 25 dt x/(1+x^2), y = x^2/20 (discrete) ---------------------
struct Toy1D {
  static constexpr int NX = 1, NU = 0, NP = 0, NY = 1;
  static constexpr bool DISCRETE = true;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U*, const P*, double dt, T* xn) {
    xn[0] = x[0] / 2.0 + 25.0 * dt * x[0] / (1.0 + x[0] * x[0]);
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) { y[0] = x[0] * x[0] / 20.0; }
};

// ---- tests/test_KFs.py:691-712: bioreactor T, cB, cS; p = [alpha, T_amb, mu_0, mu_1, K, Y]; u = D ----
struct Bioreactor3 {
  static constexpr int NX = 3, NU = 1, NP = 6, NY = 2;
  static constexpr bool DISCRETE = false;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P* p, double, T* dx) {
    const T r = (p[2] + p[3] * x[0]) * x[2] * x[1] / (p[4] + x[2]);
    dx[0] = p[0] * (p[1] - x[0]);
    dx[1] = r - u[0] * x[1];
    dx[2] = -1.0 * (r / p[5]) - u[0] * x[2];
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) { y[0] = x[0]; y[1] = x[1]; }
};

// ---- CSTR-sized benchmark model: states X,S,P,I; inputs DS,DI; p = [Sf,If,ISF,IRF] ----
struct Chemostat4 {
  static constexpr int NX = 4, NU = 2, NP = 4, NY = 2;
  static constexpr bool DISCRETE = false;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P* p, double, T* dx) {
    const T X = x[0], S = x[1], Pr = x[2], I = x[3];
    const T phi = 0.407 * S / (0.108 + S + S * S / 14814.0);
    const T mu = phi * (p[2] + 0.22 * p[3] / (0.22 + I));
    const T Rs = 2.0 * mu;
    const T Rfp = phi * (0.0005 + I) / (0.022 + I);
    const T D = T(u[0]) + u[1];
    dx[0] = mu * X - D * X;
    dx[1] = -1.0 * (Rs * X) - D * S + u[0] * p[0];
    dx[2] = Rfp * X - D * Pr;
    dx[3] = -1.0 * (D * I) + u[1] * p[1];
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) { y[0] = x[0]; y[1] = x[2]; }
};

// ---- tests/test_NMPC.py:12-43: cart-pendulum x,v,theta,omega; input F; all states measured ---------------
struct Pendulum4 {
  static constexpr int NX = 4, NU = 1, NP = 0, NY = 4;
  static constexpr bool DISCRETE = false;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P*, double, T* dx) {
    const double M = 5.0, m = 1.0, l = 1.0, g = 9.81;
    const T s = sin(x[2]), c = cos(x[2]);
    const T dv = 1.0 / (M + m - m * c) * (m * g * s - m * l * s * x[3] * x[3] + u[0]);
    dx[0] = x[1];
    dx[1] = dv;
    dx[2] = x[3];
    dx[3] = 1.0 / l * (dv * c + g * s);
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) {
    y[0] = x[0]; y[1] = x[1]; y[2] = x[2]; y[3] = x[3];
  }
};

// ---- LTI with compile-time dims; p = [A (NX*NX) | B (NX*NU) | C (NY*NX)] row-major ------------------------
template <int NX_, int NU_, int NY_>
struct Lti {
  static constexpr int NX = NX_, NU = NU_, NP = NX_ * NX_ + NX_ * NU_ + NY_ * NX_, NY = NY_;
  static constexpr bool DISCRETE = true;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P* p, double, T* xn) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      T s = T(0.0);
#pragma unroll
      for (int j = 0; j < NX; ++j) s = s + p[i * NX + j] * x[j];
#pragma unroll
      for (int j = 0; j < NU; ++j) s = s + p[NX * NX + i * NU + j] * u[j];
      xn[i] = s;
    }
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P* p, double, T* y) {
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      T s = T(0.0);
#pragma unroll
      for (int j = 0; j < NX; ++j) s = s + p[NX * NX + NX * NU + i * NX + j] * x[j];
      y[i] = s;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Explicit Runge-Kutta map, modeling.py:1213-1281.  order 1..4 -> forward Euler / midpoint / Kutta / classic.
// Written as four guarded, fully unrolled stages so `order` can be a run-time (wave-uniform) value.
// ------------------------------------------------------------------------------------------------
// Tableau entries as selects on the (wave-uniform) order so that they stay in scalar registers.
template <int I, int J> HD double erk_a(int order) {
  if constexpr (I == 1 && J == 0) return order >= 2 ? 0.5 : 0.0;
  else if constexpr (I == 2 && J == 0) return order == 3 ? -1.0 : 0.0;
  else if constexpr (I == 2 && J == 1) return order == 3 ? 2.0 : (order == 4 ? 0.5 : 0.0);
  else if constexpr (I == 3 && J == 2) return order == 4 ? 1.0 : 0.0;
  else return 0.0;
}
template <int I> HD double erk_b(int order) {
  if constexpr (I == 0) return order == 1 ? 1.0 : (order == 2 ? 0.0 : 1.0 / 6);
  else if constexpr (I == 1) return order == 2 ? 1.0 : (order == 3 ? 2.0 / 3 : (order == 4 ? 1.0 / 3 : 0.0));
  else if constexpr (I == 2) return order == 3 ? 1.0 / 6 : (order == 4 ? 1.0 / 3 : 0.0);
  else return order == 4 ? 1.0 / 6 : 0.0;
}

template <class M, int I, class T, class U, class P>
HD void erk_stage(int order, const T* x, const U* u, const P* p, double h, T (*k)[M::NX]) {
  constexpr int NX = M::NX;
  if (I < order) {
    T xi[NX];
#pragma unroll
    for (int s = 0; s < NX; ++s) {
      T acc = x[s];
      if constexpr (I >= 1) acc = acc + (h * erk_a<I, 0>(order)) * k[0][s];
      if constexpr (I >= 2) acc = acc + (h * erk_a<I, 1>(order)) * k[1][s];
      if constexpr (I >= 3) acc = acc + (h * erk_a<I, 2>(order)) * k[2][s];
      xi[s] = acc;
    }
    M::ode(xi, u, p, h, k[I]);
  } else {
#pragma unroll
    for (int s = 0; s < NX; ++s) k[I][s] = T(0.0);
  }
}

template <class M, class T, class U, class P>
HD void erk_step(int order, const T* x, const U* u, const P* p, double h, T* xn) {
  constexpr int NX = M::NX;
  T k[4][NX];
  erk_stage<M, 0>(order, x, u, p, h, k);
  erk_stage<M, 1>(order, x, u, p, h, k);
  erk_stage<M, 2>(order, x, u, p, h, k);
  erk_stage<M, 3>(order, x, u, p, h, k);
#pragma unroll
  for (int s = 0; s < NX; ++s)
    xn[s] = x[s] + (h * erk_b<0>(order)) * k[0][s] + (h * erk_b<1>(order)) * k[1][s] +
            (h * erk_b<2>(order)) * k[2][s] + (h * erk_b<3>(order)) * k[3][s];
}

// one sampling interval of the shooting map: discrete models are evaluated directly (mpc.py:1381-1389,:1665),
// continuous ones through ERK of the requested order with `nsub` equal sub-steps
template <class M, class T, class U, class P>
HD void model_step(int order, int nsub, const T* x, const U* u, const P* p, double dt, T* xn) {
  if constexpr (M::DISCRETE) {
    M::ode(x, u, p, dt, xn);
  } else {
    constexpr int NX = M::NX;
    const double h = dt / nsub;
    T xc[NX];
#pragma unroll
    for (int s = 0; s < NX; ++s) xc[s] = x[s];
    for (int it = 0; it < nsub; ++it) {
      T xt[NX];
      erk_step<M>(order, xc, u, p, h, xt);
#pragma unroll
      for (int s = 0; s < NX; ++s) xc[s] = xt[s];
    }
#pragma unroll
    for (int s = 0; s < NX; ++s) xn[s] = xc[s];
  }
}

}  // namespace hilo
