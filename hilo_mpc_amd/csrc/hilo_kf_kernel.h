// Batched Kalman / extended / unscented Kalman filter kernels (one filter instance per lane): device code, shared by the
// library (csrc/hilo_kf.hip: zoo models) and the run-time compiled filters of models written as expressions (csrc/hilo_jit.hip).
//
// Reference semantics: hilo_mpc/modules/estimator/kf.py
//   predict  :71-133   (UKF :505-554)      update :135-186 (UKF :556-604)      step = update(predict) :258-265
// HBM-bound: per step an instance moves its packed [x|P] tile in and out plus y, [u;p]
// (bytes_kf = 8*(2 nx (nx+1) + 2 ny + nu + np)).  Tiles travel HBM<->LDS with unit-stride lanes and are then
// picked up row-per-lane (hilo_common.h), all arithmetic is register-resident fp64 with compile-time shapes.
#pragma once
#include "hilo_common.h"
#include "hilo_models.h"

namespace hilo {

constexpr int KF_TPB = 64;   // one wave per workgroup

struct KfParams {
  int kind, continuous, erk_order, n_sub;
  double dt, gamma, wm0, wc0, wi;  // UKF: W_m[0], W_c[0], W[1:] (kf.py:493-500)
  // hilo_kf_steps_split: the parameters in their own array (rows of np doubles, stride pp_stride or 0 = shared) - `up` then holds
  // the inputs alone (rows of nu doubles); nullptr: `up` holds the packed rows [u; p]
  const double* pp = nullptr;
  long long pp_stride = 0;
};

template <int N> struct MaxOne { static constexpr int v = N > 0 ? N : 1; };

// [u; p] of one instance into registers: packed rows, or inputs and parameters from their own arrays (KfParams::pp)
template <class M>
__device__ __forceinline__ void kf_load_up(const KfParams& kp, const double* __restrict__ up, int64_t inst, int64_t up_stride,
                                           double* __restrict__ upv) {
  if (kp.pp != nullptr) {
    if constexpr (M::NU > 0) vec_load<M::NU>(up, inst, up_stride, upv);
    if constexpr (M::NP > 0) vec_load<M::NP>(kp.pp, inst, (int64_t)kp.pp_stride, upv + M::NU);
  } else {
    vec_load<M::NU + M::NP>(up, inst, up_stride, upv);
  }
}

// ---- small dense helpers (row-major, compile-time sizes) ------------------------------------------------
// 1/sqrt(x): v_rsq_f64 + two Newton steps (<= 2 ulp) - the Cholesky factor of Pyy below needs the reciprocal of its diagonal only
__device__ __forceinline__ double rsqrt_fast(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
  r = fma(0.5 * r, fma(-x * r, r, 1.0), r);
  return r;
}

// A = L L^T; invd (optional): the reciprocals of L's diagonal.  The diagonal through 1/sqrt (v_rsq_f64 + two Newton steps, <= 2 ulp):
// sqrt(s) = s * rsqrt(s) and the column scaling by rsqrt(s) itself - the IEEE square root and division are a dependent chain of
// ~400 cycles per column, four columns deep in the UKF's factorisation of P, which every lane of a team waits for
template <int N>
__device__ __forceinline__ void chol_lower(const double* A, double* L, double* invd = nullptr) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double s = A[j * N + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * N + k] * L[j * N + k];
    const double id = rsqrt_fast(s);
    L[j * N + j] = s * id;
    if (invd) invd[j] = id;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double t = A[i * N + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i * N + k] * L[j * N + k];
      L[i * N + j] = t * id;
    }
#pragma unroll
    for (int i = 0; i < j; ++i) L[i * N + j] = 0.0;
  }
}

// x+ = x + K (y - yp), P+ = P - K Pyy K^T with K = Pxy Pyy^-1 (kf.py:177-180); Pyy SPD -> Cholesky solve
template <int NX, int NY>
__device__ __forceinline__ void gain_update(double* x, double* P, const double* Pxy, const double* Pyy,
                                            const double* y, const double* yp) {
  double L[NY * NY], il[NY];
  chol_lower<NY>(Pyy, L, il);
  double K[NX * NY];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    // solve (L L^T) k = Pxy[i,:]^T
    double z[NY];
#pragma unroll
    for (int a = 0; a < NY; ++a) {
      double s = Pxy[i * NY + a];
#pragma unroll
      for (int b = 0; b < a; ++b) s -= L[a * NY + b] * z[b];
      z[a] = s * il[a];
    }
#pragma unroll
    for (int a = NY - 1; a >= 0; --a) {
      double s = z[a];
#pragma unroll
      for (int b = a + 1; b < NY; ++b) s -= L[b * NY + a] * K[i * NY + b];
      K[i * NY + a] = s * il[a];
    }
  }
  double KS[NX * NY];  // K Pyy
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int a = 0; a < NY; ++a) {
      double s = 0.0;
#pragma unroll
      for (int b = 0; b < NY; ++b) s += K[i * NY + b] * Pyy[b * NY + a];
      KS[i * NY + a] = s;
    }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < NY; ++a) s += K[i * NY + a] * (y[a] - yp[a]);
    x[i] += s;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      double t = 0.0;
#pragma unroll
      for (int a = 0; a < NY; ++a) t += KS[i * NY + a] * K[j * NY + a];
      P[i * NX + j] -= t;
    }
  }
}

// ---- KF / EKF -------------------------------------------------------------------------------------------
template <class M>
__device__ __forceinline__ void ekf_deriv(const double* x, const double* P, const double* u, const double* p,
                                          const double* Q, double dt, double* dx, double* dP) {
  constexpr int NX = M::NX;
  Dual<NX> xd[NX], fd[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    xd[i] = Dual<NX>(x[i]);
    xd[i].d[i] = 1.0;
  }
  M::ode(xd, u, p, dt, fd);
#pragma unroll
  for (int i = 0; i < NX; ++i) dx[i] = fd[i].v;
  // dP = F P + P F^T + Q (kf.py:98)
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      double s = Q[i * NX + j];
#pragma unroll
      for (int k = 0; k < NX; ++k) s += fd[i].d[k] * P[k * NX + j] + P[i * NX + k] * fd[j].d[k];
      dP[i * NX + j] = s;
    }
}

// LEAN (kf_multi_kernel's variant for `discretize('rk4')` with one sub-step and Q, R shared by the batch): rk4_classic without the run-time dispatch and no
// continuous-time branch - the step then fits 256 registers and two waves share a SIMD
template <class M, bool LEAN = false>
__device__ __forceinline__ void ekf_predict(const KfParams& kp, double* x, double* P, const double* u,
                                            const double* p, const double* Q) {
  constexpr int NX = M::NX;
  if constexpr (!M::DISCRETE && !LEAN) if (kp.continuous) {
    // kf.py:97-110: integrate the augmented ODE; classic RK4, n_sub steps (the reference uses CVODES)
    const double h = kp.dt / kp.n_sub;
    for (int it = 0; it < kp.n_sub; ++it) {
      double k1x[NX], k1P[NX * NX], k2x[NX], k2P[NX * NX], k3x[NX], k3P[NX * NX], k4x[NX], k4P[NX * NX];
      double xs[NX], Ps[NX * NX];
      ekf_deriv<M>(x, P, u, p, Q, kp.dt, k1x, k1P);
#pragma unroll
      for (int i = 0; i < NX; ++i) xs[i] = x[i] + 0.5 * h * k1x[i];
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) Ps[i] = P[i] + 0.5 * h * k1P[i];
      ekf_deriv<M>(xs, Ps, u, p, Q, kp.dt, k2x, k2P);
#pragma unroll
      for (int i = 0; i < NX; ++i) xs[i] = x[i] + 0.5 * h * k2x[i];
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) Ps[i] = P[i] + 0.5 * h * k2P[i];
      ekf_deriv<M>(xs, Ps, u, p, Q, kp.dt, k3x, k3P);
#pragma unroll
      for (int i = 0; i < NX; ++i) xs[i] = x[i] + h * k3x[i];
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) Ps[i] = P[i] + h * k3P[i];
      ekf_deriv<M>(xs, Ps, u, p, Q, kp.dt, k4x, k4P);
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] += h / 6.0 * (k1x[i] + 2.0 * k2x[i] + 2.0 * k3x[i] + k4x[i]);
#pragma unroll
      for (int i = 0; i < NX * NX; ++i) P[i] += h / 6.0 * (k1P[i] + 2.0 * k2P[i] + 2.0 * k3P[i] + k4P[i]);
    }
    return;
  }
  // kf.py:95-96: x- = Phi(x), P- = F P F^T + Q with F = dPhi/dx at the prior state (:91)
  Dual<NX> xd[NX], xn[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    xd[i] = Dual<NX>(x[i]);
    xd[i].d[i] = 1.0;
  }
  if constexpr (LEAN && !M::DISCRETE) rk4_classic<M>(xd, u, p, kp.dt, xn, NoExt());
  else model_step<M>(kp.erk_order, kp.n_sub, xd, u, p, kp.dt, xn);
  double FP[NX * NX];
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s += xn[i].d[k] * P[k * NX + j];
      FP[i * NX + j] = s;
    }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    x[i] = xn[i].v;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s += FP[i * NX + k] * xn[j].d[k];
      P[i * NX + j] = s + Q[i * NX + j];
    }
  }
}

template <class M>
__device__ __forceinline__ void ekf_update(const KfParams& kp, double* x, double* P, const double* y,
                                           const double* u, const double* p, const double* R, double* yp) {
  constexpr int NX = M::NX, NY = M::NY;
  Dual<NX> xd[NX], yd[NY];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    xd[i] = Dual<NX>(x[i]);
    xd[i].d[i] = 1.0;
  }
  M::meas(xd, u, p, kp.dt, yd);  // H at the predicted state (kf.py:164)
  double Pxy[NX * NY], Pyy[NY * NY];
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int a = 0; a < NY; ++a) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s += P[i * NX + k] * yd[a].d[k];
      Pxy[i * NY + a] = s;
    }
#pragma unroll
  for (int a = 0; a < NY; ++a) {
    yp[a] = yd[a].v;
#pragma unroll
    for (int b = 0; b < NY; ++b) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) s += yd[a].d[k] * Pxy[k * NY + b];
      Pyy[a * NY + b] = s + R[a * NY + b];
    }
  }
  gain_update<NX, NY>(x, P, Pxy, Pyy, y, yp);
}

// ---- UKF ------------------------------------------------------------------------------------------------
// One sigma point through the model, on FastD (hilo_ad.h: division = x * rcp_fast(y), <= 2 ulp - the scalar type the interior-point
// engine's derivative phase uses): a step's 36 right-hand sides hold 108 IEEE division sequences otherwise, a third of the one-lane
// kernel's instructions and most of the team kernel's dependent chain.  The same function in every UKF kernel.
template <class M, bool RK4 = false>
__device__ __forceinline__ void ukf_propagate(const KfParams& kp, const double* xs, const double* u, const double* p, double* xo) {
  constexpr int NX = M::NX;
  FastD xf[NX], xof[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xf[i] = FastD(xs[i]);
  if constexpr (RK4 && !M::DISCRETE)
    rk4_classic<M>(xf, u, p, kp.dt, xof, NoExt());
  else if (kp.continuous && !M::DISCRETE)
    model_step<M>(4, kp.n_sub, xf, u, p, kp.dt, xof);  // the reference integrates with CVODES
  else
    model_step<M>(kp.erk_order, kp.n_sub, xf, u, p, kp.dt, xof);
#pragma unroll
  for (int i = 0; i < NX; ++i) xo[i] = xof[i].v;
}

// The weighted sums follow the reference's accumulation order without FMA contraction: with alpha = 1e-3 the
// centre weight is ~ -1e6 and six digits cancel, so rounding order is visible in the result.
template <class M, bool LEAN = false>
__device__ __forceinline__ void ukf_predict(const KfParams& kp, double* x, double* P, double* X /*[NX][2NX+1]*/,
                                            const double* u, const double* p, const double* Q) {
#pragma clang fp contract(off)
  constexpr int NX = M::NX, NS = 2 * NX + 1;
  double L[NX * NX];
  chol_lower<NX>(P, L);  // ca.chol(P) = L^T; its column k is row k of L (kf.py:503,522-527)
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double xs[NX], xo[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double s = x[i];
      if (k >= 1 && k <= NX) s = x[i] + kp.gamma * L[(k - 1) * NX + i];
      if (k > NX) s = x[i] - kp.gamma * L[(k - 1 - NX) * NX + i];
      xs[i] = s;
    }
    ukf_propagate<M, LEAN>(kp, xs, u, p, xo);
#pragma unroll
    for (int i = 0; i < NX; ++i) X[i * NS + k] = xo[i];
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) s = s + (k == 0 ? kp.wm0 : kp.wi) * X[i * NS + k];
    x[i] = s;
  }
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      double s = Q[i * NX + j];
#pragma unroll
      for (int k = 0; k < NS; ++k)
        s = s + ((k == 0 ? kp.wc0 : kp.wi) * (X[i * NS + k] - x[i])) * (X[j * NS + k] - x[j]);
      P[i * NX + j] = s;
    }
}

template <class M>
__device__ __forceinline__ void ukf_update(const KfParams& kp, double* x, double* P, const double* X,
                                           const double* y, const double* u, const double* p, const double* R,
                                           double* yp) {
#pragma clang fp contract(off)
  constexpr int NX = M::NX, NY = M::NY, NS = 2 * NX + 1;
  double Y[NY * NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double xs[NX], ys[NY];
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = X[i * NS + k];
    M::meas(xs, u, p, kp.dt, ys);
#pragma unroll
    for (int a = 0; a < NY; ++a) Y[a * NS + k] = ys[a];
  }
#pragma unroll
  for (int a = 0; a < NY; ++a) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) s = s + (k == 0 ? kp.wm0 : kp.wi) * Y[a * NS + k];
    yp[a] = s;
  }
  double Pxy[NX * NY], Pyy[NY * NY];
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int a = 0; a < NY; ++a) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < NS; ++k)
        s = s + ((k == 0 ? kp.wc0 : kp.wi) * (X[i * NS + k] - x[i])) * (Y[a * NS + k] - yp[a]);
      Pxy[i * NY + a] = s;
    }
#pragma unroll
  for (int a = 0; a < NY; ++a)
#pragma unroll
    for (int b = 0; b < NY; ++b) {
      double s = R[a * NY + b];
#pragma unroll
      for (int k = 0; k < NS; ++k)
        s = s + ((k == 0 ? kp.wc0 : kp.wi) * (Y[a * NS + k] - yp[a])) * (Y[b * NS + k] - yp[b]);
      Pyy[a * NY + b] = s;
    }
  gain_update<NX, NY>(x, P, Pxy, Pyy, y, yp);
}

// predict + update of ONE step as one routine (the multi-step kernel's LEAN variant): the same sums in the same order as ukf_predict
// followed by ukf_update - the measured sigma points are taken when a point comes out of the model (the update measures these very
// points, kf.py:556-570), and the centred points X - x and Y - y_pred, which both covariance sums of each need, are formed ONCE in
// place of X and Y.  Peak register need: X and Y while the points are propagated, instead of X, its centred copy and Y.
// SERIAL: one sigma point at a time (a scheduling barrier after each) - 252 registers instead of 294, two waves per SIMD: faster
// when the batch fills two waves per SIMD (10.4 against 9.1 G steps/s at B = 2^20), slower when it does not (one wave per SIMD then
// has one dependent chain to issue from: 5.4 against 7.7 G at B = 65536) - csrc/hilo_kf.hip::launch_multi picks by the batch
template <class M, bool SERIAL>
__device__ __forceinline__ void ukf_step_fused(const KfParams& kp, double* x, double* P, const double* y, const double* u,
                                               const double* p, const double* Q, const double* R, double* yp) {
#pragma clang fp contract(off)
  constexpr int NX = M::NX, NY = M::NY, NS = 2 * NX + 1;
  double L[NX * NX], X[NX * NS], Y[NY * NS];
  chol_lower<NX>(P, L);
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double xs[NX], xo[NX], ys[NY];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double s = x[i];
      if (k >= 1 && k <= NX) s = x[i] + kp.gamma * L[(k - 1) * NX + i];
      if (k > NX) s = x[i] - kp.gamma * L[(k - 1 - NX) * NX + i];
      xs[i] = s;
    }
    ukf_propagate<M, true>(kp, xs, u, p, xo);
    M::meas(xo, u, p, kp.dt, ys);
#pragma unroll
    for (int i = 0; i < NX; ++i) X[i * NS + k] = xo[i];
#pragma unroll
    for (int a = 0; a < NY; ++a) Y[a * NS + k] = ys[a];
    if constexpr (SERIAL) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) s = s + (k == 0 ? kp.wm0 : kp.wi) * X[i * NS + k];
    x[i] = s;
#pragma unroll
    for (int k = 0; k < NS; ++k) X[i * NS + k] = X[i * NS + k] - s;
  }
#pragma unroll
  for (int a = 0; a < NY; ++a) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) s = s + (k == 0 ? kp.wm0 : kp.wi) * Y[a * NS + k];
    yp[a] = s;
#pragma unroll
    for (int k = 0; k < NS; ++k) Y[a * NS + k] = Y[a * NS + k] - s;
  }
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      double s = Q[i * NX + j];
#pragma unroll
      for (int k = 0; k < NS; ++k) s = s + ((k == 0 ? kp.wc0 : kp.wi) * X[i * NS + k]) * X[j * NS + k];
      P[i * NX + j] = s;
    }
  double Pxy[NX * NY], Pyy[NY * NY];
#pragma unroll
  for (int i = 0; i < NX; ++i)
#pragma unroll
    for (int a = 0; a < NY; ++a) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < NS; ++k) s = s + ((k == 0 ? kp.wc0 : kp.wi) * X[i * NS + k]) * Y[a * NS + k];
      Pxy[i * NY + a] = s;
    }
#pragma unroll
  for (int a = 0; a < NY; ++a)
#pragma unroll
    for (int b = 0; b < NY; ++b) {
      double s = R[a * NY + b];
#pragma unroll
      for (int k = 0; k < NS; ++k) s = s + ((k == 0 ? kp.wc0 : kp.wi) * Y[a * NS + k]) * Y[b * NS + k];
      Pyy[a * NY + b] = s;
    }
  gain_update<NX, NY>(x, P, Pxy, Pyy, y, yp);
}

// ---- kernels --------------------------------------------------------------------------------------------
// MODE 0 = predict, 1 = update, 2 = fused step.  UKF template flag selects sigma-point arithmetic.
#ifndef HILO_KF_WAVES
#define HILO_KF_WAVES 0   // 0: let the compiler choose the occupancy (measured best)
#endif
#if HILO_KF_WAVES > 0
#define KF_OCC __attribute__((amdgpu_waves_per_eu(HILO_KF_WAVES, HILO_KF_WAVES)))
#else
#define KF_OCC
#endif
template <class M, bool UKF, int MODE>
__device__ __forceinline__ void kf_body(const KfParams& kp, int64_t batch, const double* __restrict__ in_tile,
                                        const double* __restrict__ y, const double* __restrict__ up,
                                        int64_t up_stride, const double* __restrict__ Q,
                                        int64_t q_stride, const double* __restrict__ R,
                                        int64_t r_stride, double* __restrict__ out_tile,
                                        double* __restrict__ y_pred, int ipw) {
  constexpr int NX = M::NX, NY = M::NY, NUP = M::NU + M::NP, NS = 2 * NX + 1;
  constexpr int XP = NX * (NX + 1);                          // [x|P]
  constexpr int PRED = UKF ? NX * (1 + NX + NS) : XP;        // predict output / update input
  constexpr int IN_ROW = (MODE == 1) ? PRED : XP;
  constexpr int OUT_ROW = (MODE == 0) ? PRED : XP;
  constexpr int BIG = IN_ROW > OUT_ROW ? IN_ROW : OUT_ROW;
  __shared__ double lds[KF_TPB * TilePitch<BIG>::value];

  // `ipw` instances per workgroup (<= KF_TPB): a small batch is spread over all compute units (16 instances per wave at the
  // BASELINE size 4096 -> 256 workgroups), a large one fills every lane
  const int64_t first = (int64_t)blockIdx.x * ipw;
  const int count = (int)((batch - first) < ipw ? (batch - first) : ipw);
  const int64_t inst = first + threadIdx.x;
  const bool active = (int)threadIdx.x < count;

  double tin[IN_ROW];
  tile_load<IN_ROW, KF_TPB>(in_tile, first, count, lds, tin);

  double x[NX], P[NX * NX], X[UKF ? NX * NS : 1], upv[MaxOne<NUP>::v], yv[NY], ypv[NY];
  constexpr int W_IN = IN_ROW / NX;
  if (active) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      x[i] = tin[i * W_IN];
#pragma unroll
      for (int j = 0; j < NX; ++j) P[i * NX + j] = tin[i * W_IN + 1 + j];
      if constexpr (UKF && MODE == 1) {
#pragma unroll
        for (int k = 0; k < NS; ++k) X[i * NS + k] = tin[i * W_IN + 1 + NX + k];
      }
    }
    if constexpr (NUP > 0) vec_load<NUP>(up, inst, up_stride, upv);
    const double* u = upv;
    const double* p = upv + M::NU;
    if constexpr (MODE != 1) {
      double Qv[NX * NX];
      vec_load<NX * NX>(Q, inst, q_stride, Qv);
      if constexpr (UKF) ukf_predict<M>(kp, x, P, X, u, p, Qv);
      else ekf_predict<M>(kp, x, P, u, p, Qv);
    }
    if constexpr (MODE != 0) {
      double Rv[NY * NY];
      vec_load<NY * NY>(R, inst, r_stride, Rv);
      vec_load<NY>(y, inst, NY, yv);
      if constexpr (UKF) ukf_update<M>(kp, x, P, X, yv, u, p, Rv, ypv);
      else ekf_update<M>(kp, x, P, yv, u, p, Rv, ypv);
#pragma unroll
      for (int a = 0; a < NY; ++a) y_pred[inst * NY + a] = ypv[a];
    }
  }
  double tout[OUT_ROW];
  constexpr int W_OUT = OUT_ROW / NX;
  if (active) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      tout[i * W_OUT] = x[i];
#pragma unroll
      for (int j = 0; j < NX; ++j) tout[i * W_OUT + 1 + j] = P[i * NX + j];
      if constexpr (UKF && MODE == 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k) tout[i * W_OUT + 1 + NX + k] = X[i * NS + k];
      }
    }
  }
  tile_store<OUT_ROW, KF_TPB>(out_tile, first, count, lds, tout);
}

template <class M, bool UKF, int MODE>
__global__ __launch_bounds__(KF_TPB) KF_OCC void kf_kernel(KfParams kp, int64_t batch, const double* __restrict__ in_tile,
                                                    const double* __restrict__ y, const double* __restrict__ up,
                                                    int64_t up_stride, const double* __restrict__ Q,
                                                    int64_t q_stride, const double* __restrict__ R,
                                                    int64_t r_stride, double* __restrict__ out_tile,
                                                    double* __restrict__ y_pred, int ipw) {
  kf_body<M, UKF, MODE>(kp, batch, in_tile, y, up, up_stride, Q, q_stride, R, r_stride, out_tile, y_pred, ipw);
}

// ---- several filter steps in ONE launch: `self._function.mapaccum(steps)` of the reference (kf.py:296-306) -----------------------
// The packed tile is loaded once, `steps` fused predict + update steps run in registers - measurements y [steps][B][ny], inputs
// either the same every step or [steps][B][nu+np] (up_step = elements between two steps) - and the tile of every step (the
// reference's accumulated output) or only the last one goes back; y_pred [steps][B][ny].  A filter step at the BASELINE batch
// (4096 instances = 1.6 MB) is launch-latency bound; K steps per launch divide that latency by K and, with only the last tile
// written, move 8 (ny + nu + np + ny) bytes per step instead of the tile both ways.
template <class M, bool UKF, int LEAN = 0>      // LEAN: 1 = the common recipe's variant, 2 = (UKF) with the sigma points one at a time
__device__ __forceinline__ void kf_multi_body(const KfParams& kp, int64_t batch, int steps, const double* __restrict__ in_tile,
                                              const double* __restrict__ y, const double* __restrict__ up, int64_t up_stride,
                                              int64_t up_step, const double* __restrict__ Q, int64_t q_stride,
                                              const double* __restrict__ R, int64_t r_stride, double* __restrict__ out_tile,
                                              int64_t out_step, double* __restrict__ y_pred, int ipw) {
  constexpr int NX = M::NX, NY = M::NY, NUP = M::NU + M::NP, NS = 2 * NX + 1;
  constexpr int XP = NX * (NX + 1), W = NX + 1;
  __shared__ double lds[KF_TPB * TilePitch<XP>::value];
  const int64_t first = (int64_t)blockIdx.x * ipw;
  const int count = (int)((batch - first) < ipw ? (batch - first) : ipw);
  const int64_t inst = first + threadIdx.x;
  const bool active = (int)threadIdx.x < count;
  double tin[XP];
  tile_load<XP, KF_TPB>(in_tile, first, count, lds, tin);
  double x[NX], P[NX * NX], X[UKF ? NX * NS : 1], upv[MaxOne<NUP>::v], yv[NY], ypv[NY], Qr[LEAN ? 1 : NX * NX], Rr[LEAN ? 1 : NY * NY];
  // LEAN: Q and R are shared by the batch (strides 0) and read where a step uses them - wave-uniform addresses, no registers held
  // over the Runge-Kutta stages
  const double* Qv = LEAN ? Q : Qr;
  const double* Rv = LEAN ? R : Rr;
  if (active) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      x[i] = tin[i * W];
#pragma unroll
      for (int j = 0; j < NX; ++j) P[i * NX + j] = tin[i * W + 1 + j];
    }
    if constexpr (!LEAN) {
      vec_load<NX * NX>(Q, inst, q_stride, Qr);
      vec_load<NY * NY>(R, inst, r_stride, Rr);
    }
    if constexpr (NUP > 0) kf_load_up<M>(kp, up, inst, up_stride, upv);
  }
  for (int s = 0; s < steps; ++s) {
    if (active) {
      if constexpr (NUP > 0) {
        if (s > 0 && up_step != 0) kf_load_up<M>(kp, up + (int64_t)s * up_step, inst, up_stride, upv);
      }
      const double* u = upv;
      const double* p = upv + M::NU;
      vec_load<NY>(y + (int64_t)s * batch * NY, inst, NY, yv);
      if constexpr (UKF && LEAN && !M::DISCRETE) {
        ukf_step_fused<M, LEAN == 2>(kp, x, P, yv, u, p, Qv, Rv, ypv);
      } else if constexpr (UKF) {
        ukf_predict<M, (LEAN != 0)>(kp, x, P, X, u, p, Qv);
        ukf_update<M>(kp, x, P, X, yv, u, p, Rv, ypv);
      } else {
        ekf_predict<M, (LEAN != 0)>(kp, x, P, u, p, Qv);
        ekf_update<M>(kp, x, P, yv, u, p, Rv, ypv);
      }
#pragma unroll
      for (int a = 0; a < NY; ++a) y_pred[((int64_t)s * batch + inst) * NY + a] = ypv[a];
    }
    if (out_step != 0 || s == steps - 1) {   // wave-uniform: every lane takes part in the staged store
      double tout[XP];
      if (active) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          tout[i * W] = x[i];
#pragma unroll
          for (int j = 0; j < NX; ++j) tout[i * W + 1 + j] = P[i * NX + j];
        }
      }
      tile_store<XP, KF_TPB>(out_tile + (int64_t)s * out_step, first, count, lds, tout);
    }
  }
}

template <class M, bool UKF, int LEAN = 0>
__global__ __launch_bounds__(KF_TPB) KF_OCC void kf_multi_kernel(KfParams kp, int64_t batch, int steps,
                                                                 const double* __restrict__ in_tile, const double* __restrict__ y,
                                                                 const double* __restrict__ up, int64_t up_stride, int64_t up_step,
                                                                 const double* __restrict__ Q, int64_t q_stride,
                                                                 const double* __restrict__ R, int64_t r_stride,
                                                                 double* __restrict__ out_tile, int64_t out_step,
                                                                 double* __restrict__ y_pred, int ipw) {
  kf_multi_body<M, UKF, LEAN>(kp, batch, steps, in_tile, y, up, up_stride, up_step, Q, q_stride, R, r_stride, out_tile, out_step, y_pred, ipw);
}

// the EKF's LEAN variant with at least two waves per SIMD asked of the register allocator (chemostat4 lands on 258 registers without)
template <class M>
__global__ __launch_bounds__(KF_TPB) __attribute__((amdgpu_waves_per_eu(2))) void ekf_multi_lean_kernel(
    KfParams kp, int64_t batch, int steps, const double* __restrict__ in_tile, const double* __restrict__ y,
    const double* __restrict__ up, int64_t up_stride, int64_t up_step, const double* __restrict__ Q, int64_t q_stride,
    const double* __restrict__ R, int64_t r_stride, double* __restrict__ out_tile, int64_t out_step, double* __restrict__ y_pred,
    int ipw) {
  kf_multi_body<M, false, 1>(kp, batch, steps, in_tile, y, up, up_stride, up_step, Q, q_stride, R, r_stride, out_tile, out_step, y_pred, ipw);
}

// ---- one filter instance on a TEAM of lanes: small batches (the BASELINE's B = 4096) ------------------------------------------------
// With one instance per lane a launch lasts as long as ONE lane needs for its ~3 k (EKF) / ~9 k (UKF) dependent fp64 instructions per
// step, however few instances there are.  Here T lanes share an instance - the EKF's Jacobian columns (one forward-mode direction
// per lane, T = NX rounded up to a power of two), the UKF's sigma points (one point per lane, T >= 2 NX + 1) - and the matrix algebra
// is spread entry by entry; [x | P], the Jacobian / the sigma points and the small update matrices are staged in LDS (a team lives
// inside one wave, a workgroup is one wave: `__syncthreads()` is an LDS wait, not a barrier across waves).  Every lane runs every
// phase (lanes without work of their own repeat a neighbour's, only the LDS stores are predicated): no lane-dependent branch
// encloses a loop.  The arithmetic of each entry is that of the one-lane kernels above (same order of accumulation, the UKF's
// sums without FMA contraction).  The EKF of a CONTINUOUS model (augmented ODE, ekf_deriv) stays on the one-lane kernel.
#ifndef HILO_KF_DBG
#define HILO_KF_DBG 0   // developer builds: 1 = skip the model evaluation, 2 = skip the gain, 4 = skip the matrix phases (timing only)
#endif
#ifdef HILO_KF_PROF   // developer builds: cycle stamps of the phases of a team step (team 0 of workgroup 0, last stamps win)
__device__ unsigned long long hilo_kf_prof[16];
#define KF_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) hilo_kf_prof[k] = __builtin_readcyclecounter(); } while (0)
#else
#define KF_STAMP(k) do { } while (0)
#endif
constexpr int kf_pow2(int n) { int p = 1; while (p < n) p *= 2; return p; }

template <class M, bool UKF>
struct KfTeam {
  static constexpr int NX = M::NX, NY = M::NY, NS = 2 * NX + 1;
  static constexpr int T = kf_pow2(UKF ? NS : NX);
  static constexpr bool OK = T <= KF_TPB;
  static constexpr int TEAMS = OK ? KF_TPB / T : 1;
  static constexpr int NA = UKF ? NX * NS : NX * NX;   // EKF: F = dPhi/dx        UKF: propagated sigma points X
  static constexpr int NB = UKF ? NY * NS : NX * NX;   // EKF: F P                UKF: measured sigma points Y
  static constexpr int NH = UKF ? 0 : NY * NX;         // EKF: H = dh/dx
  static constexpr int O_X = 0, O_P = O_X + NX, O_A = O_P + NX * NX, O_B = O_A + NA, O_H = O_B + NB, O_PXY = O_H + NH,
                       O_PYY = O_PXY + NX * NY, O_K = O_PYY + NY * NY, O_KS = O_K + NX * NY, O_YP = O_KS + NX * NY,
                       USED = O_YP + NY, SIZE = USED | 1;   // odd pitch: the teams of a wave start in different banks
};

// row i of K = Pxy Pyy^-1 and of K Pyy, and K[i,:] (y - yp) (kf.py:177-180): gain_update's Cholesky solve for one row, with the
// reciprocals of the factor's diagonal (two dependent rsq instead of two square roots and six divisions in a row)
template <int NX, int NY>
__device__ __forceinline__ void gain_row(const double* Pxy_i, const double* Pyy, const double* y, const double* yp, double* K_i,
                                         double* KS_i, double* dx) {
  double L[NY * NY], id[NY], z[NY];
#pragma unroll
  for (int j = 0; j < NY; ++j) {
    double s = Pyy[j * NY + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * NY + k] * L[j * NY + k];
    id[j] = rsqrt_fast(s);
#pragma unroll
    for (int i = j + 1; i < NY; ++i) {
      double t = Pyy[i * NY + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i * NY + k] * L[j * NY + k];
      L[i * NY + j] = t * id[j];
    }
  }
#pragma unroll
  for (int a = 0; a < NY; ++a) {
    double s = Pxy_i[a];
#pragma unroll
    for (int b = 0; b < a; ++b) s -= L[a * NY + b] * z[b];
    z[a] = s * id[a];
  }
#pragma unroll
  for (int a = NY - 1; a >= 0; --a) {
    double s = z[a];
#pragma unroll
    for (int b = a + 1; b < NY; ++b) s -= L[b * NY + a] * K_i[b];
    K_i[a] = s * id[a];
  }
#pragma unroll
  for (int a = 0; a < NY; ++a) {
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < NY; ++b) s += K_i[b] * Pyy[b * NY + a];
    KS_i[a] = s;
  }
  double s = 0.0;
#pragma unroll
  for (int a = 0; a < NY; ++a) s += K_i[a] * (y[a] - yp[a]);
  *dx = s;
}

// phases shared by both filters: rows of the gain (lanes i < NX), then P -= (K Pyy) K^T entry by entry
template <class M, bool UKF>
__device__ __forceinline__ void team_gain(double* __restrict__ sm, int t, const double* yv) {
  using D = KfTeam<M, UKF>;
  constexpr int NX = D::NX, NY = D::NY, T = D::T;
  {
    const int i = t < NX ? t : NX - 1;
    double Pxy_i[NY], Pyy[NY * NY], yp[NY], K_i[NY], KS_i[NY], dx;
#pragma unroll
    for (int a = 0; a < NY; ++a) { Pxy_i[a] = sm[D::O_PXY + i * NY + a]; yp[a] = sm[D::O_YP + a]; }
#pragma unroll
    for (int a = 0; a < NY * NY; ++a) Pyy[a] = sm[D::O_PYY + a];
    gain_row<NX, NY>(Pxy_i, Pyy, yv, yp, K_i, KS_i, &dx);
    const double xi = sm[D::O_X + i] + dx;
    if (t < NX) {
      sm[D::O_X + i] = xi;
#pragma unroll
      for (int a = 0; a < NY; ++a) { sm[D::O_K + i * NY + a] = K_i[a]; sm[D::O_KS + i * NY + a] = KS_i[a]; }
    }
  }
  __syncthreads();
  KF_STAMP(4);
#pragma unroll
  for (int m = 0; m < (NX * NX + T - 1) / T; ++m) {
    const int e0 = m * T + t, e = e0 < NX * NX ? e0 : NX * NX - 1, i = e / NX, j = e - i * NX;
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < NY; ++a) s += sm[D::O_KS + i * NY + a] * sm[D::O_K + j * NY + a];
    const double v = sm[D::O_P + e] - s;
    if (e0 < NX * NX) sm[D::O_P + e] = v;
  }
  __syncthreads();
}

template <class M>
__device__ __forceinline__ void team_ekf_step(const KfParams& kp, double* __restrict__ sm, int t, const double* u, const double* p,
                                              const double* Qe, const double* Re, const double* yv) {
  using D = KfTeam<M, false>;
  constexpr int NX = D::NX, NY = D::NY, T = D::T;
  KF_STAMP(0);
  {
    // column `dir` of F = dPhi/dx at the prior state (kf.py:91) and of H = dh/dx at the predicted one (:164)
    const int dir = t < NX ? t : NX - 1;
    Dual<1> xd[NX], xn[NX], yd[NY];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      xd[i].v = sm[D::O_X + i];
      xd[i].d[0] = i == dir ? 1.0 : 0.0;
    }
    if constexpr (HILO_KF_DBG & 1) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xn[i] = xd[i];
    } else {
      model_step<M>(kp.erk_order, kp.n_sub, xd, u, p, kp.dt, xn);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      xd[i].v = xn[i].v;
      xd[i].d[0] = i == dir ? 1.0 : 0.0;
    }
    M::meas(xd, u, p, kp.dt, yd);
    if (t < NX) {
#pragma unroll
      for (int i = 0; i < NX; ++i) sm[D::O_A + i * NX + dir] = xn[i].d[0];
#pragma unroll
      for (int a = 0; a < NY; ++a) sm[D::O_H + a * NX + dir] = yd[a].d[0];
    }
    if (t == 0) {
#pragma unroll
      for (int i = 0; i < NX; ++i) sm[D::O_X + i] = xn[i].v;
#pragma unroll
      for (int a = 0; a < NY; ++a) sm[D::O_YP + a] = yd[a].v;
    }
  }
  __syncthreads();
  KF_STAMP(1);
  if constexpr (HILO_KF_DBG & 4) { if (!(HILO_KF_DBG & 2)) team_gain<M, false>(sm, t, yv); return; }
  // P- = F P F^T + Q (kf.py:95-96)
  if constexpr (T % NX == 0) {
    // every entry of a lane lies in column j = t mod NX: w = P F[j,:]^T once, then P-[i,j] = F[i,:] w - no F P staged in between
    const int j = t % NX;
    double w[NX], Pn[(NX * NX + T - 1) / T];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      double a = 0.0;
#pragma unroll
      for (int l = 0; l < NX; ++l) a += sm[D::O_P + k * NX + l] * sm[D::O_A + j * NX + l];
      w[k] = a;
    }
#pragma unroll
    for (int m = 0; m < (NX * NX + T - 1) / T; ++m) {
      const int e0 = m * T + t, e = e0 < NX * NX ? e0 : NX * NX - 1, i = e / NX;
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) a += sm[D::O_A + i * NX + k] * w[k];
      Pn[m] = a + Qe[m];
    }
#pragma unroll
    for (int m = 0; m < (NX * NX + T - 1) / T; ++m) {       // (all reads of P are done: same wave, program order)
      const int e0 = m * T + t;
      if (e0 < NX * NX) sm[D::O_P + e0] = Pn[m];
    }
  } else {
#pragma unroll
    for (int m = 0; m < (NX * NX + T - 1) / T; ++m) {
      const int e0 = m * T + t, e = e0 < NX * NX ? e0 : NX * NX - 1, i = e / NX, j = e - i * NX;
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) a += sm[D::O_A + i * NX + k] * sm[D::O_P + k * NX + j];
      if (e0 < NX * NX) sm[D::O_B + e] = a;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < (NX * NX + T - 1) / T; ++m) {
      const int e0 = m * T + t, e = e0 < NX * NX ? e0 : NX * NX - 1, i = e / NX, j = e - i * NX;
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) a += sm[D::O_B + i * NX + k] * sm[D::O_A + j * NX + k];
      if (e0 < NX * NX) sm[D::O_P + e] = a + Qe[m];
    }
  }
  __syncthreads();
  KF_STAMP(2);
  // Pxy = P- H^T and Pyy = H P- H^T + R (kf.py:171-176) in one phase: an entry of Pyy recomputes its column of P- H^T
  constexpr int NE2 = NX * NY + NY * NY;
#pragma unroll
  for (int m = 0; m < (NE2 + T - 1) / T; ++m) {
    const int e0 = m * T + t, e = e0 < NE2 ? e0 : NE2 - 1;
    if (e < NX * NY) {
      const int i = e / NY, a = e - i * NY;
      double q = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) q += sm[D::O_P + i * NX + k] * sm[D::O_H + a * NX + k];
      if (e0 < NE2) sm[D::O_PXY + e] = q;
    } else {
      const int f = e - NX * NY, a = f / NY, b = f - a * NY;
      double q = 0.0;
#pragma unroll
      for (int k = 0; k < NX; ++k) {
        double c = 0.0;
#pragma unroll
        for (int l = 0; l < NX; ++l) c += sm[D::O_P + k * NX + l] * sm[D::O_H + b * NX + l];
        q += sm[D::O_H + a * NX + k] * c;
      }
      if (e0 < NE2) sm[D::O_PYY + f] = q + Re[m];
    }
  }
  __syncthreads();
  KF_STAMP(3);
  if constexpr (!(HILO_KF_DBG & 2)) team_gain<M, false>(sm, t, yv);
  KF_STAMP(5);
}

template <class M>
__device__ __forceinline__ void team_ukf_step(const KfParams& kp, double* __restrict__ sm, int t, const double* u, const double* p,
                                              const double* Qe, const double* Re, const double* yv) {
#pragma clang fp contract(off)
  using D = KfTeam<M, true>;
  constexpr int NX = D::NX, NY = D::NY, NS = D::NS, T = D::T;
  KF_STAMP(0);
  {
    // sigma point k = lane (kf.py:503, :522-527), propagated (:529-533) and measured (:570-575)
    const int k = t < NS ? t : NS - 1;
    double x[NX], P[NX * NX], L[NX * NX], xs[NX], xo[NX], ys[NY];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = sm[D::O_X + i];
#pragma unroll
    for (int i = 0; i < NX * NX; ++i) P[i] = sm[D::O_P + i];
    chol_lower<NX>(P, L);
    const int row = k == 0 ? 0 : (k <= NX ? k - 1 : k - 1 - NX);
    const double sg = k == 0 ? 0.0 : (k <= NX ? kp.gamma : -kp.gamma);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double li = L[i];
#pragma unroll
      for (int r = 1; r < NX; ++r) li = row == r ? L[r * NX + i] : li;
      // (x + gamma l and x - gamma l: x + (-gamma) l rounds like x - gamma l; the centre point is x itself)
      xs[i] = k == 0 ? x[i] : x[i] + sg * li;
    }
    ukf_propagate<M>(kp, xs, u, p, xo);
    M::meas(xo, u, p, kp.dt, ys);
    if (t < NS) {
#pragma unroll
      for (int i = 0; i < NX; ++i) sm[D::O_A + i * NS + k] = xo[i];
#pragma unroll
      for (int a = 0; a < NY; ++a) sm[D::O_B + a * NS + k] = ys[a];
    }
  }
  __syncthreads();
  KF_STAMP(1);
  // weighted means (kf.py:535-541, :577-583)
#pragma unroll
  for (int m = 0; m < (NX + NY + T - 1) / T; ++m) {
    const int e0 = m * T + t, e = e0 < NX + NY ? e0 : NX + NY - 1;
    const int base = e < NX ? D::O_A + e * NS : D::O_B + (e - NX) * NS;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k) s = s + (k == 0 ? kp.wm0 : kp.wi) * sm[base + k];
    if (e0 < NX + NY) sm[e < NX ? D::O_X + e : D::O_YP + (e - NX)] = s;
  }
  __syncthreads();
  KF_STAMP(2);
  // P- (:543-548), Pxy and Pyy (:585-596): one list of NX^2 + NX NY + NY^2 entries
  constexpr int NE = NX * NX + NX * NY + NY * NY;
#pragma unroll
  for (int m = 0; m < (NE + T - 1) / T; ++m) {
    const int e0 = m * T + t, e = e0 < NE ? e0 : NE - 1;
    int ra, rb, ma, mb, dst;          // rows of the two factors in sm, their means
    double s;
    if (e < NX * NX) {
      const int i = e / NX, j = e - i * NX;
      ra = D::O_A + i * NS; rb = D::O_A + j * NS; ma = D::O_X + i; mb = D::O_X + j; dst = D::O_P + e; s = Qe[m];
    } else if (e < NX * NX + NX * NY) {
      const int f = e - NX * NX, i = f / NY, a = f - i * NY;
      ra = D::O_A + i * NS; rb = D::O_B + a * NS; ma = D::O_X + i; mb = D::O_YP + a; dst = D::O_PXY + f; s = 0.0;
    } else {
      const int f = e - NX * NX - NX * NY, a = f / NY, b = f - a * NY;
      ra = D::O_B + a * NS; rb = D::O_B + b * NS; ma = D::O_YP + a; mb = D::O_YP + b; dst = D::O_PYY + f; s = Re[m];
    }
    const double xa = sm[ma], xb = sm[mb];
#pragma unroll
    for (int k = 0; k < NS; ++k) s = s + ((k == 0 ? kp.wc0 : kp.wi) * (sm[ra + k] - xa)) * (sm[rb + k] - xb);
    if (e0 < NE) sm[dst] = s;
  }
  __syncthreads();
  KF_STAMP(3);
  team_gain<M, true>(sm, t, yv);
  KF_STAMP(5);
}

// arguments of kf_multi_kernel without `ipw`: a workgroup (one wave) holds KfTeam::TEAMS instances
template <class M, bool UKF>
__device__ __forceinline__ void kf_team_body(const KfParams& kp, int64_t batch, int steps, const double* __restrict__ in_tile,
                                             const double* __restrict__ y, const double* __restrict__ up, int64_t up_stride,
                                             int64_t up_step, const double* __restrict__ Q, int64_t q_stride,
                                             const double* __restrict__ R, int64_t r_stride, double* __restrict__ out_tile,
                                             int64_t out_step, double* __restrict__ y_pred) {
  using D = KfTeam<M, UKF>;
  constexpr int NX = D::NX, NY = D::NY, NUP = M::NU + M::NP, T = D::T, XP = NX * (NX + 1), W = NX + 1;
  constexpr int NE = UKF ? NX * NX + NX * NY + NY * NY : NX * NX;       // entries with a Q (R) term, per lane: Qe[m] (Re[m])
  constexpr int NER = UKF ? NE : NX * NY + NY * NY, OFFR = UKF ? NX * NX + NX * NY : NX * NY;   // the list whose tail carries R
  constexpr int MQ = (NE + T - 1) / T, MR = (NER + T - 1) / T;
  __shared__ double lds[D::TEAMS * D::SIZE];
  const int team = threadIdx.x / T, t = threadIdx.x - team * T;
  const int64_t inst0 = (int64_t)blockIdx.x * D::TEAMS + team;
  const bool valid = inst0 < batch;
  const int64_t inst = valid ? inst0 : batch - 1;      // a team past the end repeats the last instance and stores nothing
  double* sm = lds + team * D::SIZE;
#pragma unroll
  for (int m = 0; m < (XP + T - 1) / T; ++m) {
    const int e0 = m * T + t, e = e0 < XP ? e0 : XP - 1, i = e / W, c = e - i * W;
    const double v = in_tile[inst * XP + e];
    if (e0 < XP) sm[c == 0 ? D::O_X + i : D::O_P + i * NX + c - 1] = v;
  }
  double upv[MaxOne<NUP>::v], yv[NY], Qe[MQ], Re[MR];
#pragma unroll
  for (int m = 0; m < MQ; ++m) {
    const int e0 = m * T + t, e = e0 < NE ? e0 : NE - 1;
    Qe[m] = e < NX * NX ? Q[inst * q_stride + e] : 0.0;
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int e0 = m * T + t, e = e0 < NER ? e0 : NER - 1, f = e - OFFR;
    Re[m] = f >= 0 ? R[inst * r_stride + f] : 0.0;
  }
  if constexpr (NUP > 0) kf_load_up<M>(kp, up, inst, up_stride, upv);
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    if constexpr (NUP > 0) {
      if (s > 0 && up_step != 0) kf_load_up<M>(kp, up + (int64_t)s * up_step, inst, up_stride, upv);
    }
    const double* u = upv;
    const double* p = upv + M::NU;
    vec_load<NY>(y + (int64_t)s * batch * NY, inst, NY, yv);
    KF_STAMP(8);
    if constexpr (UKF) team_ukf_step<M>(kp, sm, t, u, p, Qe, Re, yv);
    else team_ekf_step<M>(kp, sm, t, u, p, Qe, Re, yv);
    KF_STAMP(9);
    if (valid && t < NY) y_pred[((int64_t)s * batch + inst) * NY + t] = sm[D::O_YP + t];
    if constexpr (NY > T) {
      if (valid && t == 0)
        for (int a = T; a < NY; ++a) y_pred[((int64_t)s * batch + inst) * NY + a] = sm[D::O_YP + a];
    }
    if (out_step != 0 || s == steps - 1) {
#pragma unroll
      for (int m = 0; m < (XP + T - 1) / T; ++m) {
        const int e0 = m * T + t, e = e0 < XP ? e0 : XP - 1, i = e / W, c = e - i * W;
        const double v = sm[c == 0 ? D::O_X + i : D::O_P + i * NX + c - 1];
        if (valid && e0 < XP) out_tile[(int64_t)s * out_step + inst * XP + e] = v;
      }
    }
    KF_STAMP(10);
  }
}

template <class M, bool UKF>
__global__ __launch_bounds__(KF_TPB) void kf_team_kernel(KfParams kp, int64_t batch, int steps, const double* __restrict__ in_tile,
                                                         const double* __restrict__ y, const double* __restrict__ up,
                                                         int64_t up_stride, int64_t up_step, const double* __restrict__ Q,
                                                         int64_t q_stride, const double* __restrict__ R, int64_t r_stride,
                                                         double* __restrict__ out_tile, int64_t out_step,
                                                         double* __restrict__ y_pred) {
  if constexpr (KfTeam<M, UKF>::OK)
    kf_team_body<M, UKF>(kp, batch, steps, in_tile, y, up, up_stride, up_step, Q, q_stride, R, r_stride, out_tile, out_step, y_pred);
}

// ---- particle filter (hilo_mpc/modules/estimator/pf.py) ----------------------------------------------------------------------
// The function the reference assembles at setup() (`_propagate_particles` :103-146, `_evaluate_likelihood` :148-166, `setup`
// :300-318) and calls once per estimate (:372):
//     X_prop = Phi(X, u, p) + w        Y = h(X_prop, u, p) + v        q_j = normpdf(Y_j; y, sqrt(R)) / sum_j normpdf(...)
// One workgroup per filter, the particles strided over its threads (the sum over the particles is a workgroup reduction).
// Layout: particle-major [N][nx] - the memory order of CasADi's column-major nx x N matrix.  A model without measurement
// equations measures all its states (:131-134).  For one measurement q is the reference's; for several (where the reference's
// flattened n_y x N weight matrix is no probability vector) it is the joint likelihood of independent measurements, R diagonal.
constexpr int PF_TPB = 256;

template <class M>
__device__ __forceinline__ void pf_body(const KfParams& kp, int n, const double* __restrict__ X, const double* __restrict__ y,
                                        const double* __restrict__ up, const double* __restrict__ w,
                                        const double* __restrict__ v, const double* __restrict__ R, double* __restrict__ Xp,
                                        double* __restrict__ Y, double* __restrict__ q) {
  constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NYE = M::NY > 0 ? M::NY : M::NX;
  __shared__ double red[PF_TPB];
  const int tid = threadIdx.x;
  double u[MaxOne<NU>::v], p[MaxOne<NP>::v], sig[NYE], ym[NYE];
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = up[i];
#pragma unroll
  for (int i = 0; i < NP; ++i) p[i] = up[NU + i];
#pragma unroll
  for (int a = 0; a < NYE; ++a) { sig[a] = ::sqrt(R[a * NYE + a]); ym[a] = y[a]; }
  double part = 0.0;
  for (int j = tid; j < n; j += PF_TPB) {
    double xs[NX], xo[NX], yy[NYE];
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = X[(int64_t)j * NX + i];
    if (kp.continuous && !M::DISCRETE)
      model_step<M>(4, kp.n_sub, xs, u, p, kp.dt, xo);  // the reference integrates with CVODES
    else
      model_step<M>(kp.erk_order, kp.n_sub, xs, u, p, kp.dt, xo);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      xo[i] += w[(int64_t)j * NX + i];
      Xp[(int64_t)j * NX + i] = xo[i];
    }
    if constexpr (M::NY > 0) {
      M::meas(xo, u, p, kp.dt, yy);
    } else {
#pragma unroll
      for (int a = 0; a < NYE; ++a) yy[a] = xo[a];
    }
    double l = 1.0;
#pragma unroll
    for (int a = 0; a < NYE; ++a) {
      yy[a] += v[(int64_t)j * NYE + a];
      Y[(int64_t)j * NYE + a] = yy[a];
      const double z = (yy[a] - ym[a]) / sig[a];
      l *= ::exp(-0.5 * (z * z)) / (2.5066282746310002 * sig[a]);   // pf.py:99, sqrt(2 pi)
    }
    q[j] = l;
    part += l;
  }
  red[tid] = part;
  __syncthreads();
  for (int o = PF_TPB / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double total = red[0];
  for (int j = tid; j < n; j += PF_TPB) q[j] = q[j] / total;   // `q /= ca.sum2(q)` (:158)
}

template <class M>
__global__ __launch_bounds__(PF_TPB) void pf_kernel(KfParams kp, int n, const double* __restrict__ X, const double* __restrict__ y,
                                                    const double* __restrict__ up, int64_t up_stride,
                                                    const double* __restrict__ w, const double* __restrict__ v,
                                                    const double* __restrict__ R, int64_t r_stride, double* __restrict__ Xp,
                                                    double* __restrict__ Y, double* __restrict__ q) {
  constexpr int NX = M::NX, NYE = M::NY > 0 ? M::NY : M::NX;
  const int64_t b = blockIdx.x;
  pf_body<M>(kp, n, X + b * n * NX, y + b * NYE, up + b * up_stride, w + b * n * NX, v + b * n * NYE, R + b * r_stride,
             Xp + b * n * NX, Y + b * n * NYE, q + b * n);
}

// Resampling (`np.random.choice(N, size=N, replace=True, p=q)`, pf.py:404-406) with the uniform draws supplied by the caller:
// numpy's algorithm - cdf = cumsum(q) / cdf[-1], index = searchsorted(cdf, u, side='right') - then the gather of the propagated
// particles and their measurements.  One workgroup per filter; the cdf lives in LDS (n <= 8192).
__global__ __launch_bounds__(PF_TPB) void pf_resample_kernel(int n, int nx, int ny, const double* __restrict__ Xp,
                                                             const double* __restrict__ Y, const double* __restrict__ q,
                                                             const double* __restrict__ uni, double* __restrict__ X,
                                                             double* __restrict__ Yr, int* __restrict__ idx) {
  extern __shared__ double cdf[];          // [n] then PF_TPB partial sums
  double* part = cdf + n;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  Xp += b * n * nx; Y += b * n * ny; q += b * n; uni += b * n; X += b * n * nx; Yr += b * n * ny; idx += b * n;
  const int chunk = (n + PF_TPB - 1) / PF_TPB, lo = tid * chunk, hi = lo + chunk < n ? lo + chunk : n;
  double s = 0.0;
  for (int j = lo; j < hi; ++j) { s += q[j]; cdf[j] = s; }      // running sums inside the thread's chunk
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    double acc = 0.0;
    for (int t = 0; t < PF_TPB; ++t) { const double c = part[t]; part[t] = acc; acc += c; }
  }
  __syncthreads();
  const double off = part[tid];
  for (int j = lo; j < hi; ++j) cdf[j] += off;
  __syncthreads();
  const double last = cdf[n - 1];
  for (int j = tid; j < n; j += PF_TPB) {
    const double uj = uni[j] * last;       // u < cdf[i] / last  <=>  u * last < cdf[i]
    int a = 0, c = n;                      // first i with cdf[i] > uj
    while (a < c) {
      const int m = (a + c) >> 1;
      if (cdf[m] > uj) c = m; else a = m + 1;
    }
    const int i = a < n ? a : n - 1;
    idx[j] = i;
    for (int k = 0; k < nx; ++k) X[(int64_t)j * nx + k] = Xp[(int64_t)i * nx + k];
    for (int k = 0; k < ny; ++k) Yr[(int64_t)j * ny + k] = Y[(int64_t)i * ny + k];
  }
}

// Statistics of the particle set (pf.py:418-420): x = mean of the particles, y = mean of their measurements, P = np.cov(X)
// (unbiased, N - 1), plus the per-state minimum / maximum the roughening needs (:409-411).  `add` (nullable): an increment
// applied to the particles first - the roughening step `X += dx` (:415).  One workgroup per filter.
__global__ __launch_bounds__(PF_TPB) void pf_stats_kernel(int n, int nx, int ny, double* __restrict__ X,
                                                          const double* __restrict__ Y, const double* __restrict__ add,
                                                          double* __restrict__ xm, double* __restrict__ ym,
                                                          double* __restrict__ P, double* __restrict__ xmin,
                                                          double* __restrict__ xmax) {
  __shared__ double red[PF_TPB];
  __shared__ double mean[16];
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  X += b * n * nx; Y += b * n * ny; xm += b * nx; ym += b * ny; P += b * nx * nx; xmin += b * nx; xmax += b * nx;
  if (add) {
    add += b * n * nx;
    for (int e = tid; e < n * nx; e += PF_TPB) X[e] += add[e];
    __syncthreads();
  }
  auto reduce = [&](double v, int op) {      // 0 sum, 1 min, 2 max
    red[tid] = v;
    __syncthreads();
    for (int o = PF_TPB / 2; o > 0; o >>= 1) {
      if (tid < o) red[tid] = op == 0 ? red[tid] + red[tid + o] : (op == 1 ? fmin(red[tid], red[tid + o]) : fmax(red[tid], red[tid + o]));
      __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
  };
  for (int i = 0; i < nx; ++i) {
    double s = 0.0, lo = INFINITY, hi = -INFINITY;
    for (int j = tid; j < n; j += PF_TPB) { const double x = X[(int64_t)j * nx + i]; s += x; lo = fmin(lo, x); hi = fmax(hi, x); }
    const double m = reduce(s, 0) / n, mn = reduce(lo, 1), mx = reduce(hi, 2);
    if (tid == 0) { xm[i] = m; mean[i] = m; xmin[i] = mn; xmax[i] = mx; }
  }
  for (int a = 0; a < ny; ++a) {
    double s = 0.0;
    for (int j = tid; j < n; j += PF_TPB) s += Y[(int64_t)j * ny + a];
    const double m = reduce(s, 0) / n;
    if (tid == 0) ym[a] = m;
  }
  __syncthreads();
  for (int i = 0; i < nx; ++i)
    for (int k = 0; k <= i; ++k) {
      double s = 0.0;
      for (int j = tid; j < n; j += PF_TPB) s += (X[(int64_t)j * nx + i] - mean[i]) * (X[(int64_t)j * nx + k] - mean[k]);
      const double c = reduce(s, 0) / (n - 1);
      if (tid == 0) { P[i * nx + k] = c; P[k * nx + i] = c; }
    }
}

}  // namespace hilo
