// Batched Kalman / extended / unscented Kalman filter kernels (one filter instance per lane) + C ABI.
//
// Reference semantics: hilo_mpc/modules/estimator/kf.py
//   predict  :71-133   (UKF :505-554)      update :135-186 (UKF :556-604)      step = update(predict) :258-265
// HBM-bound: per step an instance moves its packed [x|P] tile in and out plus y, [u;p]
// (bytes_kf = 8*(2 nx (nx+1) + 2 ny + nu + np)).  Tiles travel HBM<->LDS with unit-stride lanes and are then
// picked up row-per-lane (hilo_common.h), all arithmetic is register-resident fp64 with compile-time shapes.
#include "hilo_common.h"
#include "hilo_models.h"

namespace hilo {

constexpr int KF_TPB = 64;   // one wave per workgroup

struct KfParams {
  int kind, continuous, erk_order, n_sub;
  double dt, gamma, wm0, wc0, wi;  // UKF: W_m[0], W_c[0], W[1:] (kf.py:493-500)
};

template <int N> struct MaxOne { static constexpr int v = N > 0 ? N : 1; };

// ---- small dense helpers (row-major, compile-time sizes) ------------------------------------------------
template <int N>
__device__ __forceinline__ void chol_lower(const double* A, double* L) {
  // A = L L^T
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double s = A[j * N + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * N + k] * L[j * N + k];
    const double d = ::sqrt(s);
    L[j * N + j] = d;
    const double id = 1.0 / d;
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
  double L[NY * NY];
  chol_lower<NY>(Pyy, L);
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
      z[a] = s / L[a * NY + a];
    }
#pragma unroll
    for (int a = NY - 1; a >= 0; --a) {
      double s = z[a];
#pragma unroll
      for (int b = a + 1; b < NY; ++b) s -= L[b * NY + a] * K[i * NY + b];
      K[i * NY + a] = s / L[a * NY + a];
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

template <class M>
__device__ __forceinline__ void ekf_predict(const KfParams& kp, double* x, double* P, const double* u,
                                            const double* p, const double* Q) {
  constexpr int NX = M::NX;
  if constexpr (!M::DISCRETE) if (kp.continuous) {
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
  model_step<M>(kp.erk_order, kp.n_sub, xd, u, p, kp.dt, xn);
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
// The weighted sums follow the reference's accumulation order without FMA contraction: with alpha = 1e-3 the
// centre weight is ~ -1e6 and six digits cancel, so rounding order is visible in the result.
template <class M>
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
    if (kp.continuous && !M::DISCRETE)
      model_step<M>(4, kp.n_sub, xs, u, p, kp.dt, xo);  // the reference integrates with CVODES
    else
      model_step<M>(kp.erk_order, kp.n_sub, xs, u, p, kp.dt, xo);
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
__global__ __launch_bounds__(KF_TPB) KF_OCC void kf_kernel(KfParams kp, int64_t batch, const double* __restrict__ in_tile,
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
int launch(const KfParams& kp, int64_t batch, const double* in, const double* y, const double* up,
           int64_t up_stride, const double* Q, int64_t q_stride, const double* R, int64_t r_stride, double* out,
           double* y_pred, hipStream_t s) {
  if (batch == 0) return HILO_OK;
  int64_t ipw = (batch + 1023) / 1024;   // 256 compute units x 4 SIMDs
  ipw = ipw < 16 ? 16 : (ipw > KF_TPB ? KF_TPB : ipw);
  const unsigned grid = (unsigned)((batch + ipw - 1) / ipw);
  hipLaunchKernelGGL((kf_kernel<M, UKF, MODE>), dim3(grid), dim3(KF_TPB), 0, s, kp, batch, in, y, up, up_stride, Q,
                     q_stride, R, r_stride, out, y_pred, (int)ipw);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

template <class M>
int dispatch_kind(int mode, const KfParams& kp, int64_t batch, const double* in, const double* y,
                  const double* up, int64_t us, const double* Q, int64_t qs, const double* R, int64_t rs,
                  double* out, double* yp, hipStream_t s) {
  const bool ukf = kp.kind == HILO_KF_UKF;
  if (ukf) {
    if (mode == 0) return launch<M, true, 0>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
    if (mode == 1) return launch<M, true, 1>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
    return launch<M, true, 2>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
  }
  if (mode == 0) return launch<M, false, 0>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
  if (mode == 1) return launch<M, false, 1>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
  return launch<M, false, 2>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
}

}  // namespace hilo

using namespace hilo;

struct hilo_kf {
  hilo_kf_desc desc;
  int device;
  int nx, nu, np, ny;
  KfParams kp;
};

#define HILO_KF_MODELS(X)                 \
  X(HILO_MODEL_TOY1D, Toy1D)              \
  X(HILO_MODEL_BIOREACTOR3, Bioreactor3)  \
  X(HILO_MODEL_CHEMOSTAT4, Chemostat4)    \
  X(HILO_MODEL_PENDULUM4, Pendulum4)      \
  X(HILO_MODEL_LINEAR2, Linear2)

static int kf_model_dims(const hilo_kf_desc* d, int* nx, int* nu, int* np, int* ny, int* discrete) {
  switch (d->model_id) {
#define X(ID, T) case ID: *nx = T::NX; *nu = T::NU; *np = T::NP; *ny = T::NY; *discrete = T::DISCRETE; return HILO_OK;
    HILO_KF_MODELS(X)
#undef X
    case HILO_MODEL_LTI:
      if (d->lti_nx == 2 && d->lti_nu == 1 && d->lti_ny == 1) { *nx = 2; *nu = 1; *np = 4 + 2 + 2; *ny = 1; *discrete = 1; return HILO_OK; }
      if (d->lti_nx == 2 && d->lti_nu == 1 && d->lti_ny == 2) { *nx = 2; *nu = 1; *np = 4 + 2 + 4; *ny = 2; *discrete = 1; return HILO_OK; }
      if (d->lti_nx == 4 && d->lti_nu == 2 && d->lti_ny == 2) { *nx = 4; *nu = 2; *np = 16 + 8 + 8; *ny = 2; *discrete = 1; return HILO_OK; }
      return fail(HILO_ENOTSUP, "LTI filter built for (nx,nu,ny) in {(2,1,1),(2,1,2),(4,2,2)}, got (%d,%d,%d)",
                  d->lti_nx, d->lti_nu, d->lti_ny);
    default:
      return fail(HILO_EINVAL, "unknown model id %d", d->model_id);
  }
}

extern "C" int hilo_model_dims(int model_id, int* nx, int* nu, int* np, int* ny, int* discrete) {
  hilo_kf_desc d = {};
  d.model_id = model_id;
  if (model_id == HILO_MODEL_LTI) return fail(HILO_EINVAL, "LTI dimensions are caller-defined");
  if (model_id == HILO_MODEL_CHEMOSTAT4_GP) model_id = d.model_id = HILO_MODEL_CHEMOSTAT4;  // same signature
  if (model_id == HILO_MODEL_CSTR3) {  // controller-only model (no filter instantiation)
    if (nx) *nx = Cstr3::NX;
    if (nu) *nu = Cstr3::NU;
    if (np) *np = Cstr3::NP;
    if (ny) *ny = Cstr3::NY;
    if (discrete) *discrete = Cstr3::DISCRETE;
    return HILO_OK;
  }
  if (model_id == HILO_MODEL_ROBOT6) {  // controller-only model (no filter instantiation)
    if (nx) *nx = Robot6::NX;
    if (nu) *nu = Robot6::NU;
    if (np) *np = Robot6::NP;
    if (ny) *ny = Robot6::NY;
    if (discrete) *discrete = Robot6::DISCRETE;
    return HILO_OK;
  }
  int a, b, c, e, f;
  int rc = kf_model_dims(&d, &a, &b, &c, &e, &f);
  if (rc) return rc;
  if (nx) *nx = a;
  if (nu) *nu = b;
  if (np) *np = c;
  if (ny) *ny = e;
  if (discrete) *discrete = f;
  return HILO_OK;
}

extern "C" int hilo_kf_create(const hilo_kf_desc* desc, int device, hilo_kf** out) {
  HILO_REQUIRE(desc && out, "hilo_kf_create: NULL argument");
  HILO_REQUIRE(desc->kind >= HILO_KF_KF && desc->kind <= HILO_KF_UKF, "hilo_kf_create: unknown kind %d", desc->kind);
  HILO_REQUIRE(desc->dt > 0.0, "hilo_kf_create: dt must be positive");
  int nx, nu, np, ny, disc;
  int rc = kf_model_dims(desc, &nx, &nu, &np, &ny, &disc);
  if (rc) return rc;
  HILO_REQUIRE(disc || (desc->erk_order >= 1 && desc->erk_order <= 4) || desc->continuous,
               "hilo_kf_create: erk_order must be 1..4 (modeling.py:1239-1250), got %d", desc->erk_order);
  if (desc->kind == HILO_KF_UKF) {
    // kf.py:475-484
    HILO_REQUIRE(desc->alpha > 0.0 && desc->alpha <= 1.0,
                 "The parameter alpha needs to lie in the interval (0, 1]. Supplied alpha is %g.", desc->alpha);
    HILO_REQUIRE(desc->kappa >= 0.0,
                 "The parameter kappa needs to be greater or equal to 0. Supplied kappa is %g.", desc->kappa);
  }
  int ndev = 0;
  HILO_HIP_CHECK(hipGetDeviceCount(&ndev));
  HILO_REQUIRE(device >= 0 && device < ndev, "hilo_kf_create: device %d out of range (%d visible)", device, ndev);
  hilo_kf* kf = new hilo_kf();
  kf->desc = *desc;
  kf->device = device;
  kf->nx = nx; kf->nu = nu; kf->np = np; kf->ny = ny;
  KfParams& kp = kf->kp;
  kp.kind = desc->kind;
  kp.continuous = desc->continuous;
  kp.erk_order = desc->erk_order >= 1 ? desc->erk_order : 4;
  kp.n_sub = desc->n_sub >= 1 ? desc->n_sub : 1;
  kp.dt = desc->dt;
  // kf.py:493-500
  const double lam = desc->alpha * desc->alpha * (nx + desc->kappa) - nx;
  kp.gamma = ::sqrt(nx + lam);
  kp.wm0 = lam / (nx + lam);
  kp.wc0 = lam / (nx + lam) + 1 - desc->alpha * desc->alpha + desc->beta;
  kp.wi = 1 / (2 * (nx + lam));
  *out = kf;
  return HILO_OK;
}

extern "C" void hilo_kf_destroy(hilo_kf* kf) { delete kf; }

extern "C" int hilo_kf_dims(const hilo_kf* kf, int* nx, int* nu, int* np, int* ny, int* pred_width) {
  HILO_REQUIRE(kf, "hilo_kf_dims: NULL handle");
  if (nx) *nx = kf->nx;
  if (nu) *nu = kf->nu;
  if (np) *np = kf->np;
  if (ny) *ny = kf->ny;
  if (pred_width) *pred_width = kf->kp.kind == HILO_KF_UKF ? 1 + kf->nx + 2 * kf->nx + 1 : kf->nx + 1;
  return HILO_OK;
}

static int kf_run(hilo_kf* kf, int mode, int64_t batch, const double* in, const double* y, const double* up,
                  int64_t us, const double* Q, int64_t qs, const double* R, int64_t rs, double* out, double* yp,
                  void* stream) {
  HILO_REQUIRE(kf, "hilo_kf: NULL handle");
  HILO_REQUIRE(batch >= 0, "hilo_kf: negative batch");
  if (batch == 0) return HILO_OK;
  HILO_REQUIRE(in && out, "hilo_kf: NULL tile pointer");
  HILO_REQUIRE(kf->nu + kf->np == 0 || up, "hilo_kf: the model has %d inputs/parameters but `up` is NULL", kf->nu + kf->np);
  HILO_REQUIRE(us == 0 || us >= kf->nu + kf->np, "hilo_kf: up_stride %lld < nu+np", (long long)us);
  if (mode != 1) HILO_REQUIRE(Q && (qs == 0 || qs >= kf->nx * kf->nx), "hilo_kf: bad Q / q_stride");
  if (mode != 0) HILO_REQUIRE(R && y && yp && (rs == 0 || rs >= kf->ny * kf->ny), "hilo_kf: bad R / y / y_pred");
  HILO_HIP_CHECK(hipSetDevice(kf->device));
  hipStream_t s = (hipStream_t)stream;
  const KfParams& kp = kf->kp;
  switch (kf->desc.model_id) {
#define X(ID, T) case ID: return dispatch_kind<T>(mode, kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
    HILO_KF_MODELS(X)
#undef X
    case HILO_MODEL_LTI:
      if (kf->nx == 2 && kf->ny == 1) return dispatch_kind<Lti<2, 1, 1>>(mode, kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
      if (kf->nx == 2 && kf->ny == 2) return dispatch_kind<Lti<2, 1, 2>>(mode, kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
      return dispatch_kind<Lti<4, 2, 2>>(mode, kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
  }
  return fail(HILO_EINVAL, "unknown model id %d", kf->desc.model_id);
}

extern "C" int hilo_kf_predict(hilo_kf* kf, int64_t batch, const double* xP, const double* up, int64_t up_stride,
                               const double* Q, int64_t q_stride, double* pred, void* stream) {
  return kf_run(kf, 0, batch, xP, nullptr, up, up_stride, Q, q_stride, nullptr, 0, pred, nullptr, stream);
}
extern "C" int hilo_kf_update(hilo_kf* kf, int64_t batch, const double* pred, const double* y, const double* up,
                              int64_t up_stride, const double* R, int64_t r_stride, double* xP_out, double* y_pred,
                              void* stream) {
  return kf_run(kf, 1, batch, pred, y, up, up_stride, nullptr, 0, R, r_stride, xP_out, y_pred, stream);
}
extern "C" int hilo_kf_step(hilo_kf* kf, int64_t batch, const double* xP, const double* y, const double* up,
                            int64_t up_stride, const double* Q, int64_t q_stride, const double* R, int64_t r_stride,
                            double* xP_out, double* y_pred, void* stream) {
  return kf_run(kf, 2, batch, xP, y, up, up_stride, Q, q_stride, R, r_stride, xP_out, y_pred, stream);
}
