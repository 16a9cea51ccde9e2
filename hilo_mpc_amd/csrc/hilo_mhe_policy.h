// Policy of the interior-point engine for the moving-horizon estimator with state noise (hilo_mpc/modules/estimator/mhe.py:596-790),
// in a header so that the library's zoo instantiations (hilo_mhe.hip) and the run-time compiled ones for models written as
// expressions (hilo_jit.hip, policy JIT_MHE) are the same code.
//   v = [p | x_0..x_N | w_0..w_{N-1} (| ip_0..ip_{N-1})]                        mhe.py:614-671
//   g_k = x_{k+1} - (Phi(x_k, u_meas_k, p) + w_k) = 0                            mhe.py:726-740 (scaled noise on the scaled state)
//   J   = arrival(x_0) at k = 0; (h(x_k)-y_k)^T Wy (.) + w_k^T Ww w_k, k >= 1    mhe.py:742-748 (no stage cost at k = 0)
//   costs act on un-scaled quantities (hilo_mpc/util/modeling.py:665-672)
// Phi: explicit Runge-Kutta / the discrete map (CD = 0; 'discrete', mhe.py:563-571) or direct collocation of degree CD
// (the reference's default, mhe.py:512-561; the collocation states are eliminated inside the map - hilo_colloc.h - and rebuilt
// on output, mhe_coll_output).  In engine terms: controls := the noise w_k (B_k = I), x_0 free, per-stage data = (u_meas_k, y_meas_k).
#pragma once
#include "hilo_ocp.h"

namespace hilo {

// pc.cost = [Wx | Wy | Ww | su];  par = [model parameters | x_arrival];  sd_k = [u_meas_k | y_meas_k]
// SYM_: model / measurement derivatives from generated symbolic code when the model has it (the host selects SYM_ = false for
// sub-stepped integration); collocation keeps the Taylor path (the map is implicit)
template <class M, bool SYM_ = true, int CD = 0>
struct MheNoise {
  using Model = M;
  static constexpr bool SYM_MHE = CD == 0 && SYM_ && ModelSym<M>::value && ModelSym<M>::HAS_MEAS && !model_has_ext<M>::value &&
                                  !M::DISCRETE && M::NX % 2 == 0;
  static constexpr int NX = M::NX, NU = M::NX, NY = M::NY, MU = M::NU, NPAR = M::NP + M::NX, NSD = M::NU + M::NY;
  static constexpr bool FIX_X0 = false;
  static constexpr bool BIG = false;  // iterate in LDS
  static constexpr int NC = 0, NXV = NX, NX0 = NX, NU0 = NU;  // no inequality rows; plain [x | u] decision vector
  static constexpr bool COOP = CD == 0 && model_has_ext<M>::value;
  static constexpr bool QUAD_COST = false;  // the measurement function may be nonlinear: Taylor evaluation
  static constexpr int O_WX = 0, O_WY = O_WX + NX * NX, O_WW = O_WY + NY * NY, O_SU = O_WW + NX * NX, O_END = O_SU + MU;
  static constexpr int NCOST = O_END;
  static_assert(CD == 0 || !M::DISCRETE, "collocation needs the continuous model");

  template <class T, class E>
  __device__ __forceinline__ static void dyn(const OcpConst& pc, const double* par, const double* sd, int, const T* x,
                                             const T* w, T* xn, const E& ext) {
    T xp[NX], xo[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xp[i] = x[i] * pc.sz[i];
    if constexpr (CD > 0) {
      T ue[MU > 0 ? MU : 1];
#pragma unroll
      for (int i = 0; i < MU; ++i) ue[i] = T(sd[i] * pc.cost[O_SU + i]);
      Colloc<M, CD>::step(pc.coll, xp, ue, par, pc.dt, xo);
    } else {
      double ue[MU > 0 ? MU : 1];
#pragma unroll
      for (int i = 0; i < MU; ++i) ue[i] = sd[i] * pc.cost[O_SU + i];
      model_step<M>(pc.order, pc.nsub, xp, ue, par, pc.dt, xo, ext);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = xo[i] * (1.0 / pc.sz[i]) + w[i];  // mhe.py:739: scaled noise, scaled state
  }

  template <class T>
  __device__ __forceinline__ static T stage_cost(const OcpConst& pc, const double* par, const double* sd, int k,
                                                 const T* x, const T* w) {
    T xp[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xp[i] = x[i] * pc.sz[i];
    T acc = T(0.0);
    if (k == 0) {  // arrival cost (modeling.py:747-777); mhe.py:742-745
      T d[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) d[i] = xp[i] - par[M::NP + i];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        T s = T(0.0);
#pragma unroll
        for (int j = 0; j < NX; ++j) s = s + pc.cost[O_WX + i * NX + j] * d[j];
        acc = acc + d[i] * s;
      }
      return acc;
    }
    double ue[MU > 0 ? MU : 1];
#pragma unroll
    for (int i = 0; i < MU; ++i) ue[i] = sd[i] * pc.cost[O_SU + i];
    T yv[NY], r[NY];
    M::meas(xp, ue, par, pc.dt, yv);
#pragma unroll
    for (int a = 0; a < NY; ++a) r[a] = yv[a] - sd[MU + a];
#pragma unroll
    for (int a = 0; a < NY; ++a) {
      T s = T(0.0);
#pragma unroll
      for (int b = 0; b < NY; ++b) s = s + pc.cost[O_WY + a * NY + b] * r[b];
      acc = acc + r[a] * s;
    }
    T ws[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) ws[i] = w[i] * pc.sz[NX + i];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      T s = T(0.0);
#pragma unroll
      for (int j = 0; j < NX; ++j) s = s + pc.cost[O_WW + i * NX + j] * ws[j];
      acc = acc + ws[i] * s;
    }
    return acc;
  }

  template <class T>
  __device__ __forceinline__ static T term_cost(const OcpConst&, const double*, const double*, const T*) { return T(0.0); }
};

// Output pass of the collocation transcription, one thread per (instance, interval): the collocation states (scaled like the
// states) behind the noise block of v, and lam_g in the reference's row order - per stage [collocation rows (D nx) | continuity
// (nx)] (mhe.py:728, :740) - with the multipliers of the collocation rows from the continuity multiplier (hilo_colloc.h).
//   vc   [B][np + (N+1) nx + N nx]   the engine's result (reference layout without the collocation block)
//   lamc [B][N nx]                   multipliers of the continuity rows
//   par  [B][par_stride]             [model parameters | x_arrival];  sdata [B][sd_stride]: per stage [u_meas | y_meas]
template <class M, int D>
__device__ __forceinline__ void mhe_coll_output(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ vc,
                                                const double* __restrict__ lamc, const double* __restrict__ par, int64_t par_stride,
                                                const double* __restrict__ sdata, int64_t sd_stride, double* __restrict__ v,
                                                double* __restrict__ lam_g) {
  using PB = MheNoise<M, false, D>;
  constexpr int NX = M::NX, MU = M::NU, NP = M::NP, DN = D * NX, NSD = PB::NSD;
  const int N = pcg->N;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= batch * N) return;
  const int64_t b = e / N;
  const int k = (int)(e - b * N);
  const int nvc = NP + (N + 1) * NX + N * NX, nv = nvc + N * DN;
  const double* row = vc + b * nvc;
  double x[NX], u[MU > 0 ? MU : 1], p[NP > 0 ? NP : 1], X[DN], mat[DN * DN], lam[NX], mu[DN];
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = row[NP + k * NX + i] * pcg->sz[i];
#pragma unroll
  for (int i = 0; i < MU; ++i) u[i] = sdata[b * sd_stride + k * NSD + i] * pcg->cost[PB::O_SU + i];
#pragma unroll
  for (int i = 0; i < NP; ++i) p[i] = par[b * par_stride + i];
  Colloc<M, D>::solve(pcg->coll, x, u, p, pcg->dt, X, mat);
  double* out = v + b * nv;
  if (k == 0)
    for (int i = 0; i < nvc; ++i) out[i] = row[i];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int m = 0; m < NX; ++m) out[nvc + k * DN + i * NX + m] = X[i * NX + m] / pcg->sz[m];
  if (lam_g) {
    // rows of the scaled model: G_s = G / s_m  =>  mu_s = mu * s_m; the continuity multiplier is the engine's lambda (hilo_nmpc_coll.hip)
#pragma unroll
    for (int m = 0; m < NX; ++m) lam[m] = lamc[b * (int64_t)(N * NX) + k * NX + m] / pcg->sz[m];
    Colloc<M, D>::multipliers(pcg->coll, X, u, p, pcg->dt, lam, mu);
    double* lg = lam_g + b * (int64_t)(N * (DN + NX)) + k * (DN + NX);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int m = 0; m < NX; ++m) lg[i * NX + m] = mu[i * NX + m] * pcg->sz[m];
#pragma unroll
    for (int m = 0; m < NX; ++m) lg[DN + m] = lamc[b * (int64_t)(N * NX) + k * NX + m];
  }
}

}  // namespace hilo
