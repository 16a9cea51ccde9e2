// Moving-horizon estimation with PARAMETER estimation: the model parameters are decision variables next to the states
// (hilo_mpc/modules/estimator/mhe.py:614-623: v = [p | x_0..x_N | w_0..w_{N-1}], bounds p_lb / p_ub, guess p_guess) with
// the arrival term (p - p_arrival)^T Wp (p - p_arrival) (hilo_mpc/util/modeling.py:747-777, mhe.py:742-745); `estimate`
// returns (x_opt, p_opt) (mhe.py:377-380, :411-414).
// In engine terms the parameters ride along as constant extra states (p_{k+1} = p_k), free at stage 0 when estimated and
// pinned there when p_lb == p_ub (IPOPT removes fixed variables the same way); their box acts once (stage 0).  The linear
// copies stay exactly consistent under Newton steps, so the iterates are those of the single-p problem.
#include <string.h>

#include "hilo_mhe_est.h"

namespace hilo {

// pc.cost = [Wx (MX^2) | Wp (NP^2) | Wy | Ww | su];  par = [x_arrival | p_arrival];  sd_k = [u_meas_k | y_meas_k]
template <class M>
struct MheEst {
  static constexpr int MX = M::NX, NP = M::NP, NX = MX + NP, NU = MX, NY = M::NY, MU = M::NU, NPAR = MX + NP,
                       NSD = M::NU + M::NY;
  static constexpr bool FIX_X0 = true;   // with x0_free_mask: states free, estimated parameters free, the others pinned
  static constexpr bool BIG = false;
  static constexpr int NC = 0, NXV = NX, NX0 = NX, NU0 = NU;
  static constexpr bool COOP = false;
  static constexpr bool QUAD_COST = false;
  static constexpr int O_WX = 0, O_WP = O_WX + MX * MX, O_WY = O_WP + NP * NP, O_WW = O_WY + NY * NY, O_SU = O_WW + MX * MX,
                       O_END = O_SU + MU;
  static constexpr int NCOST = O_END;

  template <class T, class E>
  __device__ __forceinline__ static void dyn(const OcpConst& pc, const double*, const double* sd, int, const T* x, const T* w,
                                             T* xn, const E& ext) {
    T xp[MX], pp[NP > 0 ? NP : 1], xo[MX];
    double ue[MU > 0 ? MU : 1];
#pragma unroll
    for (int i = 0; i < MX; ++i) xp[i] = x[i] * pc.sz[i];
#pragma unroll
    for (int j = 0; j < NP; ++j) pp[j] = x[MX + j] * pc.sz[MX + j];
#pragma unroll
    for (int i = 0; i < MU; ++i) ue[i] = sd[i] * pc.cost[O_SU + i];
    model_step<M>(pc.order, pc.nsub, xp, ue, pp, pc.dt, xo, ext);
#pragma unroll
    for (int i = 0; i < MX; ++i) xn[i] = xo[i] * rcp_fast(pc.sz[i]) + w[i];   // mhe.py:739
#pragma unroll
    for (int j = 0; j < NP; ++j) xn[MX + j] = x[MX + j];
  }

  template <class T>
  __device__ __forceinline__ static T stage_cost(const OcpConst& pc, const double* par, const double* sd, int k, const T* x,
                                                 const T* w) {
    T xp[MX], pp[NP > 0 ? NP : 1];
#pragma unroll
    for (int i = 0; i < MX; ++i) xp[i] = x[i] * pc.sz[i];
#pragma unroll
    for (int j = 0; j < NP; ++j) pp[j] = x[MX + j] * pc.sz[MX + j];
    T acc = T(0.0);
    if (k == 0) {  // arrival cost on states and parameters (modeling.py:747-777; mhe.py:742-745)
      T d[NX];
#pragma unroll
      for (int i = 0; i < MX; ++i) d[i] = xp[i] - par[i];
#pragma unroll
      for (int j = 0; j < NP; ++j) d[MX + j] = pp[j] - par[MX + j];
#pragma unroll
      for (int i = 0; i < MX; ++i) {
        T s = T(0.0);
#pragma unroll
        for (int j = 0; j < MX; ++j) s = s + pc.cost[O_WX + i * MX + j] * d[j];
        acc = acc + d[i] * s;
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        T s = T(0.0);
#pragma unroll
        for (int j = 0; j < NP; ++j) s = s + pc.cost[O_WP + i * NP + j] * d[MX + j];
        acc = acc + d[MX + i] * s;
      }
      return acc;
    }
    double ue[MU > 0 ? MU : 1];
#pragma unroll
    for (int i = 0; i < MU; ++i) ue[i] = sd[i] * pc.cost[O_SU + i];
    T yv[NY], r[NY];
    M::meas(xp, ue, pp, pc.dt, yv);
#pragma unroll
    for (int a = 0; a < NY; ++a) r[a] = yv[a] - sd[MU + a];
#pragma unroll
    for (int a = 0; a < NY; ++a) {
      T s = T(0.0);
#pragma unroll
      for (int b = 0; b < NY; ++b) s = s + pc.cost[O_WY + a * NY + b] * r[b];
      acc = acc + r[a] * s;
    }
    T ws[MX];
#pragma unroll
    for (int i = 0; i < MX; ++i) ws[i] = w[i] * pc.sz[NX + i];
#pragma unroll
    for (int i = 0; i < MX; ++i) {
      T s = T(0.0);
#pragma unroll
      for (int j = 0; j < MX; ++j) s = s + pc.cost[O_WW + i * MX + j] * ws[j];
      acc = acc + ws[i] * s;
    }
    return acc;
  }

  template <class T>
  __device__ __forceinline__ static T term_cost(const OcpConst&, const double*, const double*, const T*) { return T(0.0); }
};

// reference row [p | x (N+1)*mx | w N*mx] -> engine row [xa (N+1)*(mx+np) | w]; p replicated over the stages
__global__ void mhe_est_to_engine(int64_t batch, int N, int mx, int np, const double* __restrict__ v, int64_t v_stride,
                                  double* __restrict__ ve, int has_w) {
  const int nxa = mx + np, ne = (N + 1) * nxa + (has_w ? N * mx : 0);
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= batch * ne) return;
  const int64_t b = e / ne;
  const int q = (int)(e - b * ne);
  const double* row = v + b * v_stride;
  double val;
  if (q < (N + 1) * nxa) {
    const int k = q / nxa, i = q - k * nxa;
    val = i < mx ? row[np + k * mx + i] : row[i - mx];
  } else {
    val = row[np + (N + 1) * mx + (q - (N + 1) * nxa)];
  }
  ve[e] = val;
}

// engine solution -> reference row, multipliers of the state defects, x_opt = x_N un-scaled
__global__ void mhe_est_from_engine(const OcpConst* __restrict__ pc, int64_t batch, int N, int mx, int np,
                                    const double* __restrict__ ve, const double* __restrict__ lame, double* __restrict__ v,
                                    double* __restrict__ lam_g, double* __restrict__ x_opt) {
  const int nxa = mx + np, nv = np + (N + 1) * mx + N * mx, ne = (N + 1) * nxa + N * mx;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= batch * nv) return;
  const int64_t b = e / nv;
  const int q = (int)(e - b * nv);
  const double* row = ve + b * ne;
  double val;
  if (q < np) val = row[mx + q];                                   // stage-0 copy
  else if (q < np + (N + 1) * mx) {
    const int r = q - np, k = r / mx, i = r - k * mx;
    val = row[k * nxa + i];
    if (k == N && x_opt) x_opt[b * mx + i] = val * pc->sz[i];       // mhe.py:381-384
  } else val = row[(N + 1) * nxa + (q - np - (N + 1) * mx)];
  v[e] = val;
  if (lam_g && q < N * mx) {
    const int k = q / mx, i = q - k * mx;
    lam_g[b * (int64_t)(N * mx) + q] = lame[b * (int64_t)(N * nxa) + k * nxa + i];
  }
}

template <class M>
static int est_launch(const MheEstArgs& a) {
  using PB = MheEst<M>;
  if (a.lds_bytes > 64 * 1024)
    HILO_HIP_CHECK(hipFuncSetAttribute((const void*)ocp_solve_kernel<PB, OCP_TPB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)a.lds_bytes));
  hipLaunchKernelGGL((ocp_solve_kernel<PB, OCP_TPB>), dim3((unsigned)a.batch), dim3(OCP_TPB), a.lds_bytes, a.stream, a.dev,
                     a.batch, a.x0e, a.par, (int64_t)PB::NPAR, a.sd, a.sd_stride, a.v0e, a.v0e_stride, 0, 0, a.ve, a.f_opt,
                     a.lame, (double*)nullptr, 0, a.status, a.iters, a.kkt, (long long*)nullptr);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}
template <class M>
static size_t est_lds(int N) { return Ocp<MheEst<M>>::lds_doubles(N) * sizeof(double); }
template <class M>
static void est_offsets(int* o) {
  using PB = MheEst<M>;
  o[0] = PB::O_WX; o[1] = PB::O_WP; o[2] = PB::O_WY; o[3] = PB::O_WW; o[4] = PB::O_SU;
}

const MheEstVariant* mhe_est_find(int model_id) {
  static const MheEstVariant v[] = {
      {HILO_MODEL_CHEMOSTAT4, &est_lds<Chemostat4>, &est_launch<Chemostat4>, &est_offsets<Chemostat4>},
      {HILO_MODEL_BIOREACTOR3, &est_lds<Bioreactor3>, &est_launch<Bioreactor3>, &est_offsets<Bioreactor3>},
  };
  for (const auto& c : v)
    if (c.model_id == model_id) return &c;
  return nullptr;
}

__global__ void mhe_est_pack_kernel(int64_t batch, int mx, int np, const double* __restrict__ p, int64_t p_stride,
                                    const double* __restrict__ xa, double* __restrict__ x0e, double* __restrict__ par) {
  const int nxa = mx + np;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= batch * nxa) return;
  const int64_t b = e / nxa;
  const int i = (int)(e - b * nxa);
  const double pv = i >= mx ? p[b * p_stride + (i - mx)] : 0.0;
  x0e[e] = pv;
  par[e] = i < mx ? xa[b * mx + i] : pv;
}
int mhe_est_pack(int64_t batch, int mx, int np, const double* p, int64_t p_stride, const double* xa, double* x0e, double* par,
                 hipStream_t s) {
  const int64_t tot = batch * (mx + np);
  hipLaunchKernelGGL(mhe_est_pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, batch, mx, np, p, p_stride, xa,
                     x0e, par);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

int mhe_est_convert_in(int64_t batch, int N, int mx, int np, const double* v, int64_t v_stride, double* ve, hipStream_t s, int has_w) {
  const int64_t tot = batch * ((int64_t)(N + 1) * (mx + np) + (has_w ? (int64_t)N * mx : 0));
  hipLaunchKernelGGL(mhe_est_to_engine, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, batch, N, mx, np, v, v_stride, ve, has_w);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}
int mhe_est_convert_out(const OcpConst* pc, int64_t batch, int N, int mx, int np, const double* ve, const double* lame,
                        double* v, double* lam_g, double* x_opt, hipStream_t s) {
  const int64_t tot = batch * ((int64_t)np + (int64_t)(2 * N + 1) * mx);
  hipLaunchKernelGGL(mhe_est_from_engine, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, pc, batch, N, mx, np, ve, lame,
                     v, lam_g, x_opt);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

}  // namespace hilo
