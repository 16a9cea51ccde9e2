// Tracking NMPC with per-stage data: trajectory-tracking references (`add_states(..., trajectory_tracking=True)` +
// `optimize(ref_sc=..., ref_tc=...)`, hilo_mpc/modules/controller/mpc.py:365-463, hilo_mpc/util/modeling.py:262-283) and
// time-varying parameters (`set_time_varying_parameters`, `optimize(tvp=...)`, mpc.py:335-364, optimizer.py:905-929).
// Per instance and stage k the caller hands [zref_k (nz, already divided by the scaling) | p_k (np)], k = 0..N; row N
// carries the terminal reference in its first nx entries.
#include "hilo_nmpc_gen.h"
#include "hilo_nmpc_track.h"

namespace hilo {

template <class M>
struct NmpcTv : NmpcTrack<M, false, false> {   // its own shooting map: Taylor derivatives
  using Base = NmpcTrack<M, false, false>;
  static constexpr int NX = Base::NX, NU = Base::NU, NZ = Base::NZ, NP = M::NP;
  static constexpr int NSD = NZ + NP;
  using Base::O_HASDU; using Base::O_WDU; using Base::O_WN; using Base::O_WZ;

  template <class T, class E>
  __device__ __forceinline__ static void dyn(const OcpConst& pc, const double*, const double* sd, int, const T* x,
                                             const T* u, T* xn, const E& ext) {
    T xp[NX], up[NU > 0 ? NU : 1], xo[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xp[i] = x[i] * pc.sz[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) up[i] = u[i] * pc.sz[NX + i];
    model_step<M>(pc.order, pc.nsub, xp, up, sd + NZ, pc.dt, xo, ext);   // p_k (mpc.py:1641 `_rearrange_parameters`)
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = xo[i] * rcp_fast(pc.sz[i]);
  }

  template <class T>
  __device__ __forceinline__ static T stage_cost(const OcpConst& pc, const double* par, const double* sd, int k,
                                                 const T* x, const T* u) {
    T z[NZ];
#pragma unroll
    for (int i = 0; i < NX; ++i) z[i] = x[i] - sd[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) z[NX + i] = u[i] - sd[NX + i];
    T acc = T(0.0);
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      T s = T(0.0);
#pragma unroll
      for (int j = 0; j < NZ; ++j) s = s + pc.cost[O_WZ + i * NZ + j] * z[j];
      acc = acc + z[i] * s;
    }
    if (k == 0 && pc.cost[O_HASDU] != 0.0) {
      T d[NU > 0 ? NU : 1];
#pragma unroll
      for (int i = 0; i < NU; ++i) d[i] = u[i] - par[M::NP + i];
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        T s = T(0.0);
#pragma unroll
        for (int j = 0; j < NU; ++j) s = s + pc.cost[O_WDU + i * NU + j] * d[j];
        acc = acc + d[i] * s;
      }
    }
    return acc;
  }

  __device__ __forceinline__ static double cost_grad(const OcpConst& pc, const double* par, const double* sd, int k, int i,
                                                     const double* z) {
    double g = 0.0;
#pragma unroll
    for (int j = 0; j < NZ; ++j) g += (pc.cost[O_WZ + i * NZ + j] + pc.cost[O_WZ + j * NZ + i]) * (z[j] - sd[j]);
    if (k == 0 && i >= NX && pc.cost[O_HASDU] != 0.0) {
#pragma unroll
      for (int j = 0; j < NU; ++j)
        g += (pc.cost[O_WDU + (i - NX) * NU + j] + pc.cost[O_WDU + j * NU + (i - NX)]) * (z[NX + j] - par[M::NP + j]);
    }
    return g;
  }

  template <class T>
  __device__ __forceinline__ static T term_cost(const OcpConst& pc, const double*, const double* sd, const T* x) {
    T z[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) z[i] = x[i] - sd[i];
    T acc = T(0.0);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      T s = T(0.0);
#pragma unroll
      for (int j = 0; j < NX; ++j) s = s + pc.cost[O_WN + i * NX + j] * z[j];
      acc = acc + z[i] * s;
    }
    return acc;
  }
};

template <class M>
static int tv_launch(const GenLaunchArgs& a, const double* sd, int64_t sd_stride) {
  using PB = NmpcTv<M>;
  if (a.lds_bytes > 64 * 1024)
    HILO_HIP_CHECK(hipFuncSetAttribute((const void*)ocp_solve_kernel<PB, OCP_TPB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)a.lds_bytes));
  hipLaunchKernelGGL((ocp_solve_kernel<PB, OCP_TPB>), dim3((unsigned)a.batch), dim3(OCP_TPB), a.lds_bytes, a.stream, a.dev,
                     a.batch, a.x0, a.par, a.par_stride, sd, sd_stride, a.v0, a.v0_stride, 0, 0, a.v_opt, a.f_opt, a.lam_g, a.u0,
                     0, a.status, a.iters, a.kkt, a.prof, a.ws);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}
template <class M>
static size_t tv_lds(int N) { return Ocp<NmpcTv<M>>::lds_doubles(N) * sizeof(double); }

const TvVariant* nmpc_tv_find(int model_id) {
  static const TvVariant v[] = {
      {HILO_MODEL_CHEMOSTAT4, &tv_lds<Chemostat4>, &tv_launch<Chemostat4>},
      {HILO_MODEL_BIOREACTOR3, &tv_lds<Bioreactor3>, &tv_launch<Bioreactor3>},
      {HILO_MODEL_PENDULUM4, &tv_lds<Pendulum4>, &tv_launch<Pendulum4>},
  };
  for (const auto& c : v)
    if (c.model_id == model_id) return &c;
  return nullptr;
}

}  // namespace hilo
