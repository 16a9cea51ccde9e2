// Tracking NMPC with the reference's default transcription, direct collocation (hilo_colloc.h): policy = NmpcTrack with
// the implicit shooting map; an output pass reconstructs the collocation states and the multipliers of their equations so
// that `v` and `lam_g` have the reference's layout (mpc.py:1497-1518: v = [x | u | ip], :1657-1669: g per stage =
// [collocation rows | continuity]).
#include "hilo_nmpc_gen.h"
#include "hilo_nmpc_track.h"

namespace hilo {

template <class M, int D>
struct NmpcColl : NmpcTrack<M, false, false> {   // its own shooting map: Taylor derivatives
  using Base = NmpcTrack<M, false, false>;
  static constexpr int NX = Base::NX, NU = Base::NU;
  template <class T, class E>
  __device__ __forceinline__ static void dyn(const OcpConst& pc, const double* par, const double*, int, const T* x,
                                             const T* u, T* xn, const E&) {
    T xp[NX], up[NU > 0 ? NU : 1], xo[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xp[i] = x[i] * pc.sz[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) up[i] = u[i] * pc.sz[NX + i];
    Colloc<M, D>::step(pc.coll, xp, up, par, pc.dt, xo);
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = xo[i] * rcp_fast(pc.sz[i]);
  }
};

// one thread per (instance, interval): collocation states (scaled like the states) and the multipliers of their rows
template <class M, int D>
__global__ void coll_output_kernel(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ vc,
                                   const double* __restrict__ lamc, const double* __restrict__ par, int64_t par_stride,
                                   double* __restrict__ v, double* __restrict__ lam_g) {
  constexpr int NX = M::NX, NU = M::NU, DN = D * NX;
  const int N = pcg->N;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= batch * N) return;
  const int64_t b = e / N;
  const int k = (int)(e - b * N);
  const int nvc = (N + 1) * NX + N * NU, nv = nvc + N * DN;
  const double* row = vc + b * nvc;
  double x[NX], u[NU > 0 ? NU : 1], p[M::NP > 0 ? M::NP : 1], X[DN], mat[DN * DN], lam[NX], mu[DN];
#pragma unroll
  for (int i = 0; i < NX; ++i) x[i] = row[k * NX + i] * pcg->sz[i];
#pragma unroll
  for (int i = 0; i < NU; ++i) u[i] = row[(N + 1) * NX + k * NU + i] * pcg->sz[NX + i];
#pragma unroll
  for (int i = 0; i < M::NP; ++i) p[i] = par[b * par_stride + i];
  Colloc<M, D>::solve(pcg->coll, x, u, p, pcg->dt, X, mat);
  double* out = v + b * nv;
  if (k == 0)
    for (int i = 0; i < nvc; ++i) out[i] = row[i];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int m = 0; m < NX; ++m) out[nvc + k * DN + i * NX + m] = X[i * NX + m] / pcg->sz[m];
  if (lam_g) {
    // the engine's defect is in scaled states, x_{k+1}/s - Phi/s; the reference's rows are those of the scaled model
    // (base.py:1562-1591): G_s = dt f(X s)/s - sum C X_s, i.e. row m of G_s = row m of G / s_m  =>  mu_s = mu * s_m, and the
    // continuity multiplier is the engine's lambda as it stands
#pragma unroll
    for (int m = 0; m < NX; ++m) lam[m] = lamc[b * (int64_t)(N * NX) + k * NX + m] / pcg->sz[m];
    if (k == N - 1 && (pcg->flags & 1)) {
      // the engine reports the last defect multiplier in the reference's convention (terminal cost on the integrated end
      // state, mpc.py:1682): lambda_ref = lambda + grad V(x_N); the collocation rows need the engine's own lambda
      using PB = NmpcTrack<M>;
#pragma unroll
      for (int m = 0; m < NX; ++m) {
        double gv = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j)
          gv += (pcg->cost[PB::O_WN + m * NX + j] + pcg->cost[PB::O_WN + j * NX + m]) * (row[N * NX + j] - pcg->cost[PB::O_XREFN + j]);
        lam[m] -= gv / pcg->sz[m];
      }
    }
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

template <class M, int D>
static int coll_launch(const GenLaunchArgs& a) { return gen_launch<NmpcColl<M, D>>(a); }
template <class M, int D>
static size_t coll_lds(int N) { return Ocp<NmpcColl<M, D>>::lds_doubles(N) * sizeof(double); }
template <class M, int D>
static int coll_output(const OcpConst* dev, int64_t batch, int N, const double* vc, const double* lamc, const double* par,
                       int64_t par_stride, double* v, double* lam_g, hipStream_t s) {
  const int64_t tot = batch * N;
  hipLaunchKernelGGL((coll_output_kernel<M, D>), dim3((unsigned)((tot + 63) / 64)), dim3(64), 0, s, dev, batch, vc, lamc, par,
                     par_stride, v, lam_g);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

const CollVariant* nmpc_coll_find(int model_id, int degree) {
  static const CollVariant v[] = {
      {HILO_MODEL_CHEMOSTAT4, 3, &coll_lds<Chemostat4, 3>, &coll_launch<Chemostat4, 3>, &coll_output<Chemostat4, 3>},
      {HILO_MODEL_PENDULUM4, 3, &coll_lds<Pendulum4, 3>, &coll_launch<Pendulum4, 3>, &coll_output<Pendulum4, 3>},
      {HILO_MODEL_CSTR3, 3, &coll_lds<Cstr3, 3>, &coll_launch<Cstr3, 3>, &coll_output<Cstr3, 3>},
      {HILO_MODEL_CHEMOSTAT4, 2, &coll_lds<Chemostat4, 2>, &coll_launch<Chemostat4, 2>, &coll_output<Chemostat4, 2>},
      {HILO_MODEL_CHEMOSTAT4, 4, &coll_lds<Chemostat4, 4>, &coll_launch<Chemostat4, 4>, &coll_output<Chemostat4, 4>},
      {HILO_MODEL_CHEMOSTAT4, 1, &coll_lds<Chemostat4, 1>, &coll_launch<Chemostat4, 1>, &coll_output<Chemostat4, 1>},
  };
  for (const auto& c : v)
    if (c.model_id == model_id && c.degree == degree) return &c;
  return nullptr;
}

}  // namespace hilo
