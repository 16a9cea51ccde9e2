// General NMPC policy for the stage-structured interior-point engine: path following and nonlinear stage constraints.
//
// Restated from hilo_mpc/modules/controller/mpc.py (pre-discretised model + `integration_method='discrete'`):
//   * path following (:1025-1053, :1173-1204): the path variable theta is appended to the model's states with its own
//     virtual input, theta+ = theta + dt u_theta (explicit Euler for a discrete model, :1188-1191); `optimize` pins only
//     the ORIGINAL states of x_0 (:785-789), so theta_0 is a free bounded variable.  Path cost
//     (s - r(theta))^T W (s - r(theta)) with the reference an expression of theta (hilo_mpc/util/modeling.py:252-283).
//   * stage constraints (`GenericConstraint`, modeling.py:820-1005; mpc.py:1271-1283, :1700-1725): rows
//     lb <= c(x_k,u_k) <= ub, k = 0..N-1, on un-scaled variables; soft: rows c - e <= ub, -c - e <= -lb with ONE slack
//     vector e >= 0 shared by all stages (:1529-1537) and e^T W e added once per stage (:1708).
// The shared slack is carried as NE constant extra states (e+ = e), free at stage 0 and boxed only there: the linear
// copies stay exactly consistent under Newton steps, so the iterates are those of the reference's single-e problem.
// Engine state = [model x (MX) | theta (NTH) | e (NE)], engine input = [model u (MU) | u_theta (NTH)].
#pragma once
#ifndef __HIPCC_RTC__
#include <stdlib.h>
#endif
#include "hilo_expr.h"
#include "hilo_ocp.h"

namespace hilo {

constexpr int GEN_NPT = 4;    // path terms per cost (stage / terminal)
constexpr int GEN_NEXPR = 2;  // constraint expressions

template <class M, int NTH_, int NE_, int NC_, bool BIG_ = false>
struct NmpcGen {
  static constexpr int MX = M::NX, MU = M::NU, NTH = NTH_, NE = NE_;
  static constexpr int NX = MX + NTH + NE, NU = MU + NTH, NZ = NX + NU, NPAR = M::NP + M::NU, NSD = 0;
  static constexpr int NC = NC_, NXV = MX + NTH, NX0 = MX, NU0 = MU;
  static constexpr bool FIX_X0 = true;
  static constexpr bool COOP = false;
  static constexpr bool BIG = BIG_;  // iterate in a global-memory workspace instead of LDS (long horizons)
  static constexpr bool QUAD_COST = NTH == 0;  // path terms are nonlinear in theta: Taylor evaluation of the cost
  // pc.cost layout
  static constexpr int O_WZ = 0, O_ZREF = O_WZ + NZ * NZ, O_WN = O_ZREF + NZ, O_XREFN = O_WN + NX * NX,
                       O_WDU = O_XREFN + NX, O_HASDU = O_WDU + MU * MU,
                       O_NPS = O_HASDU + 1, O_NPT = O_NPS + 1, O_IDXS = O_NPT + 1, O_WS = O_IDXS + GEN_NPT,
                       O_IDXT = O_WS + GEN_NPT * GEN_NPT, O_WT = O_IDXT + GEN_NPT,
                       O_NEXPR = O_WT + GEN_NPT * GEN_NPT, O_NTEXPR = O_NEXPR + 1, O_ROWX = O_NTEXPR + 1, O_ROWS = O_ROWX + OCP_MAXNC,
                       O_ROWE = O_ROWS + OCP_MAXNC, O_TSOFT = O_ROWE + OCP_MAXNC, O_TROWX = O_TSOFT + 1,
                       O_TROWS = O_TROWX + OCP_MAXNC, O_TROWE = O_TROWS + OCP_MAXNC, O_PROG = O_TROWE + OCP_MAXNC;
  static_assert(O_PROG + 64 <= OCP_NCOST, "cost block too small");
  static constexpr int NCOST = OCP_NCOST;  // expression programs have run-time length: the whole block

  template <class T, class E>
  __device__ __forceinline__ static void dyn(const OcpConst& pc, const double* par, const double*, int, const T* x,
                                             const T* u, T* xn, const E& ext) {
    T xp[MX], up[MU > 0 ? MU : 1], xo[MX];
#pragma unroll
    for (int i = 0; i < MX; ++i) xp[i] = x[i] * pc.sz[i];
#pragma unroll
    for (int i = 0; i < MU; ++i) up[i] = u[i] * pc.sz[NX + i];
    model_step<M>(pc.order, pc.nsub, xp, up, par, pc.dt, xo, ext);
#pragma unroll
    for (int i = 0; i < MX; ++i) xn[i] = xo[i] * rcp_fast(pc.sz[i]);
    if constexpr (NTH > 0) xn[MX] = x[MX] + pc.dt * u[MU];  // mpc.py:1191
#pragma unroll
    for (int e = 0; e < NE; ++e) xn[MX + NTH + e] = x[MX + NTH + e];
  }

  // sum over path terms: (x[idx] - r(theta))^T W (x[idx] - r(theta)); programs start at `first`
  template <class T>
  __device__ __forceinline__ static T path_cost(const OcpConst& pc, const double* par, int n, int o_idx, int o_w, int first,
                                                const T* x) {
    T d[GEN_NPT];
    const double* prog = expr_program(pc.cost + O_PROG, first);
#pragma unroll
    for (int a = 0; a < GEN_NPT; ++a) {
      d[a] = T(0.0);
      if (a < n) {
        const T r = expr_eval<NX, 1>(prog, x, x, par);
        d[a] = pick<NX>(x, (int)pc.cost[o_idx + a]) - r;
        prog += 1 + (int)prog[0];
      }
    }
    T acc = T(0.0);
#pragma unroll
    for (int a = 0; a < GEN_NPT; ++a) {
      T s = T(0.0);
#pragma unroll
      for (int b = 0; b < GEN_NPT; ++b) s = s + pc.cost[o_w + a * GEN_NPT + b] * d[b];
      acc = acc + d[a] * s;
    }
    return acc;
  }

  template <class T>
  __device__ __forceinline__ static T stage_cost(const OcpConst& pc, const double* par, const double*, int k,
                                                 const T* x, const T* u) {
    T z[NZ];
#pragma unroll
    for (int i = 0; i < NX; ++i) z[i] = x[i] - pc.cost[O_ZREF + i];
#pragma unroll
    for (int i = 0; i < NU; ++i) z[NX + i] = u[i] - pc.cost[O_ZREF + NX + i];
    T acc = T(0.0);
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      T s = T(0.0);
#pragma unroll
      for (int j = 0; j < NZ; ++j) s = s + pc.cost[O_WZ + i * NZ + j] * z[j];
      acc = acc + z[i] * s;
    }
    if (k == 0 && pc.cost[O_HASDU] != 0.0) {  // mpc.py:1631-1635
      T d[MU > 0 ? MU : 1];
#pragma unroll
      for (int i = 0; i < MU; ++i) d[i] = u[i] - par[M::NP + i];
#pragma unroll
      for (int i = 0; i < MU; ++i) {
        T s = T(0.0);
#pragma unroll
        for (int j = 0; j < MU; ++j) s = s + pc.cost[O_WDU + i * MU + j] * d[j];
        acc = acc + d[i] * s;
      }
    }
    if constexpr (NTH > 0) {
      const int nps = (int)pc.cost[O_NPS];
      if (nps > 0) acc = acc + path_cost(pc, par, nps, O_IDXS, O_WS, 0, x);
    }
    return acc;
  }

  __device__ __forceinline__ static double cost_grad(const OcpConst& pc, const double* par, const double*, int k, int i, const double* z) {
    double g = 0.0;
#pragma unroll
    for (int j = 0; j < NZ; ++j) g += (pc.cost[O_WZ + i * NZ + j] + pc.cost[O_WZ + j * NZ + i]) * (z[j] - pc.cost[O_ZREF + j]);
    if (k == 0 && i >= NX && i < NX + MU && pc.cost[O_HASDU] != 0.0) {
#pragma unroll
      for (int j = 0; j < MU; ++j)
        g += (pc.cost[O_WDU + (i - NX) * MU + j] + pc.cost[O_WDU + j * MU + (i - NX)]) * (z[NX + j] - par[M::NP + j]);
    }
    return g;
  }
  __device__ __forceinline__ static double cost_hess(const OcpConst& pc, int k, int i, int j) {
    double h = pc.cost[O_WZ + i * NZ + j] + pc.cost[O_WZ + j * NZ + i];
    if (k == 0 && i >= NX && j >= NX && i < NX + MU && j < NX + MU && pc.cost[O_HASDU] != 0.0)
      h += pc.cost[O_WDU + (i - NX) * MU + (j - NX)] + pc.cost[O_WDU + (j - NX) * MU + (i - NX)];
    return h;
  }

  template <class T>
  __device__ __forceinline__ static T term_cost(const OcpConst& pc, const double* par, const double*, const T* x) {
    T z[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) z[i] = x[i] - pc.cost[O_XREFN + i];
    T acc = T(0.0);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      T s = T(0.0);
#pragma unroll
      for (int j = 0; j < NX; ++j) s = s + pc.cost[O_WN + i * NX + j] * z[j];
      acc = acc + z[i] * s;
    }
    if constexpr (NTH > 0) {
      const int npt = (int)pc.cost[O_NPT];
      if (npt > 0) acc = acc + path_cost(pc, par, npt, O_IDXT, O_WT, (int)pc.cost[O_NPS], x);
    }
    return acc;
  }

  // inequality rows d_m = sign_m c_{expr_m}(x sx, u su) - e_{slack_m}  (mpc.py:1276-1277; modeling.py:843-849); at the last
  // stage additionally the terminal rows: hard, c_T(x_end sx) on the integrated end state xn (mpc.py:1693-1700), or soft,
  // sign c_T(x_{N-1} sx) - e_T on the state the last interval starts from (mpc.py:1684-1692)
  template <class T>
  __device__ __forceinline__ static void con(const OcpConst& pc, const double* par, const double*, int k, const T* x,
                                             const T* u, const T* xn, T* d) {
    T xs[MX], us[MU > 0 ? MU : 1], ce[GEN_NEXPR];
#pragma unroll
    for (int i = 0; i < MX; ++i) xs[i] = x[i] * pc.sz[i];
#pragma unroll
    for (int i = 0; i < MU; ++i) us[i] = u[i] * pc.sz[NX + i];
    const int nexpr = (int)pc.cost[O_NEXPR];
    const double* prog = expr_program(pc.cost + O_PROG, (int)pc.cost[O_NPS] + (int)pc.cost[O_NPT]);
#pragma unroll
    for (int j = 0; j < GEN_NEXPR; ++j) {
      ce[j] = T(0.0);
      if (j < nexpr) {
        ce[j] = expr_eval<MX, MU>(prog, xs, us, par);
        prog += 1 + (int)prog[0];
      }
    }
#pragma unroll
    for (int m = 0; m < (NC > 0 ? NC : 1); ++m) {
      if (m < NC) {
        T v = pc.cost[O_ROWS + m] * pick<GEN_NEXPR>(ce, (int)pc.cost[O_ROWX + m]);
        if constexpr (NE > 0) {
          const int ei = (int)pc.cost[O_ROWE + m];
          if (ei >= 0) v = v - pick<NE>(x + MX + NTH, ei);
        }
        d[m] = v;
      }
    }
    const int nte = (int)pc.cost[O_NTEXPR];
    if (nte > 0 && k == pc.N - 1) {
      const bool soft = pc.cost[O_TSOFT] != 0.0;
      T xe[MX], ct[GEN_NEXPR];
#pragma unroll
      for (int i = 0; i < MX; ++i) xe[i] = soft ? xs[i] : xn[i] * pc.sz[i];
#pragma unroll
      for (int j = 0; j < GEN_NEXPR; ++j) {
        ct[j] = T(0.0);
        if (j < nte) {
          ct[j] = expr_eval<MX, MU>(prog, xe, us, par);
          prog += 1 + (int)prog[0];
        }
      }
#pragma unroll
      for (int m = 0; m < (NC > 0 ? NC : 1); ++m) {
        const int r = m - pc.nc;   // terminal row index
        if (m < NC && r >= 0 && r < pc.nc_term) {
          T v = pc.cost[O_TROWS + r] * pick<GEN_NEXPR>(ct, (int)pc.cost[O_TROWX + r]);
          if constexpr (NE > 0) {
            const int ei = (int)pc.cost[O_TROWE + r];
            if (ei >= 0) v = v - pick<NE>(x + MX + NTH, ei);
          }
          d[m] = v;
        }
      }
    }
  }
};

#ifndef __HIPCC_RTC__
// host-side description of one general instantiation (filled per (model, NTH, NE, NC) in hilo_nmpc_gen_*.hip)
struct GenLaunchArgs {
  const OcpConst* dev;
  int64_t batch;
  const double *x0, *par;
  int64_t par_stride;
  const double* v0;
  int64_t v0_stride;
  double *v_opt, *f_opt, *lam_g, *u0;
  int32_t *status, *iters;
  double* kkt;
  long long* prof;
  size_t lds_bytes;
  hipStream_t stream;
  double* ws;   // per-instance workspace (BIG variants) or NULL
  OcpExtra ex = OcpExtra();
};
struct GenVariant {
  int model_id, nth, ne, nc, big;            // key
  int nx, nu, nxv, mx, mu, np;               // engine / reference dimensions
  int o_wz, o_zref, o_wn, o_xrefn, o_wdu, o_hasdu, o_nps, o_npt, o_idxs, o_ws, o_idxt, o_wt, o_nexpr, o_rowx, o_rows,
      o_rowe, o_prog, o_ntexpr, o_tsoft, o_trowx, o_trows, o_trowe;
  size_t (*lds_bytes)(int N);
  size_t (*ws_bytes)(int N);
  int (*launch)(const GenLaunchArgs& a);
};
// smallest variant that covers the request and whose LDS footprint at horizon N fits (LDS variants are preferred)
const GenVariant* nmpc_gen_find(int model_id, int nth, int ne, int nc_needed, int N);

// collocation variants of the tracking policy (hilo_nmpc_coll.hip)
struct CollVariant {
  int model_id, degree;
  size_t (*lds_bytes)(int N);
  int (*launch)(const GenLaunchArgs& a);
  // v = [compact v | collocation states], lam_g = per stage [collocation rows | continuity] from the engine's compact output
  int (*output)(const OcpConst* dev, int64_t batch, int N, const double* vc, const double* lamc, const double* par,
                int64_t par_stride, double* v, double* lam_g, hipStream_t s);
};
const CollVariant* nmpc_coll_find(int model_id, int degree);

// long-horizon variants of the tracking policy: iterate in a global-memory workspace (hilo_nmpc_long.hip)
struct TrackBigVariant {
  int model_id;
  size_t (*lds_bytes)(int N);
  size_t (*ws_bytes)(int N);
  int (*launch)(const GenLaunchArgs& a);
};
const TrackBigVariant* nmpc_track_big_find(int model_id);

// per-stage-data variants of the tracking policy (hilo_nmpc_tv.hip)
struct TvVariant {
  int model_id;
  size_t (*lds_bytes)(int N);
  int (*launch)(const GenLaunchArgs& a, const double* stage_data, int64_t sd_stride);
};
const TvVariant* nmpc_tv_find(int model_id);

// workgroups of a workspace-mode launch: by default one per instance.  HILO_BIG_SLOTS=n makes n workgroups walk over the
// instances with one cache-resident workspace slot each - measured on C5 (B = 8192): 1024 slots 420 ms, 512: 655 ms, 256: 1164 ms
// against 392 ms with one workgroup per instance, i.e. the kernel is bound by per-wave latency, not by workspace bandwidth
inline int64_t big_grid_slots() {
  static int64_t slots = 0;
  if (!slots) {
    const char* e = getenv("HILO_BIG_SLOTS");
    slots = e && atoll(e) > 0 ? atoll(e) : (int64_t)1 << 40;
  }
  return slots;
}

template <class PB>
int gen_launch(const GenLaunchArgs& a) {
  if (a.lds_bytes > 64 * 1024)
    HILO_HIP_CHECK(hipFuncSetAttribute((const void*)ocp_solve_kernel<PB, OCP_TPB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)a.lds_bytes));
  unsigned grid = (unsigned)a.batch;
  if constexpr (PB::BIG) grid = (unsigned)(a.batch < big_grid_slots() ? a.batch : big_grid_slots());
  hipLaunchKernelGGL((ocp_solve_kernel<PB, OCP_TPB>), dim3(grid), dim3(OCP_TPB), a.lds_bytes, a.stream, a.dev,
                     a.batch, a.x0, a.par, a.par_stride, (const double*)nullptr, (int64_t)0, a.v0, a.v0_stride, 0, 0, a.v_opt,
                     a.f_opt, a.lam_g, a.u0, 0, a.status, a.iters, a.kkt, a.prof, a.ws, a.ex);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}
template <class PB>
size_t gen_lds(int N) { return Ocp<PB>::lds_doubles(N) * sizeof(double); }
template <class PB>
size_t gen_ws(int N) { return Ocp<PB>::ws_doubles(N) * sizeof(double); }

template <class M, int NTH, int NE, int NC, bool BIG = false>
GenVariant gen_variant(int model_id) {
  using PB = NmpcGen<M, NTH, NE, NC, BIG>;
  return GenVariant{model_id, NTH, NE, NC, BIG ? 1 : 0, PB::NX, PB::NU, PB::NXV, PB::MX, PB::MU, M::NP,
                    PB::O_WZ, PB::O_ZREF, PB::O_WN, PB::O_XREFN, PB::O_WDU, PB::O_HASDU, PB::O_NPS, PB::O_NPT, PB::O_IDXS,
                    PB::O_WS, PB::O_IDXT, PB::O_WT, PB::O_NEXPR, PB::O_ROWX, PB::O_ROWS, PB::O_ROWE, PB::O_PROG, PB::O_NTEXPR,
                    PB::O_TSOFT, PB::O_TROWX, PB::O_TROWS, PB::O_TROWE,
                    &gen_lds<PB>, &gen_ws<PB>, &gen_launch<PB>};
}
#endif  // !__HIPCC_RTC__

}  // namespace hilo
