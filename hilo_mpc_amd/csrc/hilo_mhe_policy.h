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
#include "hilo_expr.h"
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
    for (int i = 0; i < NX; ++i) xn[i] = xo[i] * rcp_fast(pc.sz[i]) + w[i];  // mhe.py:739: scaled noise, scaled state
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

// ---- the estimator in general: ESTIMATED parameters and / or NO state noise, models compiled at run time -------------------------
// What the reference's own tests configure (tests/test_MHE.py:20-110, :150-230: continuous model, default collocation, no state
// noise, `quad_arrival_cost.add_parameters`): mhe.py:596-790 with
//   v = [p | x_0..x_N | w_0..w_{N-1} (only with state noise, :599) | ip]          J(k = 0) = arrival(x_0, p)  (modeling.py:747-777)
// In engine terms: the model parameters ride along as constant extra states (p_{k+1} = p_k; estimated ones free at stage 0 and boxed
// there, pinned ones fixed through x_0 - the form of hilo_mhe_est.hip, same argument for the equality of the iterates), controls :=
// the noise (NOISE) or none at all (the trajectory is a function of (x_0, p): NU = 0).  Collocation: the augmented model
// [x' = f(x, u, p); p' = 0] goes through hilo_colloc.h (the Taylor coefficients with respect to p come with it).
//   pc.cost = [Wx (MX^2) | Wp (NP^2) | Wy | Ww | su];  par = [x_arrival | p_arrival];  sd_k = [u_meas_k | y_meas_k]
template <class M>
struct MheParAug {   // [x | p] with p' = 0 / p+ = p
  static constexpr int NX = M::NX + M::NP, NU = M::NU, NP = 0, NY = M::NY;
  static constexpr bool DISCRETE = M::DISCRETE;
  template <class T, class U, class P>
  __device__ __forceinline__ static void ode(const T* x, const U* u, const P*, double dt, T* dx) {
    M::ode(x, u, x + M::NX, dt, dx);
#pragma unroll
    for (int j = 0; j < M::NP; ++j) dx[M::NX + j] = M::DISCRETE ? x[M::NX + j] : T(0.0);
  }
};

// F for estimators without a stage constraint
struct NoMheFun {
  static constexpr int NEXPR = 0;
};

// Stage constraint of the estimator (`mhe.stage_constraint.constraint = ...`, mhe.py:498-508): HARD rows lb <= c(x, p) <= ub at
// every node k < N (mhe.py:749-757) and - under collocation - at every collocation point (mhe.py:536-550), like the controller's
// (hilo_nmpc_user.h).  QUIRKS restated: the expression is evaluated on the NLP's SCALED variables (the estimator never calls the
// constraint's `_check_and_setup`, so no scaling is substituted), and the soft branch cannot run in the reference (its penalty
// function is only created by that call): soft constraints are refused.  F::con(x, p, c): NEXPR expressions; row m uses expression
// pc.cost[O_ROWX + m] (one row per expression with a finite bound).
template <class M, int CD, bool NOISE, class F = NoMheFun, int NC_ = 0>
struct MheGen {
  using Model = M;
  static constexpr int MX = M::NX, NP = M::NP, NX = MX + NP, NU = NOISE ? MX : 0, NY = M::NY, MU = M::NU, NPAR = MX + NP,
                       NSD = M::NU + M::NY;
  static constexpr bool FIX_X0 = true;   // with x0_free_mask: states free, estimated parameters free, the others pinned
  static constexpr bool BIG = false;
  static constexpr int NC = NC_, NXV = NX, NX0 = NX, NU0 = NU > 0 ? NU : 1;
  static constexpr bool COOP = false;
  static constexpr bool QUAD_COST = false;
  // rows at the collocation points need the collocation states: evaluated together with the shooting map and the stage cost
  static constexpr bool FUSED = NC_ > 0 && CD > 0, FUSED_CON = FUSED;
  static constexpr int O_WX = 0, O_WP = O_WX + MX * MX, O_WY = O_WP + NP * NP, O_WW = O_WY + NY * NY, O_SU = O_WW + MX * MX,
                       O_ROWX = O_SU + MU, O_NROW = O_ROWX + OCP_MAXNC, O_NCR = O_NROW + 1, O_RREF = O_NCR + 1,
                       O_END = O_RREF + OCP_MAXNC;
  static constexpr int NCOST = O_END;
  static_assert(CD == 0 || !M::DISCRETE, "collocation needs the continuous model");
  static_assert(NCOST <= OCP_NCOST, "cost block too small");
  using MA = MheParAug<M>;

  // x_{k+1} of the interval (scaled), optionally the collocation states (un-scaled, augmented [x | p])
  template <class T, class E>
  __device__ __forceinline__ static void map(const OcpConst& pc, const double* sd, const T* x, const T* w, T* xn, T* Xc, const E& ext) {
    T xa[NX], xo[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xa[i] = x[i] * pc.sz[i];
    if constexpr (CD > 0) {
      T ue[MU > 0 ? MU : 1];
#pragma unroll
      for (int i = 0; i < MU; ++i) ue[i] = T(sd[i] * pc.cost[O_SU + i]);
      Colloc<MA, CD>::step(pc.coll, xa, ue, (const double*)nullptr, pc.dt, xo, Xc);
    } else {
      double ue[MU > 0 ? MU : 1];
#pragma unroll
      for (int i = 0; i < MU; ++i) ue[i] = sd[i] * pc.cost[O_SU + i];
      model_step<M>(pc.order, pc.nsub, xa, ue, xa + MX, pc.dt, xo, ext);
    }
#pragma unroll
    for (int i = 0; i < MX; ++i) {
      T v = xo[i] * rcp_fast(pc.sz[i]);
      if constexpr (NOISE) v = v + w[i];                       // mhe.py:731 / :736: scaled noise, scaled state
      xn[i] = v;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) xn[MX + j] = x[MX + j];
  }

  template <class T, class E>
  __device__ __forceinline__ static void dyn(const OcpConst& pc, const double*, const double* sd, int, const T* x, const T* w,
                                             T* xn, const E& ext) {
    map(pc, sd, x, w, xn, (T*)nullptr, ext);
  }

  // rows of ONE point: xs = SCALED [x | p] of the point, d[m0 + r], r < nrow
  template <class T>
  __device__ __forceinline__ static void rows_at(const OcpConst& pc, const T* xs, int m0, int nrow, T* d) {
    if constexpr (F::NEXPR > 0) {
      T ce[F::NEXPR];
      F::con(xs, xs + MX, ce);
#pragma unroll
      for (int m = 0; m < (NC > 0 ? NC : 1); ++m) {
        const int r = m - m0;
        if (m < NC && r >= 0 && r < nrow) d[m] = pick<F::NEXPR>(ce, (int)pc.cost[O_ROWX + r]);
      }
    }
  }
  template <class T>
  __device__ __forceinline__ static void con(const OcpConst& pc, const double*, const double*, int, const T* x, const T*, const T*,
                                             T* d) {
    rows_at(pc, x, 0, pc.nc, d);
  }
  template <class T, class E>
  __device__ __forceinline__ static T dyn_cost(const OcpConst& pc, const double* par, const double* sd, int k, const T* x,
                                               const T* w, T* xn, const E& ext) {
    map(pc, sd, x, w, xn, (T*)nullptr, ext);
    return stage_cost(pc, par, sd, k, x, w);
  }
  template <class T, class E>
  __device__ __forceinline__ static T dyn_cost_con(const OcpConst& pc, const double* par, const double* sd, int k, const T* x,
                                                   const T* w, T* xn, T* dv, const E& ext) {
    constexpr int DD = CD > 0 ? CD : 1;
    T Xc[DD * NX];
    map(pc, sd, x, w, xn, Xc, ext);
    const int nrow = (int)pc.cost[O_NROW];
    rows_at(pc, x, 0, nrow, dv);                                   // the node (mhe.py:749-757)
    if constexpr (CD > 0) {
#pragma unroll
      for (int i = 0; i < CD; ++i) {                               // the collocation points (mhe.py:536-550): scaled like the states
        T xs[NX];
#pragma unroll
        for (int a = 0; a < NX; ++a) xs[a] = Xc[i * NX + a] * rcp_fast(pc.sz[a]);
        rows_at(pc, xs, (i + 1) * nrow, nrow, dv);
      }
    }
    return stage_cost(pc, par, sd, k, x, w);
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
    if constexpr (NOISE) {
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
    }
    return acc;
  }

  template <class T>
  __device__ __forceinline__ static T term_cost(const OcpConst&, const double*, const double*, const T*) { return T(0.0); }
};

// Output pass of the general estimator, one thread per (instance, interval k < N): from the engine's result in ENGINE layout -
// ve = [xa_0..xa_N (MX + NP each) | w], lame = per interval [MX + NP defect multipliers | nc row multipliers (node rows, then the
// collocation points')] - to the reference's
//   v     = [p (the stage-0 copy, scaled) | x_0..x_N | w (NOISE) | ip_0..ip_{N-1} (collocation)]            mhe.py:614-671
//   lam_g = per interval [rows at the collocation points (D R) | collocation rows (D MX) | continuity (MX) | rows at the node (R)]
//                                                                                                          mhe.py:536-553, :728-757
// and x_opt = x_N un-scaled (mhe.py:381-384).  The collocation states and the multipliers of their rows are rebuilt with the model's
// own MX x MX blocks at the ESTIMATED parameter values (the parameter rows of the augmented system decouple, hilo_colloc.h); a row
// at a collocation point enters the stationarity of that collocation state: G_X^T mu = D_i lambda - sum_r nu_{i,r} grad c_r(X_i).
template <class M, int D, bool NOISE, class F = NoMheFun, int NC_ = 0>
__device__ __forceinline__ void mhe_gen_output(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ ve,
                                               const double* __restrict__ lame, const double* __restrict__ sdata, int64_t sd_stride,
                                               double* __restrict__ v, double* __restrict__ lam_g, double* __restrict__ x_opt) {
  using PB = MheGen<M, D, NOISE, F, NC_>;
  constexpr int MX = M::NX, MU = M::NU, NP = M::NP, NXA = MX + NP, DD = D > 0 ? D : 1, DN = DD * MX, NSD = PB::NSD;
  constexpr int NEX = F::NEXPR > 0 ? F::NEXPR : 1;
  const int N = pcg->N;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= batch * N) return;
  const int64_t b = e / N;
  const int k = (int)(e - b * N);
  const int nve = (N + 1) * NXA + (NOISE ? N * MX : 0), nw = NOISE ? N * MX : 0;
  const int nv = NP + (N + 1) * MX + nw + (D > 0 ? N * DN : 0);
  const int ncc = NC_ > 0 ? pcg->nc : 0;                                     // the engine's rows per interval
  const int nrow = NC_ > 0 ? (int)pcg->cost[PB::O_NROW] : 0, R = NC_ > 0 ? (int)pcg->cost[PB::O_NCR] : 0;
  const double* row = ve + b * nve;
  double* out = v + b * nv;
  // head: p, x_k (and x_N by the last interval's thread), w_k
#pragma unroll
  for (int j = 0; j < NP; ++j)
    if (k == 0) out[j] = row[MX + j];
#pragma unroll
  for (int i = 0; i < MX; ++i) {
    out[NP + k * MX + i] = row[k * NXA + i];
    if (k == N - 1) {
      const double xv = row[N * NXA + i];
      out[NP + N * MX + i] = xv;
      if (x_opt) x_opt[b * MX + i] = xv * pcg->sz[i];
    }
    if constexpr (NOISE) out[NP + (N + 1) * MX + k * MX + i] = row[(N + 1) * NXA + k * MX + i];
  }
  const int DR = D > 0 ? DD * R : 0, DC = D > 0 ? DN : 0;
  const int rows = DR + DC + MX + R;
  double* lg = lam_g ? lam_g + b * (int64_t)(N * rows) + (int64_t)k * rows : nullptr;
  const double* lr = lame ? lame + b * (int64_t)(N * (NXA + ncc)) + (int64_t)k * (NXA + ncc) : nullptr;
  const double* nu = lr ? lr + NXA : nullptr;
  if constexpr (D > 0) {
    double x[MX], u[MU > 0 ? MU : 1], p[NP > 0 ? NP : 1], X[DN], mat[DN * DN];
#pragma unroll
    for (int i = 0; i < MX; ++i) x[i] = row[k * NXA + i] * pcg->sz[i];
#pragma unroll
    for (int i = 0; i < MU; ++i) u[i] = sdata[b * sd_stride + k * NSD + i] * pcg->cost[PB::O_SU + i];
#pragma unroll
    for (int j = 0; j < NP; ++j) p[j] = row[MX + j] * pcg->sz[MX + j];
    Colloc<M, DD>::solve(pcg->coll, x, u, p, pcg->dt, X, mat);
#pragma unroll
    for (int i = 0; i < DD; ++i)
#pragma unroll
      for (int m = 0; m < MX; ++m) out[NP + (N + 1) * MX + nw + k * DN + i * MX + m] = X[i * MX + m] / pcg->sz[m];
    if (lg) {
      double y[DN], F_[DN];
#pragma unroll
      for (int i = 0; i < DD; ++i) {
#pragma unroll
        for (int m = 0; m < MX; ++m) y[i * MX + m] = pcg->coll.Dc[i + 1] * lr[m] / pcg->sz[m];
        if constexpr (NC_ > 0) {
          // - sum_r nu_{i,r} grad c_r at the collocation state: the expression acts on the SCALED variables; with the rows of the
          // scaled model G_s = G / s the right-hand side in un-scaled units is (D_i lambda - sum nu dc/dx_s) / s
          Dual<MX> xs[NXA], ce[NEX];
#pragma unroll
          for (int a = 0; a < MX; ++a) {
            xs[a] = Dual<MX>(X[i * MX + a] / pcg->sz[a]);
            xs[a].d[a] = 1.0;
          }
#pragma unroll
          for (int j = 0; j < NP; ++j) xs[MX + j] = Dual<MX>(row[MX + j]);
          F::con(xs, xs + MX, ce);
          for (int r = 0; r < nrow; ++r) {
            const double w8 = nu[(i + 1) * nrow + r];
            const Dual<MX> c = pick<NEX>(ce, (int)pcg->cost[PB::O_ROWX + r]);
#pragma unroll
            for (int a = 0; a < MX; ++a) y[i * MX + a] -= w8 * c.d[a] / pcg->sz[a];
          }
        }
      }
      Colloc<M, DD>::newton_matrix(pcg->coll, X, u, p, pcg->dt, mat, F_);
      Colloc<M, DD>::lu(mat);
      Colloc<M, DD>::lu_solve_t(mat, y);
#pragma unroll
      for (int i = 0; i < DD; ++i)
#pragma unroll
        for (int a = 0; a < MX; ++a) {
          double s = 0.0;
#pragma unroll
          for (int j = 0; j < DD; ++j) s -= pcg->coll.A[j * DD + i] * y[j * MX + a];
          lg[DR + i * MX + a] = s * pcg->sz[a];
        }
    }
  }
  if (lg) {
#pragma unroll
    for (int m = 0; m < MX; ++m) lg[DR + DC + m] = lr[m];
    if constexpr (NC_ > 0) {
      for (int q = 0; q < DR; ++q) lg[q] = 0.0;
      for (int q = 0; q < R; ++q) lg[DR + DC + MX + q] = 0.0;
      for (int r = 0; r < nrow; ++r) {
        const int ref = (int)pcg->cost[PB::O_RREF + r];
        lg[DR + DC + MX + ref] = nu[r];
        if constexpr (D > 0) {
          for (int i = 0; i < DD; ++i) lg[i * R + ref] = nu[(i + 1) * nrow + r];
        }
      }
    }
  }
}

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
