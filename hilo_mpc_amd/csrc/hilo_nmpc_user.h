// The general NMPC policy of the interior-point engine for problems compiled at RUN TIME (hilo_jit.hip, hiprtc).
//
// The reference accepts any model and any cost / constraint expression because everything is a CasADi graph
// (`Model.set_dynamical_equations`, hilo_mpc/modules/dynamic_model/dynamic_model.py:1293-1553; `nmpc.stage_cost.cost = ...`,
// hilo_mpc/util/modeling.py:38-87; `nmpc.stage_constraint.constraint = ...`, modeling.py:930-940).  Here the host turns the
// same expressions into the source of two functors
//     M   the model, in the shape of the zoo functors of hilo_models.h (`ode`, `meas`, templated on the scalar type)
//     F   the problem's free-form functions: generic stage / terminal cost, constraint expressions, path references
// and a configuration struct C of compile-time switches; the translation unit `#include`s this header, instantiates
// NmpcUser<M, F, C> and is compiled against the engine at `NMPC.setup()`.
//
// Restated from hilo_mpc/modules/controller/mpc.py:1133-1787 (everything the precompiled policies NmpcTrack / NmpcGen /
// NmpcColl / NmpcTv cover, in one place), plus
//   * GenericCost (mpc.py:213-218): QUIRK - the expression is attached to the model AFTER `_scale_problem()` (mpc.py:1210
//     then :1283) and `Model.scale` only substitutes inside the model's own equations (base.py:1169-1179), so the cost's
//     symbols are the NLP's SCALED variables: F::stage / F::term receive (x / x_scaling, u / u_scaling);
//   * the continuous objective (optimizer.py:1423-1426, the default for a continuous model): the Lagrange term is integrated
//     with the shooting map - collocation: dt sum_{i=1..d} B_i l(x_{k,i}, u_k) (modeling.py:1195); explicit Runge-Kutta:
//     h sum_i b_i l(X_i, u_k) over the stage points (modeling.py:1263-1275);
//   * a control horizon Nc < N (mpc.py:1629-1630: `u_ii` stays the last input): the inputs are carried as NH extra states
//     uh_{k+1} = u_k (k < Nc) / uh_k (k >= Nc); stages k >= Nc evaluate the model and the cost at uh_k and have no free input;
//   * the path variable as a state of the model (mpc.py:1181-1191): theta' = u_theta for a continuous model (it takes
//     part in the collocation / Runge-Kutta scheme), theta+ = theta + dt u_theta for a discrete one.
// Engine state = [model x (MX) | theta (NTH) | shared slacks e (NE) | held inputs uh (NH)], input = [model u (MU) | u_theta].
#pragma once
#include "hilo_expr.h"
#include "hilo_ocp.h"

namespace hilo {

// ---- layout of pc.cost for NmpcUser: plain function of the dimensions so that the host (hilo_jit.hip) fills the block ----
struct UserLayout {
  int mza, o_wz, o_zref, o_wn, o_xrefn, o_wdu, o_hasdu, o_we, o_wet, o_ws, o_idxs, o_wt, o_idxt, o_rowx, o_rows, o_rowe,
      o_trowx, o_trows, o_trowe, o_tsoft, o_wzm, o_end;
  __host__ __device__ constexpr UserLayout(int mx, int mu, int nth, int ne, int nps, int npt)
      : mza(mx + nth + mu + nth),
        o_wz(0),                                   // [mza x mza] weights on the (scaled) augmented z = [x, theta | u, u_theta]
        o_zref(o_wz + mza * mza),                  // [mza]
        o_wn(o_zref + mza),                        // [(mx+nth) x (mx+nth)] terminal weights
        o_xrefn(o_wn + (mx + nth) * (mx + nth)),   // [mx+nth]
        o_wdu(o_xrefn + mx + nth),                 // [mu x mu] input-change weights (interval 0 only, mpc.py:1631-1635)
        o_hasdu(o_wdu + mu * mu),
        o_we(o_hasdu + 1),                         // [ne x ne] stage penalty of the shared slacks, once per stage (mpc.py:1708)
        o_wet(o_we + ne * ne),                     // [ne x ne] terminal penalty (slack of a soft terminal constraint, mpc.py:1686)
        o_ws(o_wet + ne * ne),                     // [nps x nps] path-term weights (stage)
        o_idxs(o_ws + nps * nps),                  // [nps] state index of each path term
        o_wt(o_idxs + nps),                        // [npt x npt], [npt]: terminal path terms
        o_idxt(o_wt + npt * npt),
        o_rowx(o_idxt + npt),                      // per inequality row: expression index, sign, slack index (-1: none)
        o_rows(o_rowx + OCP_MAXNC),
        o_rowe(o_rows + OCP_MAXNC),
        o_trowx(o_rowe + OCP_MAXNC),
        o_trows(o_trowx + OCP_MAXNC),
        o_trowe(o_trows + OCP_MAXNC),
        o_tsoft(o_trowe + OCP_MAXNC),
        o_wzm(o_tsoft + 1),                        // [mza] per row of Wz: bit j set iff Wz[i][j] != 0 (as a double: exact below 2^53)
        o_end(o_wzm + mza) {}
};

// the model with the path variable(s) appended as states driven by virtual inputs (mpc.py:1181-1191)
template <class M, int NTH>
struct ThetaAug {
  static constexpr int NX = M::NX + NTH, NU = M::NU + NTH, NP = M::NP, NY = M::NY;
  static constexpr bool DISCRETE = M::DISCRETE;
  template <class T, class U, class P>
  __device__ __forceinline__ static void ode(const T* x, const U* u, const P* p, double dt, T* dx) {
    M::ode(x, u, p, dt, dx);
#pragma unroll
    for (int a = 0; a < NTH; ++a) {
      if constexpr (M::DISCRETE) dx[M::NX + a] = x[M::NX + a] + dt * u[M::NU + a];
      else dx[M::NX + a] = T(u[M::NU + a]);
    }
  }
};

// F for problems without free-form functions
struct NoUserFun {
  static constexpr bool HAS_STAGE = false, HAS_TERM = false;
  static constexpr int NEXPR = 0, NTEXPR = 0, NPS = 0, NPT = 0;
};

template <class M, class F, class C>
struct NmpcUser {
  static constexpr int MX = M::NX, MU = M::NU, NTH = C::NTH, NE = C::NE;
  static constexpr int MXA = MX + NTH, MUA = MU + NTH, MZA = MXA + MUA;   // augmented model
  static constexpr int NH = C::HOLD ? MUA : 0;
  static constexpr int NX = MXA + NE + NH, NU = MUA, NZ = NX + NU;
  static constexpr int NXV = MXA, NX0 = MX, NU0 = MU, NC = C::NC;
  static constexpr int NPAR = M::NP + M::NU;
  static constexpr int NSD = C::TV ? (MX + MU + M::NP) : 0;   // per stage [zref_k (model z, scaled) | p_k]; row N: terminal ref
  static constexpr bool FIX_X0 = true, COOP = false, BIG = C::BIG;
  static constexpr int VEC_N = C::N;    // the horizon is part of the compiled problem: Ocp::VEC_LDS may keep the vectors in LDS
  static constexpr int D = C::COLL_D;                          // collocation degree, 0 = explicit Runge-Kutta / discrete map
  static constexpr bool CONT = C::CONT;                        // continuous objective
  static constexpr bool FUSED = true;
  static constexpr bool QUAD_COST = false;
  static constexpr UserLayout L = UserLayout(MX, MU, NTH, NE, F::NPS, F::NPT);
  static constexpr int NCOST = L.o_end;
  static_assert(NCOST <= OCP_NCOST, "cost block too small for this problem");
  static_assert(D <= COLL_MAXD, "collocation degree");
  using MA = ThetaAug<M, NTH>;

  // ---- Lagrange term at a point of the augmented model, SCALED variables xs [MXA], us [MUA] --------------------------
  // MASKED: the quadratic form runs over the non-zero weights only (row masks prepared by the host).  The output pass of the
  // collocation transcription asks for the plain loop: there (constant seeds, twelve unrolled copies sharing the masks) the
  // masked form compiled to a wrong gradient for the weighted states with ROCm 7.2 - multipliers of the collocation rows off
  // by up to 1 % while the solve kernel's result was bit-identical to the plain loop's (tools/dbg/dae_lam.py) - cause not found.
  template <class T, bool MASKED = true>
  __device__ __forceinline__ static T lagrange(const OcpConst& pc, const double* par, const double* sd, const double* p, int k,
                                               const T* xs, const T* us) {
    T z[MZA];
#pragma unroll
    for (int i = 0; i < MXA; ++i) {
      double r = pc.cost[L.o_zref + i];
      if constexpr (C::TV) { if (i < MX) r = sd[i]; }
      z[i] = xs[i] - r;
    }
#pragma unroll
    for (int i = 0; i < MUA; ++i) {
      double r = pc.cost[L.o_zref + MXA + i];
      if constexpr (C::TV) { if (i < MU) r = sd[MX + i]; }
      z[MXA + i] = us[i] - r;
    }
    // z^T Wz z over the NON-ZERO weights only (row masks prepared by the host; wave-uniform scalar tests): a weight matrix is
    // mostly zeros - in a Taylor sweep every skipped entry is an LDS read and a three-coefficient multiply-add
    T acc = T(0.0);
#pragma unroll
    for (int i = 0; i < MZA; ++i) {
      if constexpr (MASKED) {
        const unsigned m = (unsigned)uni((int)pc.cost[L.o_wzm + i]);
        if (m != 0u) {
          T s = T(0.0);
#pragma unroll
          for (int j = 0; j < MZA; ++j)
            if ((m >> j) & 1u) s = s + pc.cost[L.o_wz + i * MZA + j] * z[j];
          acc = acc + z[i] * s;
        }
      } else {
        T s = T(0.0);
#pragma unroll
        for (int j = 0; j < MZA; ++j) s = s + pc.cost[L.o_wz + i * MZA + j] * z[j];
        acc = acc + z[i] * s;
      }
    }
    if (k == 0 && pc.cost[L.o_hasdu] != 0.0) {   // mpc.py:1631-1635: the change penalty only sees u_old in interval 0
      T d[MU > 0 ? MU : 1];
#pragma unroll
      for (int i = 0; i < MU; ++i) d[i] = us[i] - par[M::NP + i];
#pragma unroll
      for (int i = 0; i < MU; ++i) {
        T s = T(0.0);
#pragma unroll
        for (int j = 0; j < MU; ++j) s = s + pc.cost[L.o_wdu + i * MU + j] * d[j];
        acc = acc + d[i] * s;
      }
    }
    if constexpr (F::HAS_STAGE) acc = acc + F::stage(xs, us, p);
    if constexpr (F::NPS > 0) {   // (x[idx] - r(theta))^T W (x[idx] - r(theta)), modeling.py:252-283
      T r[F::NPS], d[F::NPS];
      F::path_stage(xs, p, r);
#pragma unroll
      for (int a = 0; a < F::NPS; ++a) d[a] = pick<MXA>(xs, (int)pc.cost[L.o_idxs + a]) - r[a];
#pragma unroll
      for (int a = 0; a < F::NPS; ++a) {
        T s = T(0.0);
#pragma unroll
        for (int b = 0; b < F::NPS; ++b) s = s + pc.cost[L.o_ws + a * F::NPS + b] * d[b];
        acc = acc + d[a] * s;
      }
    }
    return acc;
  }

  // one explicit Runge-Kutta step of the augmented model with the quadrature of the Lagrange term over its stage points
  // (modeling.py:1263-1275); `xp`, `up` in original units, the cost sees them scaled
  template <class T>
  __device__ __forceinline__ static void erk_quad(const OcpConst& pc, const double* par, const double* sd, const double* p, int k,
                                                  int order, const T* x, const T* up, const T* us, double h, T* xn, T& q) {
    T kk[4][MXA];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < order) {
        T xi[MXA], xis[MXA];
#pragma unroll
        for (int s = 0; s < MXA; ++s) {
          T acc = x[s];
          if (i >= 1) acc = acc + (h * (i == 1 ? erk_a<1, 0>(order) : (i == 2 ? erk_a<2, 0>(order) : 0.0))) * kk[0][s];
          if (i >= 2) acc = acc + (h * (i == 2 ? erk_a<2, 1>(order) : 0.0)) * kk[1][s];
          if (i >= 3) acc = acc + (h * erk_a<3, 2>(order)) * kk[2][s];
          xi[s] = acc;
          xis[s] = acc * (1.0 / pc.sz[s]);
        }
        MA::ode(xi, up, p, h, kk[i]);
        const double bi = i == 0 ? erk_b<0>(order) : (i == 1 ? erk_b<1>(order) : (i == 2 ? erk_b<2>(order) : erk_b<3>(order)));
        if (bi != 0.0) q = q + (h * bi) * lagrange(pc, par, sd, p, k, xis, us);
      } else {
#pragma unroll
        for (int s = 0; s < MXA; ++s) kk[i][s] = T(0.0);
      }
    }
#pragma unroll
    for (int s = 0; s < MXA; ++s)
      xn[s] = x[s] + (h * erk_b<0>(order)) * kk[0][s] + (h * erk_b<1>(order)) * kk[1][s] + (h * erk_b<2>(order)) * kk[2][s] +
              (h * erk_b<3>(order)) * kk[3][s];
  }

  // ---- shooting map + Lagrange term of interval k -------------------------------------------------------------------
  template <class T, class E>
  __device__ __forceinline__ static T dyn_cost(const OcpConst& pc, const double* par, const double* sd, int k, const T* x,
                                               const T* u, T* xn, const E&) {
    const double* p = C::TV ? sd + MX + MU : par;
    T xp[MXA], up[MUA > 0 ? MUA : 1], us[MUA > 0 ? MUA : 1], xs[MXA], xo[MXA];
#pragma unroll
    for (int i = 0; i < MXA; ++i) { xs[i] = x[i]; xp[i] = x[i] * pc.sz[i]; }
#pragma unroll
    for (int i = 0; i < MUA; ++i) {
      T ui = u[i];
      if constexpr (NH > 0) { if (k >= pc.Nc) ui = x[MXA + NE + i]; }   // held input
      us[i] = ui;
      up[i] = ui * pc.sz[NX + i];
    }
    T lc = T(0.0);
    if constexpr (D > 0) {
      T Xc[D * MXA];
      Colloc<MA, D>::step(pc.coll, xp, up, p, pc.dt, xo, CONT ? Xc : nullptr);
      if constexpr (CONT) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
          T xcs[MXA];
#pragma unroll
          for (int m = 0; m < MXA; ++m) xcs[m] = Xc[i * MXA + m] * (1.0 / pc.sz[m]);
          lc = lc + (pc.dt * pc.coll.Bq[i + 1]) * lagrange(pc, par, sd, p, k, xcs, us);
        }
      }
    } else if constexpr (M::DISCRETE) {
      MA::ode(xp, up, p, pc.dt, xo);
    } else if constexpr (CONT) {
      const double h = pc.dt / pc.nsub;
      T xc[MXA];
#pragma unroll
      for (int s = 0; s < MXA; ++s) xc[s] = xp[s];
      for (int it = 0; it < pc.nsub; ++it) {
        T xt[MXA];
        erk_quad(pc, par, sd, p, k, pc.order, xc, up, us, h, xt, lc);
#pragma unroll
        for (int s = 0; s < MXA; ++s) xc[s] = xt[s];
      }
#pragma unroll
      for (int s = 0; s < MXA; ++s) xo[s] = xc[s];
    } else {
      model_step<MA>(pc.order, pc.nsub, xp, up, p, pc.dt, xo);
    }
    if constexpr (!CONT || (D == 0 && M::DISCRETE)) lc = lagrange(pc, par, sd, p, k, xs, us);   // discrete objective: l(x_k, u_k)
#pragma unroll
    for (int i = 0; i < MXA; ++i) xn[i] = xo[i] * (1.0 / pc.sz[i]);
#pragma unroll
    for (int e = 0; e < NE; ++e) xn[MXA + e] = x[MXA + e];            // shared slacks: constant states
#pragma unroll
    for (int a = 0; a < NH; ++a) {                                     // held inputs
      T v = u[a];
      if (k >= pc.Nc) v = x[MXA + NE + a];
      xn[MXA + NE + a] = v;
    }
    if constexpr (NE > 0) {   // e^T W e once per stage, outside the quadrature (mpc.py:1708)
#pragma unroll
      for (int a = 0; a < NE; ++a) {
        T s = T(0.0);
#pragma unroll
        for (int b = 0; b < NE; ++b) s = s + pc.cost[L.o_we + a * NE + b] * x[MXA + b];
        lc = lc + x[MXA + a] * s;
      }
    }
    return lc;
  }

  template <class T>
  __device__ __forceinline__ static T term_cost(const OcpConst& pc, const double* par, const double* sd, const T* x) {
    const double* p = C::TV ? sd + MX + MU : par;
    T z[MXA];
#pragma unroll
    for (int i = 0; i < MXA; ++i) {
      double r = pc.cost[L.o_xrefn + i];
      if constexpr (C::TV) { if (i < MX) r = sd[i]; }
      z[i] = x[i] - r;
    }
    T acc = T(0.0);
#pragma unroll
    for (int i = 0; i < MXA; ++i) {
      T s = T(0.0);
#pragma unroll
      for (int j = 0; j < MXA; ++j) s = s + pc.cost[L.o_wn + i * MXA + j] * z[j];
      acc = acc + z[i] * s;
    }
    if constexpr (F::HAS_TERM) acc = acc + F::term(x, p);
    if constexpr (F::NPT > 0) {
      T r[F::NPT], d[F::NPT];
      F::path_term(x, p, r);
#pragma unroll
      for (int a = 0; a < F::NPT; ++a) d[a] = pick<MXA>(x, (int)pc.cost[L.o_idxt + a]) - r[a];
#pragma unroll
      for (int a = 0; a < F::NPT; ++a) {
        T s = T(0.0);
#pragma unroll
        for (int b = 0; b < F::NPT; ++b) s = s + pc.cost[L.o_wt + a * F::NPT + b] * d[b];
        acc = acc + d[a] * s;
      }
    }
    if constexpr (NE > 0) {   // slack of a soft terminal constraint: penalty once (mpc.py:1686)
#pragma unroll
      for (int a = 0; a < NE; ++a) {
        T s = T(0.0);
#pragma unroll
        for (int b = 0; b < NE; ++b) s = s + pc.cost[L.o_wet + a * NE + b] * x[MXA + b];
        acc = acc + x[MXA + a] * s;
      }
    }
    return acc;
  }

  // inequality rows (same construction as NmpcGen::con): d_m = sign_m c_{expr_m}(x sx, u su) - e_{slack_m}; at the last
  // stage additionally the terminal rows - hard on the integrated end state (mpc.py:1693-1700), soft on x_{N-1} (:1684-1692)
  template <class T>
  __device__ __forceinline__ static void con(const OcpConst& pc, const double* par, const double* sd, int k, const T* x,
                                             const T* u, const T* xn, T* d) {
    const double* p = C::TV ? sd + MX + MU : par;
    T xs[MXA], us[MUA > 0 ? MUA : 1];
#pragma unroll
    for (int i = 0; i < MXA; ++i) xs[i] = x[i] * pc.sz[i];
#pragma unroll
    for (int i = 0; i < MUA; ++i) {
      T ui = u[i];
      if constexpr (NH > 0) { if (k >= pc.Nc) ui = x[MXA + NE + i]; }
      us[i] = ui * pc.sz[NX + i];
    }
    if constexpr (F::NEXPR > 0) {
      T ce[F::NEXPR];
      F::con(xs, us, p, ce);
#pragma unroll
      for (int m = 0; m < (NC > 0 ? NC : 1); ++m) {
        if (m < NC && m < pc.nc) {
          T v = pc.cost[L.o_rows + m] * pick<F::NEXPR>(ce, (int)pc.cost[L.o_rowx + m]);
          if constexpr (NE > 0) {
            const int ei = (int)pc.cost[L.o_rowe + m];
            if (ei >= 0) v = v - pick<NE>(x + MXA, ei);
          }
          d[m] = v;
        }
      }
    }
    if constexpr (F::NTEXPR > 0) {
      if (k == pc.N - 1) {
        const bool soft = pc.cost[L.o_tsoft] != 0.0;
        T xe[MXA], ct[F::NTEXPR];
#pragma unroll
        for (int i = 0; i < MXA; ++i) xe[i] = soft ? xs[i] : xn[i] * pc.sz[i];
        F::tcon(xe, us, p, ct);
#pragma unroll
        for (int m = 0; m < (NC > 0 ? NC : 1); ++m) {
          const int r = m - pc.nc;
          if (m < NC && r >= 0 && r < pc.nc_term) {
            T v = pc.cost[L.o_trows + r] * pick<F::NTEXPR>(ct, (int)pc.cost[L.o_trowx + r]);
            if constexpr (NE > 0) {
              const int ei = (int)pc.cost[L.o_trowe + r];
              if (ei >= 0) v = v - pick<NE>(x + MXA, ei);
            }
            d[m] = v;
          }
        }
      }
    }
  }
  // ---- collocation output pass: one thread per (instance, interval) ---------------------------------------------------
  // The engine eliminates the collocation states (hilo_colloc.h); this reconstructs them and the multipliers of their
  // equations so that `v` = [x | u | e | ip] and `lam_g` = per stage [collocation rows | continuity] have the reference's
  // layout (mpc.py:1497-1518, :1657-1669).  Stationarity of the reference's Lagrangian in a collocation state X_i:
  //     G_X^T mu = D_i lambda - dt B_i grad l(X_i)        (the second term only with the continuous objective)
  // solved with the Runge-Kutta form of the equations, mu = -(A^T (x) I) Mat^-T rhs (hilo_colloc.h::multipliers).
  __device__ static void coll_output(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ vc,
                                     const double* __restrict__ lamc, const double* __restrict__ par, int64_t par_stride,
                                     const double* __restrict__ sdata, int64_t sd_stride, double* __restrict__ v,
                                     double* __restrict__ lam_g) {
    constexpr int DD = D > 0 ? D : 1, DN = DD * MXA;
    // semi-explicit DAE model (codegen.py::dae_model_source): the reference carries the algebraic states as variables - node
    // blocks z_0..z_N behind the inputs (they enter no constraint: the guess), zp_k = z at the d collocation points behind the
    // interval's collocation states - and the algebraic rows behind each collocation point's rows (mpc.py:1488-1518,
    // modeling.py:1183-1190).  The engine solved the ODE with z eliminated; z and the multipliers of its rows are rebuilt here.
    constexpr int NZA = model_nz<M>::value, DB = DN + DD * NZA, NZA1 = NZA > 0 ? NZA : 1;
    static_assert(NZA == 0 || NTH == 0, "a DAE model with a path variable is not built");
    const OcpConst& pc = *pcg;
    const int N = pc.N, Nc = NH > 0 ? pc.Nc : N;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= batch * N) return;
    const int64_t b = e / N;
    const int k = (int)(e - b * N);
    const int n_vc = (N + 1) * MXA + Nc * MUA + NE, n_zn = (N + 1) * NZA, nv = n_vc + n_zn + N * DB;
    const double* row = vc + b * n_vc;
    const double* pr = par + b * par_stride;
    const double* sd = C::TV ? sdata + b * sd_stride + (int64_t)k * NSD : nullptr;
    const double* p = C::TV ? sd + MX + MU : pr;
    double x[MXA], u[MUA > 0 ? MUA : 1], us[MUA > 0 ? MUA : 1], X[DN], mat[DN * DN];
    const int ku = k < Nc ? k : Nc - 1;
#pragma unroll
    for (int i = 0; i < MXA; ++i) x[i] = row[k * MXA + i] * pc.sz[i];
#pragma unroll
    for (int i = 0; i < MUA; ++i) {
      us[i] = row[(N + 1) * MXA + ku * MUA + i];
      u[i] = us[i] * pc.sz[NX + i];
    }
    Colloc<MA, DD>::solve(pc.coll, x, u, p, pc.dt, X, mat);
    double* out = v + b * nv;
    if (k == 0)
      for (int i = 0; i < n_vc; ++i) out[i] = row[i];
#pragma unroll
    for (int i = 0; i < DD; ++i)
#pragma unroll
      for (int m = 0; m < MXA; ++m) out[n_vc + n_zn + k * DB + i * MXA + m] = X[i * MXA + m] / pc.sz[m];
    double zc[DD * NZA1];
    if constexpr (NZA > 0) {
#pragma unroll
      for (int a = 0; a < NZA; ++a) {
        out[n_vc + k * NZA + a] = M::z_guess(a);
        if (k == N - 1) out[n_vc + N * NZA + a] = M::z_guess(a);
      }
#pragma unroll
      for (int i = 0; i < DD; ++i) {
        dae_solve<M>(X + i * MXA, u, p, zc + i * NZA);
#pragma unroll
        for (int a = 0; a < NZA; ++a) out[n_vc + n_zn + k * DB + DN + i * NZA + a] = zc[i * NZA + a];
      }
    }
    if (!lam_g) return;
    double lam[MXA], y[DN], F_[DN], mu[DN];
#pragma unroll
    for (int m = 0; m < MXA; ++m) lam[m] = lamc[b * (int64_t)(N * MXA) + k * MXA + m];
    if (k == N - 1 && (pc.flags & 1)) {
      // the engine reports the last defect multiplier in the reference's convention (terminal cost on the integrated end
      // state, mpc.py:1682): lambda_ref = lambda + grad V(x_N); the collocation rows need the engine's own lambda
      const double* sdN = C::TV ? sdata + b * sd_stride + (int64_t)N * NSD : nullptr;
#pragma unroll
      for (int m = 0; m < MXA; ++m) {
        Jet2 xj[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xj[i] = Jet2(i < MXA ? row[N * MXA + i] : 0.0, i == m ? 1.0 : 0.0, 0.0);
        lam[m] -= term_cost(pc, pr, sdN, xj).a;
      }
    }
#pragma unroll
    for (int i = 0; i < DD; ++i) {
      double gl[MXA];
#pragma unroll
      for (int m = 0; m < MXA; ++m) gl[m] = 0.0;
      if constexpr (CONT) {   // gradient of the Lagrange term at the collocation state, in scaled variables
#pragma unroll
        for (int m = 0; m < MXA; ++m) {
          Jet2 xj[MXA], uj[MUA > 0 ? MUA : 1];
#pragma unroll
          for (int q = 0; q < MXA; ++q) xj[q] = Jet2(X[i * MXA + q] / pc.sz[q], q == m ? 1.0 : 0.0, 0.0);
#pragma unroll
          for (int q = 0; q < MUA; ++q) uj[q] = Jet2(us[q]);
          gl[m] = pc.dt * pc.coll.Bq[i + 1] * lagrange<Jet2, false>(pc, pr, sd, p, k, xj, uj).a;
        }
      }
      // rows of the scaled model: G_s = G / s (base.py:1562-1591)  =>  everything in un-scaled units, mu_s = mu * s
#pragma unroll
      for (int a = 0; a < MXA; ++a) y[i * MXA + a] = (pc.coll.Dc[i + 1] * lam[a] - gl[a]) / pc.sz[a];
    }
    Colloc<MA, DD>::newton_matrix(pc.coll, X, u, p, pc.dt, mat, F_);
    Colloc<MA, DD>::lu(mat);
    Colloc<MA, DD>::lu_solve_t(mat, y);
#pragma unroll
    for (int i = 0; i < DD; ++i)
#pragma unroll
      for (int a = 0; a < MXA; ++a) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < DD; ++j) s -= pc.coll.A[j * DD + i] * y[j * MXA + a];
        mu[i * MXA + a] = s;
      }
    double* lg = lam_g + b * (int64_t)(N * (DB + MXA)) + k * (DB + MXA);
#pragma unroll
    for (int i = 0; i < DD; ++i)
#pragma unroll
      for (int m = 0; m < MXA; ++m) lg[i * (MXA + NZA) + m] = mu[i * MXA + m] * pc.sz[m];
    if constexpr (NZA > 0) {
      // stationarity in z_i:  dt (df/dz)^T mu_i + (dg/dz)^T nu_i = 0   (mu_i: multiplier of the un-scaled collocation row)
#pragma unroll
      for (int i = 0; i < DD; ++i) {
        Dual<NZA1> xd[MX], ud[MU > 0 ? MU : 1], zd[NZA1], fd[MX], gd[NZA1];
#pragma unroll
        for (int q = 0; q < MX; ++q) xd[q] = Dual<NZA1>(X[i * MXA + q]);
#pragma unroll
        for (int q = 0; q < MU; ++q) ud[q] = Dual<NZA1>(u[q]);
#pragma unroll
        for (int a = 0; a < NZA; ++a) {
          zd[a] = Dual<NZA1>(zc[i * NZA + a]);
          zd[a].d[a] = 1.0;
        }
        M::ode_z(xd, zd, ud, p, fd);
        M::alg(xd, zd, ud, p, gd);
        double G[NZA1 * NZA1], rhs[NZA1];      // G = (dg/dz)^T, rhs = -dt (df/dz)^T mu_i
#pragma unroll
        for (int a = 0; a < NZA; ++a) {
          double acc = 0.0;
#pragma unroll
          for (int m = 0; m < MX; ++m) acc += fd[m].d[a] * mu[i * MXA + m];
          rhs[a] = -pc.dt * acc;
#pragma unroll
          for (int c = 0; c < NZA; ++c) G[a * NZA + c] = gd[c].d[a];
        }
#pragma unroll
        for (int c = 0; c < NZA; ++c) {        // Gaussian elimination with partial pivoting (static indices)
#pragma unroll
          for (int q = c + 1; q < NZA; ++q) {
            if (fabs(G[q * NZA + c]) > fabs(G[c * NZA + c])) {
#pragma unroll
              for (int j = c; j < NZA; ++j) { const double t = G[c * NZA + j]; G[c * NZA + j] = G[q * NZA + j]; G[q * NZA + j] = t; }
              const double t = rhs[c]; rhs[c] = rhs[q]; rhs[q] = t;
            }
          }
#pragma unroll
          for (int q = c + 1; q < NZA; ++q) {
            const double f = G[q * NZA + c] / G[c * NZA + c];
#pragma unroll
            for (int j = c + 1; j < NZA; ++j) G[q * NZA + j] -= f * G[c * NZA + j];
            rhs[q] -= f * rhs[c];
          }
        }
#pragma unroll
        for (int c = NZA - 1; c >= 0; --c) {
          double acc = rhs[c];
#pragma unroll
          for (int j = c + 1; j < NZA; ++j) acc -= G[c * NZA + j] * rhs[j];
          rhs[c] = acc / G[c * NZA + c];
        }
#pragma unroll
        for (int a = 0; a < NZA; ++a) lg[i * (MXA + NZA) + MXA + a] = rhs[a];
      }
    }
#pragma unroll
    for (int m = 0; m < MXA; ++m) lg[DB + m] = lamc[b * (int64_t)(N * MXA) + k * MXA + m];
  }
};

// ---- auxiliary kernel of a user problem: plant step --------------------------------------------
template <class M>
__device__ __forceinline__ void user_plant_step(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ x,
                                                const double* __restrict__ u, const double* __restrict__ par,
                                                int64_t par_stride, double* __restrict__ xn) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  constexpr int NX = M::NX, NU = M::NU, NP = M::NP;
  double xv[NX], uv[NU > 0 ? NU : 1], pv[NP > 0 ? NP : 1], xo[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xv[i] = x[b * NX + i];
#pragma unroll
  for (int i = 0; i < NU; ++i) uv[i] = u[b * NU + i];
#pragma unroll
  for (int i = 0; i < NP; ++i) pv[i] = par[b * par_stride + i];
  model_step<M>(pcg->order, pcg->nsub, xv, uv, pv, pcg->dt, xo);
#pragma unroll
  for (int i = 0; i < NX; ++i) xn[b * NX + i] = xo[i];
}

}  // namespace hilo
