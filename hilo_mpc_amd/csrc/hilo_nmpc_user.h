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
//   * a custom constraint function over the WHOLE decision vector (`set_custom_constraints_function`, optimizer.py:1180-1208,
//     mpc.py:1729-1745) in the stage-additive form the host derives from it (hilo_mpc_amd/custom.py): every row r is carried by an
//     accumulator state q_r behind the shared slacks, q_{r,k+1} = q_{r,k} + sum_j a[r][k][j] psi_j(x_k, u_k) (node values, SCALED
//     variables like every entry of v; outside the integrator like the slacks), q_{r,0} = 0 pinned, and is imposed as a hard row
//     q_{r,N} + sum_j a[r][N][j] psi_j(x_N) in [lb, ub] at the end of the last interval - the same NLP as the reference's dense row.
// Engine state = [model x (MX) | theta (NTH) | shared slacks e (NE) | accumulators q (NQ) | held inputs uh (NH)],
// input = [model u (MU) | u_theta].
#pragma once
#include "hilo_expr.h"
#include "hilo_ocp.h"

namespace hilo {

constexpr int USER_QROW = 1000;   // terminal row table (o_trowx): USER_QROW + r = a row of custom constraint function r

// ---- layout of pc.cost for NmpcUser: plain function of the dimensions so that the host (hilo_jit.hip) fills the block ----
struct UserLayout {
  int mza, o_wz, o_zref, o_wn, o_xrefn, o_wdu, o_hasdu, o_we, o_wet, o_ws, o_idxs, o_wt, o_idxt, o_rowx, o_rows, o_rowe,
      o_trowx, o_trows, o_trowe, o_tsoft, o_wzm, o_nrow, o_ncr, o_ntr, o_rref, o_trref, o_acc, o_end;
  __host__ __device__ constexpr UserLayout(int mx, int mu, int nth, int ne, int nps, int npt, int nq = 0, int npsi = 0, int N = 0)
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
        o_nrow(o_wzm + mza),                       // collocation + constraints: rows per POINT (the engine's rows of a stage are the
                                                   // node's rows followed by those of the d collocation points, mpc.py:1338-1356)
        o_ncr(o_nrow + 1),                         // ... rows per point / terminal rows in the REFERENCE's g (dropped rows included)
        o_ntr(o_ncr + 1),
        o_rref(o_ntr + 1),                         // ... and where row r of a point / terminal row r sits there
        o_trref(o_rref + OCP_MAXNC),
        o_acc(o_trref + OCP_MAXNC),                // [nq][N + 1][npsi] coefficients of the accumulator states (custom constraint rows)
        o_end(o_acc + nq * (N + 1) * npsi) {}
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
  static constexpr int NEXPR = 0, NTEXPR = 0, NPS = 0, NPT = 0, NACC = 0;
  static constexpr bool CON_USES_Z = false;
};

template <class M, class F, class C>
struct NmpcUser {
  static constexpr int MX = M::NX, MU = M::NU, NTH = C::NTH, NE = C::NE;
  static constexpr int MXA = MX + NTH, MUA = MU + NTH, MZA = MXA + MUA;   // augmented model
  static constexpr int NH = C::HOLD ? MUA : 0;
  static constexpr int NQ = C::NQ, NPSI = F::NACC;             // accumulators of custom constraint rows, their stage expressions
  static_assert(NQ == 0 || (NH == 0 && NPSI > 0), "accumulator states: full control horizon, at least one stage expression");
  static constexpr int NX = MXA + NE + NQ + NH, NU = MUA, NZ = NX + NU;
  static constexpr int NXV = MXA, NX0 = MX, NU0 = MU, NC = C::NC;
  static constexpr int NPAR = M::NP + M::NU;
  static constexpr int NSD = C::TV ? (MX + MU + M::NP) : 0;   // per stage [zref_k (model z, scaled) | p_k]; row N: terminal ref
  static constexpr bool FIX_X0 = true, COOP = false, BIG = C::BIG;
  static constexpr int VEC_N = C::N;    // the horizon is part of the compiled problem: Ocp::VEC_LDS may keep the vectors in LDS
  static constexpr int D = C::COLL_D;                          // collocation degree, 0 = explicit Runge-Kutta / discrete map
  static constexpr bool CONT = C::CONT;                        // continuous objective
  static constexpr bool FUSED = true;
  // collocation with nonlinear stage constraints: the reference imposes them at every collocation point as well as at the node
  // (mpc.py:1338-1356, :1700-1725) - rows of the interior of the shooting map, evaluated together with it
  static constexpr bool FUSED_CON = C::COLL_D > 0 && F::NEXPR > 0;
  static constexpr int NZALG = model_nz<M>::value;             // algebraic states of a semi-explicit DAE model (eliminated, see below)
  // collocation with the iterate in the workspace: the converged collocation states and the factors of the Newton matrix of an
  // interval are computed ONCE per derivative evaluation and read by all of its Taylor directions (hilo_colloc.h::prepare)
  // ... and when a group of [collocation unknowns | model directions | 1] lanes fits the wave, the lanes of a group solve the
  // interval's system TOGETHER (hilo_colloc.h::CoopLU: one column per lane, in registers) and the derivatives come from the
  // implicit-function theorem instead of Taylor sweeps through the Newton iteration (coll_pass below):
  //     XC    per interval in the workspace: [X (DNC) | the x, u slots they belong to (NWD)]  - written by the values pass
  //     PREP  per interval: [X (DNC) | dX/dw, one column per model direction (NWD x DNC) | kappa (DNC) | x, u the block belongs to (NWD)]
  // Lanes of a group: one per column of the Newton matrix; the right-hand sides (NWD tangents + the Newton residual) ride as a
  // SECOND column in the first lanes when there are no more of them than matrix columns (CTWO: three intervals of configuration
  // 5's 21 x 21 systems per pass of the wave), else in lanes of their own behind the matrix lanes.
  static constexpr int DNC = C::COLL_D * (M::NX + C::NTH), NWD = M::NX + M::NU + 2 * C::NTH;
  static constexpr bool CTWO = NWD + 1 <= DNC;
  static constexpr int CGS = CTWO ? DNC : DNC + NWD + 1;
#ifdef HILO_USER_NO_COOP_COLL   // developer knob: the per-lane Newton solve and the Taylor sweeps against its factors
  static constexpr bool COOP_COLL = false;
#else
  static constexpr bool COOP_COLL = C::COLL_D > 0 && C::BIG && CGS <= 64 && M::NU + C::NTH <= M::NX + C::NTH;
#endif
  static constexpr int CG = COOP_COLL ? 64 / CGS : 1;          // intervals per pass of the wave
#ifdef HILO_USER_VEC_MAX
  static constexpr int VEC_MAX = HILO_USER_VEC_MAX;            // developer knob: cap on the LDS-resident vectors (Ocp::VEC_LEVEL)
#endif
  static constexpr int XCW = COOP_COLL ? DNC + NWD : 0;
  static constexpr int CSTAGE = COOP_COLL ? CG * DNC * (1 + DNC) : 0;   // LDS staging of coll_pass: the groups' states and factors U
#ifdef HILO_USER_NO_PREP   // developer knob (tools/dbg/c5dae_prep.py): every Taylor direction solves the collocation system itself
  static constexpr int PREP = 0;
#else
  static constexpr int PREP = COOP_COLL ? DNC * (NWD + 2) + NWD : ((C::COLL_D > 0 && C::BIG) ? DNC * (1 + DNC) : 0);
#endif
  static constexpr bool QUAD_COST = false;
  static constexpr UserLayout L = UserLayout(MX, MU, NTH, NE, F::NPS, F::NPT, NQ, NPSI, C::N);
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
      if constexpr (MASKED && C::HAS_WZM) {
        // the pattern of the weights is part of the compiled problem (hilo_nmpc_user.hip hands the row masks to the run-time
        // compiler): the form unrolls over the non-zero entries.  (Tested at run time, the inner test was turned into selects: 1500
        // of the 2900 instructions of a collocation point's sweep for configuration 5, whose ten-by-ten weight has three entries.)
        if (C::WZM[i] != 0u) {
          T s = T(0.0);
#pragma unroll
          for (int j = 0; j < MZA; ++j)
            if ((C::WZM[i] >> j) & 1u) s = s + pc.cost[L.o_wz + i * MZA + j] * z[j];
          acc = acc + z[i] * s;
        }
      } else if constexpr (MASKED) {
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
          xis[s] = acc * rcp_fast(pc.sz[s]);
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
    return dyn_cost_impl<false>(pc, par, sd, k, x, u, xn, (T*)nullptr);
  }
  template <class T, class E>
  __device__ __forceinline__ static T dyn_cost_con(const OcpConst& pc, const double* par, const double* sd, int k, const T* x,
                                                   const T* u, T* xn, T* dv, const E&) {
    return dyn_cost_impl<true>(pc, par, sd, k, x, u, xn, dv);
  }

  template <class PP>
  __device__ __forceinline__ static void prepare(const OcpConst& pc, const double* par, const double* sd, int k, const double* x,
                                                 const double* u, PP prep) {
    const double* p = C::TV ? sd + MX + MU : par;
    double xp[MXA], up[MUA > 0 ? MUA : 1];
#pragma unroll
    for (int i = 0; i < MXA; ++i) xp[i] = x[i] * pc.sz[i];
#pragma unroll
    for (int i = 0; i < MUA; ++i) {
      double ui = u[i];
      if constexpr (NH > 0) { if (k >= pc.Nc) ui = x[MXA + NE + i]; }
      up[i] = ui * pc.sz[NX + i];
    }
    if constexpr (D > 0) Colloc<MA, D>::prepare(pc.coll, xp, up, p, pc.dt, prep);
  }
  template <class PP>
  __device__ __forceinline__ static Jet2 dyn_cost_prep(const OcpConst& pc, const double* par, const double* sd, int k, const Jet2* x,
                                                       const Jet2* u, Jet2* xn, Jet2* dv, PP prep) {
    return dyn_cost_impl<FUSED_CON>(pc, par, sd, k, x, u, xn, dv, prep);
  }

  // values pass of the cooperative transcription: the collocation states come from the workspace (coll_pass<false>)
  template <class XP>
  __device__ __forceinline__ static double dyn_cost_xc(const OcpConst& pc, const double* par, const double* sd, int k, const double* x,
                                                       const double* u, double* xn, double* dv, XP xcp) {
    return dyn_cost_impl<FUSED_CON>(pc, par, sd, k, x, u, xn, dv, xcp);
  }

  // inequality row m of a stage is in force (the engine's row_on for a stage row)
  __device__ __forceinline__ static bool stage_row_live(const OcpConst& pc, int m) {
    return m < pc.nc && (pc.dlb[m] > -INFINITY || pc.dub[m] < INFINITY);
  }

  // ---- the collocation systems of the horizon, CG intervals per pass of the wave ----------------------------------------------
  // Lane c < DNC of a group owns column c = (point j, state a) of the Newton matrix  Mat = I - dt (A (x) I) blockdiag(f_x(X_j))
  // in `col`; right-hand side r (tangent dX/dw_r for r < NWD - w: the model's x, theta | u, u_theta, un-scaled -, the Newton
  // residual for r = NWD) sits in `ecol` of lane r (CTWO) or of lane DNC + r.  Every pass of the iteration: lane (j, a) evaluates
  // the model at ITS point in Dual arithmetic seeded with x_a (and u_a) - one evaluation per lane, the D points side by side -,
  // the right-hand sides collect the values / input derivatives they need from the lanes that hold them, the group eliminates
  // (hilo_colloc.h::CoopLU), substitutes, and the residual's lane updates the states (staged in LDS: `stage`, CG x DNC).
  // DERIV = false (line search, start of the solve): iterate to 1e-8 relative - quadratic
  // convergence puts the update's own error at round-off - and leave [X | slots] in `xc`.  DERIV = true (derivative phase):
  // start from `xc` when it belongs to this point (it does after an accepted trial point: one pass), iterate to 1e-12, keep the
  // tangents of the last pass and add the ADJOINT of the system for the second-order terms: with
  //     Phi(X, w) = sum_i dt B_i l(X_i) - lam^T x+(X) + sum_(i,r) nu_(i,r) c_r(X_i)       (what the engine's q sums up)
  // and y = Mat^-T Phi_X, the second derivative of w -> Phi(X(w), w) along a direction is the second Taylor coefficient of
  // Phi + sum_j kappa_j^T f(X_j, u), kappa_j = dt sum_i A_ij y_i, along the STRAIGHT line (X + t X_w dir, w + t dir): the curvature
  // of X(w) drops out against the stationarity of the Lagrangian in X.  The direction tasks (dyn_cost_impl) therefore sweep model,
  // cost and rows once, with no linear solve, where they swept the Newton iteration twice with two substitutions each.
  // `from_prep` (values pass of a trial point): start from the first-order prediction off the iterate's states and tangents
  template <bool DERIV, class ZP, class LP, class NUP, class XP, class PP>
  __device__ __forceinline__ static void coll_pass(const OcpConst& pc, const double* par, const double* sd0, ZP Z, LP lam, NUP cnu,
                                                   int N, XP xc, PP prep, lds_double* stage, bool from_prep = false) {
    static_assert(COOP_COLL, "cooperative collocation pass");
    const int lane = threadIdx.x;
    const int g0 = lane / CGS;
    const bool ingroup = g0 < CG;
    const int g = ingroup ? g0 : 0;
    const int gbase = g * CGS;
    const int c = ingroup ? lane - gbase : (1 << 20);                // lanes behind the last group: no role
    const bool is_mat = c < DNC;
    const int j = is_mat ? c / MXA : 0, a = is_mat ? c - j * MXA : 0;
    const int r = !ingroup ? -1 : (CTWO ? c : c - DNC);              // right-hand side this lane carries (-1 / > NWD: none)
    const bool is_tan = r >= 0 && r < NWD, is_res = r == NWD;
    const int bsel = (r >= MXA && r < NWD) ? r - MXA : 0;            // the input a tangent lane differentiates in
    lds_double* Xs = stage + g * DNC;
    const double tol = DERIV ? 1e-12 : 1e-8;
    constexpr int ND = DERIV ? 2 : 1;
    for (int k0 = 0; k0 < N; k0 += CG) {
      const int k = k0 + g;
      const bool act = ingroup && k < N;
      const int kk = k < N ? k : N - 1;                              // (a group past the horizon repeats the last interval, unrecorded)
      const double* sd = C::TV ? sd0 + (size_t)kk * NSD : nullptr;
      const double* p = C::TV ? sd + MX + MU : par;
      double xp[MXA], up[MUA > 0 ? MUA : 1];
#pragma unroll
      for (int i = 0; i < MXA; ++i) xp[i] = Z[kk * NZ + i] * pc.sz[i];
#pragma unroll
      for (int i = 0; i < MUA; ++i) {
        double ui = Z[kk * NZ + NX + i];
        if constexpr (NH > 0) { if (kk >= pc.Nc) ui = Z[kk * NZ + MXA + NE + i]; }
        up[i] = ui * pc.sz[NX + i];
      }
      // the slot of the iterate behind model direction c (lanes c < NWD): tag of the states in `xc`
      int slot = 0;
      if (c < MXA) slot = c;
      else if (c < NWD) {
        slot = NX + (c - MXA);
        if constexpr (NH > 0) { if (kk >= pc.Nc) slot = MXA + NE + (c - MXA); }
      }
      const double tagv = c < NWD ? Z[kk * NZ + slot] : 0.0;
      bool warm = false;
      if constexpr (DERIV) {
        const bool miss = c < NWD && !(xc[(size_t)kk * XCW + DNC + c] == tagv);
        const unsigned long long mm = __ballot(miss);
        const unsigned long long gm = CGS >= 64 ? ~0ull : (((1ull << CGS) - 1ull) << gbase);
        warm = (mm & gm) == 0ull;
      }
      double xa = 0.0;                                               // this lane's own state component x_a (matrix lanes)
#pragma unroll
      for (int m = 0; m < MXA; ++m) xa = a == m ? xp[m] : xa;
      __syncthreads();
      if (is_mat) {
        double x0c = xa;
        if constexpr (DERIV) { if (warm) x0c = xc[(size_t)kk * XCW + c]; }
        else if (from_prep) {
          // first-order prediction from the iterate's states and tangents: X(w + dw) = X + X_w dw + O(dw^2)
          x0c = prep[(size_t)kk * PREP + c];
#pragma unroll
          for (int w = 0; w < NWD; ++w) {
            const double wv = w < MXA ? xp[w < MXA ? w : 0] : up[w >= MXA ? w - MXA : 0];
            x0c = fma(prep[(size_t)kk * PREP + DNC + w * DNC + c], wv - prep[(size_t)kk * PREP + DNC * (NWD + 2) + w], x0c);
          }
        }
        Xs[c] = x0c;
      }
      __syncthreads();
      double col[DNC], ecol[DNC];
      for (int it = 0; it < 12; ++it) {
        Dual<ND> fd[MXA];
        {
          Dual<ND> xd[MXA], ud[MUA > 0 ? MUA : 1];
#pragma unroll
          for (int m = 0; m < MXA; ++m) {
            xd[m] = Dual<ND>(Xs[j * MXA + m]);
            xd[m].d[0] = m == a ? 1.0 : 0.0;
          }
#pragma unroll
          for (int b = 0; b < MUA; ++b) {
            ud[b] = Dual<ND>(up[b]);
            if constexpr (DERIV) ud[b].d[1] = b == a ? 1.0 : 0.0;
          }
          MA::ode(xd, ud, p, pc.dt, fd);
        }
        // column (j, a) of Mat; the right-hand sides from the lanes that hold their ingredients: f(X_jj) in lane (jj, 0),
        // df/du_b(X_jj) in lane (jj, b)
        // (accumulated straight into the lane's right-hand side: tangent  Mat X_w = (1 (x) e_w) | dt (A (x) I) f_u,  residual
        // x - X + dt (A (x) I) f(X) = minus the residual of the Runge-Kutta form)
#pragma unroll
        for (int i = 0; i < D; ++i) {
          const double aij = pc.dt * pc.coll.A[i * D + j];
#pragma unroll
          for (int m = 0; m < MXA; ++m) {
            const int q = i * MXA + m;
            col[q] = (q == c ? 1.0 : 0.0) - aij * fd[m].d[0];
            ecol[q] = is_res ? xp[m] - Xs[q] : ((DERIV && m == r) ? 1.0 : 0.0);
          }
        }
#pragma unroll
        for (int jj = 0; jj < D; ++jj)
#pragma unroll
          for (int m = 0; m < MXA; ++m) {
            double fsel = lane_bcast(fd[m].v, gbase + jj * MXA);
            if constexpr (DERIV) {
              const double fu = lane_bcast(fd[m].d[1], gbase + jj * MXA + bsel);
              fsel = is_res ? fsel : ((is_tan && r >= MXA) ? fu : 0.0);
            } else {
              fsel = is_res ? fsel : 0.0;
            }
            fsel *= pc.dt;
#pragma unroll
            for (int i = 0; i < D; ++i) ecol[i * MXA + m] = fma(pc.coll.A[i * D + jj], fsel, ecol[i * MXA + m]);
          }
        if constexpr (!DERIV) {
          // a point whose residual is at round-off needs no further step (the Newton matrix is within O(dt) of the identity:
          // the step would be as small as the residual): the pass that only confirms convergence ends in front of its elimination
          double rmax = 0.0, xmax = 1.0;
          if (is_res) {
#pragma unroll
            for (int q = 0; q < DNC; ++q) {
              rmax = fmax(rmax, fabs(ecol[q]));
              xmax = fmax(xmax, fabs(Xs[q]));
            }
          }
          // (a trial point is started from the first-order prediction off the iterate: late in the solve, where the steps are
          // small, that prediction already IS the solution to round-off and no elimination runs at all)
          if ((it > 0 || from_prep) && !__any((int)(is_res && act && !(rmax <= 1e-12 * xmax)))) break;
        }
        CoopLU<DNC>::template eliminate2<DERIV>(col, ecol, c, gbase);
#ifdef HILO_COLL_BPERM_BACKSUB
        CoopLU<DNC>::back_substitute2(col, ecol, gbase);
        __syncthreads();
#else
        CoopLU<DNC>::back_substitute2_lds(col, ecol, c, stage + CG * DNC + g * DNC * DNC);
#endif
        double dmax = 0.0, scale = 1.0;
        if (is_res) {
#pragma unroll
          for (int q = 0; q < DNC; ++q) {
            const double xq = Xs[q] + ecol[q];
            Xs[q] = xq;
            dmax = fmax(dmax, fabs(ecol[q]));
            scale = fmax(scale, fabs(xq));
          }
        }
        __syncthreads();
        // wave-uniform exit: every group iterates until the last one has converged (further passes of a converged group are
        // round-off); a NaN (singular pivot) also leaves the loop
        if (!__any((int)(is_res && act && dmax > tol * scale))) break;
      }
      if constexpr (!DERIV) {
        if (act && is_mat) xc[(size_t)kk * XCW + c] = Xs[c];
        if (act && c < NWD) xc[(size_t)kk * XCW + DNC + c] = tagv;
      } else {
        // Phi_X, entry (j, a): derivative of the Lagrange term, the continuity weights and the rows at point j in X_j[a]
        double gphi = 0.0;
        {
          Jet2 xu[MXA], uu[MUA > 0 ? MUA : 1], xcs[MXA], usj[MUA > 0 ? MUA : 1];
#pragma unroll
          for (int m = 0; m < MXA; ++m) {
            xu[m] = Jet2(Xs[j * MXA + m], m == a ? 1.0 : 0.0, 0.0);
            xcs[m] = xu[m] * rcp_fast(pc.sz[m]);
          }
#pragma unroll
          for (int b = 0; b < MUA; ++b) {
            uu[b] = Jet2(up[b]);
            usj[b] = Jet2(up[b] * rcp_fast(pc.sz[NX + b]));
          }
          if constexpr (CONT) gphi += (pc.dt * pc.coll.Bq[j + 1]) * lagrange(pc, par, sd, p, kk, xcs, usj).a;
#pragma unroll
          for (int m = 0; m < MXA; ++m) gphi -= (a == m) ? lam[kk * NX + m] * pc.coll.Dc[j + 1] * rcp_fast(pc.sz[m]) : 0.0;
          if constexpr (FUSED_CON) {
            const int nrow = (int)pc.cost[L.o_nrow];
            constexpr int NZ1 = NZALG > 0 ? NZALG : 1;
            Jet2 zc[NZ1], xe[NX], dvl[NC > 0 ? NC : 1];
#pragma unroll
            for (int i = 0; i < NX; ++i) xe[i] = Jet2(0.0);          // (the slacks enter the rows linearly: no X-derivative)
#pragma unroll
            for (int m = 0; m < (NC > 0 ? NC : 1); ++m) dvl[m] = Jet2(0.0);
            if constexpr (NZALG > 0 && F::CON_USES_Z) dae_solve<M>(xu, uu, p, zc);
            rows_at(pc, p, xu, uu, (NZALG > 0 && F::CON_USES_Z) ? zc : (const Jet2*)nullptr, xe, 0, nrow, dvl);
#pragma unroll
            for (int rr = 0; rr < (NC > 0 ? NC : 1); ++rr) {
              if (rr < nrow) {
                const int m = (j + 1) * nrow + rr;
                if (stage_row_live(pc, m)) gphi += cnu[kk * NC + m] * dvl[rr].a;
              }
            }
          }
        }
        const double y = CoopLU<DNC>::solve_transposed(col, is_mat ? gphi : 0.0, c, gbase);
        double kap = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) kap += pc.coll.A[i * D + j] * lane_bcast(y, gbase + i * MXA + a);
        kap *= pc.dt;
        if (act) {
          if (is_mat) {
            prep[(size_t)kk * PREP + c] = Xs[c];
            prep[(size_t)kk * PREP + DNC * (1 + NWD) + c] = kap;
            if (j == 0) prep[(size_t)kk * PREP + DNC * (NWD + 2) + a] = xa;
            if (j == 0 && a < MUA) {
              double ua = 0.0;
#pragma unroll
              for (int b = 0; b < MUA; ++b) ua = a == b ? up[b] : ua;
              prep[(size_t)kk * PREP + DNC * (NWD + 2) + MXA + a] = ua;
            }
          }
          if (is_tan) {
#pragma unroll
            for (int q = 0; q < DNC; ++q) prep[(size_t)kk * PREP + DNC + r * DNC + q] = ecol[q];
          }
        }
      }
    }
    __syncthreads();
  }

  // rows of ONE point (un-scaled xu, uu, algebraic state z or nullptr): d[m0 + r] = sign_r c_{expr_r} - e_{slack_r}, r < nrow
  template <class T>
  __device__ __forceinline__ static void rows_at(const OcpConst& pc, const double* p, const T* xu, const T* uu, const T* z,
                                                 const T* xeng, int m0, int nrow, T* d) {
    if constexpr (F::NEXPR > 0) {
      T ce[F::NEXPR];
      F::con(xu, uu, z, p, ce);
#pragma unroll
      for (int m = 0; m < (NC > 0 ? NC : 1); ++m) {
        const int r = m - m0;
        if (m < NC && r >= 0 && r < nrow) {
          T v = pc.cost[L.o_rows + r] * pick<F::NEXPR>(ce, (int)pc.cost[L.o_rowx + r]);
          if constexpr (NE > 0) {
            const int ei = (int)pc.cost[L.o_rowe + r];
            if (ei >= 0) v = v - pick<NE>(xeng + MXA, ei);
          }
          d[m] = v;
        }
      }
    }
  }

  template <bool WITH_CON, class T, class PP = const double*>
  __device__ __forceinline__ static T dyn_cost_impl(const OcpConst& pc, const double* par, const double* sd, int k, const T* x,
                                                    const T* u, T* xn, T* dv, PP prep = nullptr) {
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
    double kbq = 0.0;   // cooperative collocation: second-order term of the eliminated collocation equations (added to the cost below)
    if constexpr (D > 0) {
      T Xc[D * MXA];
      if constexpr (COOP_COLL) {
        // The collocation states come from the block the cooperative pass left for this interval (coll_pass): `prep` = the
        // workspace row [X | slots] for a values evaluation, the staged block [X | dX/dw | kappa] for a direction of the derivative
        // phase.  A direction moves the states along their tangent (at most two model directions carry a seed: a unit direction
        // or a pair), the shooting map is LINEAR in them, and the curvature of the eliminated equations enters through kappa.
        // ONE point at a time in a loop that stays a loop: unrolled, the compiler hoists the table reads of all points to the top
        // and the task spills (500 scratch instructions per direction pass, 6 MB of scratch traffic per iteration and instance).
        constexpr bool JET = same_type<T, Jet2>::value;
        int w1 = 0, w2 = 0;
        double s1 = 0.0, s2 = 0.0;
        if constexpr (JET) {
#pragma unroll
          for (int w = 0; w < NWD; ++w) {
            const double aw = w < MXA ? xp[w < MXA ? w : 0].a : up[w >= MXA ? w - MXA : 0].a;
            const bool nz = aw != 0.0, first = nz && s1 == 0.0;
            w2 = (nz && !first) ? w : w2;
            s2 = (nz && !first) ? aw : s2;
            w1 = first ? w : w1;
            s1 = first ? aw : s1;
          }
        }
#pragma unroll
        for (int m = 0; m < MXA; ++m) xo[m] = pc.coll.Dc[0] * xp[m];
        const int nrow = WITH_CON ? (int)pc.cost[L.o_nrow] : 0;
        constexpr int NZ1 = NZALG > 0 ? NZALG : 1;
        constexpr bool ZR = NZALG > 0 && F::CON_USES_Z;
        double kb = 0.0;
        T zc[NZ1];
#pragma unroll 1
        for (int jj = 0; jj < D; ++jj) {
          T Xj[MXA];
#pragma unroll
          for (int m = 0; m < MXA; ++m) {
            const int q = jj * MXA + m;
            if constexpr (JET) Xj[m] = Jet2(prep[q], s1 * prep[DNC + w1 * DNC + q] + s2 * prep[DNC + w2 * DNC + q], 0.0);
            else Xj[m] = prep[q];
          }
          const double dj = pc.coll.Dc[jj + 1];
#pragma unroll
          for (int m = 0; m < MXA; ++m) xo[m] = xo[m] + dj * Xj[m];
          if constexpr (JET) {
            T Fj[MXA];
            MA::ode(Xj, up, p, pc.dt, Fj);
#pragma unroll
            for (int m = 0; m < MXA; ++m) kb = fma(prep[DNC * (1 + NWD) + jj * MXA + m], Fj[m].b, kb);
          }
          if constexpr (WITH_CON) {
            if constexpr (ZR) dae_solve<M>(Xj, up, p, zc);
            rows_at(pc, p, Xj, up, ZR ? zc : (const T*)nullptr, x, (jj + 1) * nrow, nrow, dv);
          }
          if constexpr (CONT) {
            T xcs[MXA];
#pragma unroll
            for (int m = 0; m < MXA; ++m) xcs[m] = Xj[m] * rcp_fast(pc.sz[m]);
            lc = lc + (pc.dt * pc.coll.Bq[jj + 1]) * lagrange(pc, par, sd, p, k, xcs, us);
          }
        }
        if constexpr (JET) kbq = kb;
        if constexpr (WITH_CON) {   // rows at the node (the algebraic state they see: below; d = 1 keeps the collocation point's)
          if constexpr (ZR && D > 1) dae_solve<M>(xp, up, p, zc);
          rows_at(pc, p, xp, up, ZR ? zc : (const T*)nullptr, x, 0, nrow, dv);
        }
      } else if constexpr (PREP > 0 && same_type<T, Jet2>::value) {
        if (prep) Colloc<MA, D>::step_prepared(pc.coll, xp, up, p, pc.dt, xo, (CONT || WITH_CON) ? Xc : nullptr, prep);
        else Colloc<MA, D>::step(pc.coll, xp, up, p, pc.dt, xo, (CONT || WITH_CON) ? Xc : nullptr);
      } else
      Colloc<MA, D>::step(pc.coll, xp, up, p, pc.dt, xo, (CONT || WITH_CON) ? Xc : nullptr);
#ifndef HILO_DBG_SKIP_ROWS
      if constexpr (WITH_CON && !COOP_COLL) {
        // rows of the node [0, nrow), then of the collocation points i = 1..d [i nrow, (i + 1) nrow); an expression that names an
        // algebraic state gets z(x_{k,i}, u_k) at a collocation point; at the node the reference passes the interval's whole zp
        // block (mpc.py:1707) - for d = 1 that IS z at the collocation point; for d > 1 CasADi rejects the call, and the row is
        // evaluated with z consistent with the node
        const int nrow = (int)pc.cost[L.o_nrow];
        constexpr int NZ1 = NZALG > 0 ? NZALG : 1;
        T zc[NZ1];
#pragma unroll
        for (int i = 0; i < D; ++i) {
          if constexpr (NZALG > 0 && F::CON_USES_Z) dae_solve<M>(Xc + i * MXA, up, p, zc);
          rows_at(pc, p, Xc + i * MXA, up, (NZALG > 0 && F::CON_USES_Z) ? zc : (const T*)nullptr, x, (i + 1) * nrow, nrow, dv);
        }
        if constexpr (NZALG > 0 && F::CON_USES_Z) {
          if constexpr (D > 1) dae_solve<M>(xp, up, p, zc);
        }
        rows_at(pc, p, xp, up, (NZALG > 0 && F::CON_USES_Z) ? zc : (const T*)nullptr, x, 0, nrow, dv);
      }
#endif
#ifndef HILO_DBG_SKIP_QUAD
      if constexpr (CONT && !COOP_COLL) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
          T xcs[MXA];
#pragma unroll
          for (int m = 0; m < MXA; ++m) xcs[m] = Xc[i * MXA + m] * rcp_fast(pc.sz[m]);
          lc = lc + (pc.dt * pc.coll.Bq[i + 1]) * lagrange(pc, par, sd, p, k, xcs, us);
        }
      }
#endif
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
    if constexpr (COOP_COLL && same_type<T, Jet2>::value) lc = lc + Jet2(0.0, 0.0, kbq);
#pragma unroll
    for (int i = 0; i < MXA; ++i) xn[i] = xo[i] * rcp_fast(pc.sz[i]);
#pragma unroll
    for (int e = 0; e < NE; ++e) xn[MXA + e] = x[MXA + e];            // shared slacks: constant states
    if constexpr (NQ > 0) {                                            // accumulators: + sum_j a[r][k][j] psi_j(x_k, u_k) at the NODE
      T ps[NPSI];
      F::acc(xs, us, p, ps);
#pragma unroll
      for (int r = 0; r < NQ; ++r) {
        T s = x[MXA + NE + r];
#pragma unroll
        for (int j = 0; j < NPSI; ++j) s = s + pc.cost[L.o_acc + (r * (pc.N + 1) + k) * NPSI + j] * ps[j];
        xn[MXA + NE + r] = s;
      }
    }
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
    if constexpr (WITH_CON) term_rows(pc, p, k, xp, up, x, xn, dv);
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
    rows_at(pc, p, xs, us, (const T*)nullptr, x, 0, pc.nc, d);
    term_rows(pc, p, k, xs, us, x, xn, d);
  }
  // terminal rows of the last stage behind the stage rows: engine row pc.nc + r
  template <class T>
  __device__ __forceinline__ static void term_rows(const OcpConst& pc, const double* p, int k, const T* xs, const T* us, const T* x,
                                                   const T* xn, T* d) {
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
          if (m < NC && r >= 0 && r < pc.nc_term && (int)pc.cost[L.o_trowx + r] < USER_QROW) {
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
    if constexpr (NQ > 0) {
      // custom constraint rows: the terminal rows marked USER_QROW + r, sign (q_{r,N} + sum_j a[r][N][j] psi_j(x_N)) - e_cus on the
      // end state of the horizon (scaled variables; an expression that names an input has a zero coefficient here - v holds no input
      // of stage N); soft rows (mpc.py:1733-1739): fun - e_cus <= ub and -(fun + e_cus) <= -lb
      if (k == pc.N - 1) {
        T xe[MXA], ue[MUA > 0 ? MUA : 1], ps[NPSI], val[NQ];
#pragma unroll
        for (int i = 0; i < MXA; ++i) xe[i] = xn[i];
#pragma unroll
        for (int i = 0; i < MUA; ++i) ue[i] = us[i] * rcp_fast(pc.sz[NX + i]);
        F::acc(xe, ue, p, ps);
#pragma unroll
        for (int r = 0; r < NQ; ++r) {
          T sacc = xn[MXA + NE + r];
#pragma unroll
          for (int j = 0; j < NPSI; ++j) {
            // SELECTED, not multiplied: an expression of an input is evaluated here at a value it was never meant for (x / u, log u:
            // inf or NaN), and 0 * NaN would poison the row
            const double cf = pc.cost[L.o_acc + (r * (pc.N + 1) + pc.N) * NPSI + j];
            if (cf != 0.0) sacc = sacc + cf * ps[j];
          }
          val[r] = sacc;
        }
#pragma unroll
        for (int m = 0; m < (NC > 0 ? NC : 1); ++m) {
          const int r = m - pc.nc;
          if (m < NC && r >= 0 && r < pc.nc_term) {
            const int q = (int)pc.cost[L.o_trowx + r] - USER_QROW;
            if (q >= 0) {
              T v = pc.cost[L.o_trows + r] * pick<NQ>(val, q);
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
  }
  // The reference imposes a custom constraint row on the NODE variable x_N (v[x_ind[N]], mpc.py:1734-1742), whose stationarity
  // puts  - nu_r dc_r/dx_N  into the multiplier of the last continuity row; here the row acts on the integrated end state and the
  // engine's multiplier of that defect does not know it.  Entry i of the correction (scaled variables).
  static constexpr bool LAM_FIX = NQ > 0;
  __device__ __forceinline__ static double lam_fix(const OcpConst& pc, const double* par, const double* sd, int i, const double* xN,
                                                   const double* nuN) {
    if constexpr (NQ > 0) {
      const double* p = C::TV ? sd + MX + MU : par;
      Jet2 xe[MXA], ue[MUA > 0 ? MUA : 1], ps[NPSI];
#pragma unroll
      for (int q = 0; q < MXA; ++q) xe[q] = Jet2(xN[q], q == i ? 1.0 : 0.0, 0.0);
#pragma unroll
      for (int q = 0; q < MUA; ++q) ue[q] = Jet2(0.0);      // (expressions of an input carry a zero coefficient at stage N)
      F::acc(xe, ue, p, ps);
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < NQ; ++r) {
        double nu = 0.0;
#pragma unroll
        for (int m = 0; m < (NC > 0 ? NC : 1); ++m) {      // a soft function has two rows: sign times multiplier, summed
          const int t = m - pc.nc;
          if (m < NC && t >= 0 && t < pc.nc_term && (int)pc.cost[L.o_trowx + t] == USER_QROW + r) nu += pc.cost[L.o_trows + t] * nuN[m];
        }
#pragma unroll
        for (int j = 0; j < NPSI; ++j) {
          const double cf = pc.cost[L.o_acc + (r * (pc.N + 1) + pc.N) * NPSI + j];
          if (cf != 0.0) s -= nu * cf * ps[j].a;       // (selected: see term_rows)
        }
      }
      return s;
    } else {
      (void)pc; (void)par; (void)sd; (void)i; (void)xN; (void)nuN;
      return 0.0;
    }
  }
  // ---- output pass of algebraic states under an EXPLICIT Runge-Kutta transcription: one thread per (instance, interval) -------
  // The reference carries one block of algebraic variables per stage and interval, Z_{k,i}, with the rows
  //     alg(k_i, Z_{k,i}, u_k) = 0,  k_i = f(X_i, Z_{k,i}, u_k),  X_i = x_k + h sum_(j<i) a_ij k_j       (modeling.py:1258-1275: the
  // algebraic equations see the stage's SLOPE where the state belongs - restated as it is; the emitted M::alg IS that composition
  // G(X, Z, u) = g(f(X, Z, u), Z, u), codegen.py::dae_model_source) in front of the continuity rows (mpc.py:1647-1670), and node
  // blocks z_0..z_N that enter no row (they keep the guess).  The engine eliminates the Z inside the shooting map; this pass rebuilds
  //     v     = [x | u | z_0..z_N | per interval (Z_{k,1..s})]
  //     lam_g = per interval [algebraic rows, stage by stage | continuity]
  // Multipliers of the algebraic rows from the stationarity of the reference's Lagrangian in Z_{k,j}, backwards over the stages:
  //     kbar_j = -h b_j lambda / s + sum_(m>j) h a_mj Xbar_m          (adjoint of the slope k_j; lambda: the engine's own multiplier)
  //     nu_j   = -G_Z^-T f_z^T kbar_j
  //     Xbar_j = h b_j grad l(X_j) + G_X^T nu_j + f_x^T kbar_j        (continuous objective: the quadrature term of the stage point)
  __device__ static void erk_dae_output(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ vc,
                                        const double* __restrict__ lamc, const double* __restrict__ par, int64_t par_stride,
                                        const double* __restrict__ sdata, int64_t sd_stride, double* __restrict__ v,
                                        double* __restrict__ lam_g) {
    if constexpr (NZALG > 0 && D == 0 && !M::DISCRETE && NTH == 0 && NE == 0 && NQ == 0 && NH == 0 && NC == 0) {
      constexpr int NZA = NZALG, NV = MX + NZA;
      const OcpConst& pc = *pcg;
      const int N = pc.N, s = pc.order;
      const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
      if (e >= batch * N) return;
      const int64_t b = e / N;
      const int k = (int)(e - b * N);
      const int n_head = (N + 1) * MX + N * MU, nv = n_head + (N + 1) * NZA + N * s * NZA;
      const double* row = vc + b * n_head;
      const double* lrow = lamc ? lamc + b * (int64_t)(N * MX) + (int64_t)k * MX : nullptr;
      const double* pr = par + b * par_stride;
      const double* sd = C::TV ? sdata + b * sd_stride + (int64_t)k * NSD : nullptr;
      const double* p = C::TV ? sd + MX + MU : pr;
      double* out = v + b * nv;
      for (int i0 = 0; i0 < n_head; i0 += N) {
        const int i = i0 + k;
        if (i < n_head) out[i] = row[i];
      }
#pragma unroll
      for (int a = 0; a < NZA; ++a) {
        out[n_head + k * NZA + a] = M::z_guess(a);
        if (k == N - 1) out[n_head + N * NZA + a] = M::z_guess(a);
      }
      double A[4][4], bw[4];
      A[0][0] = A[0][1] = A[0][2] = A[0][3] = A[1][1] = A[1][2] = A[1][3] = A[2][2] = A[2][3] = A[3][3] = A[3][0] = A[3][1] = A[2][0] = 0.0;
      A[1][0] = erk_a<1, 0>(s); A[2][0] = erk_a<2, 0>(s); A[2][1] = erk_a<2, 1>(s); A[3][2] = erk_a<3, 2>(s);
      bw[0] = erk_b<0>(s); bw[1] = erk_b<1>(s); bw[2] = erk_b<2>(s); bw[3] = erk_b<3>(s);
      const double h = pc.dt;
      double x[MX], u[MU > 0 ? MU : 1], us[MU > 0 ? MU : 1];
#pragma unroll
      for (int i = 0; i < MX; ++i) x[i] = row[k * MX + i] * pc.sz[i];
#pragma unroll
      for (int i = 0; i < MU; ++i) {
        us[i] = row[(N + 1) * MX + k * MU + i];
        u[i] = us[i] * pc.sz[NX + i];
      }
      double X[4][MX], Z[4][NZA], K[4][MX];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < s) {
#pragma unroll
          for (int a = 0; a < MX; ++a) {
            double acc = x[a];
#pragma unroll
            for (int j = 0; j < i; ++j) acc += h * A[i][j] * K[j][a];
            X[i][a] = acc;
          }
          dae_solve<M>(X[i], u, p, Z[i]);
          M::ode_z(X[i], Z[i], u, p, K[i]);
#pragma unroll
          for (int a = 0; a < NZA; ++a) out[n_head + (N + 1) * NZA + (k * s + i) * NZA + a] = Z[i][a];
        } else {
#pragma unroll
          for (int a = 0; a < MX; ++a) { X[i][a] = x[a]; K[i][a] = 0.0; }
#pragma unroll
          for (int a = 0; a < NZA; ++a) Z[i][a] = M::z_guess(a);
        }
      }
      if (!lam_g) return;
      const int per = s * NZA + MX;
      double* lg = lam_g + b * (int64_t)(N * per) + (int64_t)k * per;
      double lam[MX];
#pragma unroll
      for (int m = 0; m < MX; ++m) lam[m] = lrow[m];
      {   // the engine's own multiplier of the last defect (see coll_output): lambda_ref - grad V(x_N)
        const bool last = k == N - 1 && (pc.flags & 1);
        const double* sdN = C::TV ? sdata + b * sd_stride + (int64_t)N * NSD : nullptr;
#pragma unroll
        for (int m = 0; m < MX; ++m) {
          Jet2 xj[NX];
#pragma unroll
          for (int i = 0; i < NX; ++i) xj[i] = Jet2(row[N * MX + i], i == m ? 1.0 : 0.0, 0.0);
          const double gv = term_cost(pc, pr, sdN, xj).a;
          lam[m] -= last ? gv : 0.0;
        }
      }
      double kbar[4][MX], Xbar[4][MX], nu[4][NZA];
#pragma unroll
      for (int j = 3; j >= 0; --j) {
        if (j < s) {
#pragma unroll
          for (int a = 0; a < MX; ++a) {
            double acc = -h * bw[j] * lam[a] / pc.sz[a];
#pragma unroll
            for (int m = j + 1; m < 4; ++m) acc += (m < s) ? h * A[m][j] * Xbar[m][a] : 0.0;
            kbar[j][a] = acc;
          }
          Dual<NV> xd[MX], zd[NZA], ud[MU > 0 ? MU : 1], fd[MX], gd[NZA];
#pragma unroll
          for (int q = 0; q < MX; ++q) { xd[q] = Dual<NV>(X[j][q]); xd[q].d[q] = 1.0; }
#pragma unroll
          for (int q = 0; q < NZA; ++q) { zd[q] = Dual<NV>(Z[j][q]); zd[q].d[MX + q] = 1.0; }
#pragma unroll
          for (int q = 0; q < MU; ++q) ud[q] = Dual<NV>(u[q]);
          M::ode_z(xd, zd, ud, p, fd);
          M::alg(xd, zd, ud, p, gd);
          double G[NZA * NZA], rhs[NZA];      // G = G_Z^T, rhs = -f_z^T kbar_j
#pragma unroll
          for (int a = 0; a < NZA; ++a) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < MX; ++m) acc += fd[m].d[MX + a] * kbar[j][m];
            rhs[a] = -acc;
#pragma unroll
            for (int c = 0; c < NZA; ++c) G[a * NZA + c] = gd[c].d[MX + a];
          }
#pragma unroll
          for (int c = 0; c < NZA; ++c) {        // Gaussian elimination with partial pivoting (static indices)
#pragma unroll
            for (int q = c + 1; q < NZA; ++q) {
              if (fabs(G[q * NZA + c]) > fabs(G[c * NZA + c])) {
#pragma unroll
                for (int jj = c; jj < NZA; ++jj) { const double t = G[c * NZA + jj]; G[c * NZA + jj] = G[q * NZA + jj]; G[q * NZA + jj] = t; }
                const double t = rhs[c]; rhs[c] = rhs[q]; rhs[q] = t;
              }
            }
#pragma unroll
            for (int q = c + 1; q < NZA; ++q) {
              const double f = G[q * NZA + c] / G[c * NZA + c];
#pragma unroll
              for (int jj = c + 1; jj < NZA; ++jj) G[q * NZA + jj] -= f * G[c * NZA + jj];
              rhs[q] -= f * rhs[c];
            }
          }
#pragma unroll
          for (int c = NZA - 1; c >= 0; --c) {
            double acc = rhs[c];
#pragma unroll
            for (int jj = c + 1; jj < NZA; ++jj) acc -= G[c * NZA + jj] * rhs[jj];
            rhs[c] = acc / G[c * NZA + c];
          }
#pragma unroll
          for (int a = 0; a < NZA; ++a) nu[j][a] = rhs[a];
          double gl[MX];
#pragma unroll
          for (int a = 0; a < MX; ++a) gl[a] = 0.0;
          if constexpr (CONT) {     // gradient of the Lagrange term at the stage point, in scaled variables
            Dual<MX> xs[MX], uq[MU > 0 ? MU : 1];
#pragma unroll
            for (int q = 0; q < MX; ++q) { xs[q] = Dual<MX>(X[j][q] / pc.sz[q]); xs[q].d[q] = 1.0; }
#pragma unroll
            for (int q = 0; q < MU; ++q) uq[q] = Dual<MX>(us[q]);
            const Dual<MX> lv = lagrange<Dual<MX>, false>(pc, pr, sd, p, k, xs, uq);
#pragma unroll
            for (int a = 0; a < MX; ++a) gl[a] = h * bw[j] * lv.d[a] / pc.sz[a];
          }
#pragma unroll
          for (int a = 0; a < MX; ++a) {
            double acc = gl[a];
#pragma unroll
            for (int c = 0; c < NZA; ++c) acc += gd[c].d[a] * nu[j][c];
#pragma unroll
            for (int m = 0; m < MX; ++m) acc += fd[m].d[a] * kbar[j][m];
            Xbar[j][a] = acc;
          }
#pragma unroll
          for (int a = 0; a < NZA; ++a) lg[j * NZA + a] = nu[j][a];
        } else {
#pragma unroll
          for (int a = 0; a < MX; ++a) { kbar[j][a] = 0.0; Xbar[j][a] = 0.0; }
        }
      }
#pragma unroll
      for (int m = 0; m < MX; ++m) lg[s * NZA + m] = lrow[m];
    } else {
      (void)pcg; (void)batch; (void)vc; (void)lamc; (void)par; (void)par_stride; (void)sdata; (void)sd_stride; (void)v; (void)lam_g;
    }
  }

  // ---- collocation output pass: one thread per (instance, interval) ---------------------------------------------------
  // The engine eliminates the collocation states (hilo_colloc.h) and, for a semi-explicit DAE model (codegen.py::
  // dae_model_source), the algebraic states; this pass rebuilds them and the multipliers of their equations, so that `v` and
  // `lam_g` have the reference's layout (mpc.py:1462-1548, :1338-1372, :1657-1725):
  //     v     = [x | u | z_0..z_N (enter no row: the guess) | per interval (ip_k, zp_k) | e]     (the slacks BEHIND the blocks, :1529)
  //     lam_g = per interval [rows at the d collocation points | collocation equations, per point (ode rows, alg rows)
  //                           | continuity | terminal rows (last interval) | rows at the node]
  // The engine's compact results: vc = [x | u | e], lamc = per interval [continuity | rows: node, point 1..d] with the terminal
  // rows of the last interval between its continuity multipliers and its rows (Ocp's write-back with identity row maps).
  // Stationarity of the reference's Lagrangian in a collocation state X_i (z eliminated: total derivatives through zeta):
  //     G_X^T mu = D_i lambda - dt B_i grad l(X_i) - sum_r nu_{i,r} sign_r grad c_r(X_i)
  // solved with the Runge-Kutta form of the equations, mu = -(A^T (x) I) Mat^-T rhs (hilo_colloc.h::multipliers); in z_i:
  //     dt (df/dz)^T mu_i + (dg/dz)^T nu_alg_i + sum_r nu_{i,r} sign_r dc_r/dz = 0.
  __device__ static void coll_output(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ vc,
                                     const double* __restrict__ lamc, const double* __restrict__ par, int64_t par_stride,
                                     const double* __restrict__ sdata, int64_t sd_stride, double* __restrict__ v,
                                     double* __restrict__ lam_g) {
    constexpr int DD = D > 0 ? D : 1, DN = DD * MXA;
    constexpr int NZA = NZALG, DB = DN + DD * NZA, NZA1 = NZA > 0 ? NZA : 1;
    constexpr bool ROWS = FUSED_CON;                       // stage constraints: rows at the collocation points and the node
    constexpr bool ZROWS = ROWS && NZA > 0 && F::CON_USES_Z;
    constexpr int NEX = F::NEXPR > 0 ? F::NEXPR : 1;
    const OcpConst& pc = *pcg;
    const int N = pc.N, Nc = NH > 0 ? pc.Nc : N;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= batch * N) return;
    const int64_t b = e / N;
    const int k = (int)(e - b * N);
    // (rows of vc and v carry the accumulators of custom constraint rows as hidden tail entries behind the slacks, hilo_nmpc_user.hip)
    const int n_head = (N + 1) * MXA + Nc * MUA, n_vc = n_head + NE + NQ, n_zn = (N + 1) * NZA, nv = n_vc + n_zn + N * DB;
    const int ncc = NC > 0 ? pc.nc : 0, ntc = NC > 0 ? pc.nc_term : 0;       // the engine's rows (compact)
    const int nrow = ROWS ? (int)pc.cost[L.o_nrow] : 0;                       // rows per point
    const int R = ROWS ? (int)pc.cost[L.o_ncr] : 0, TR = NC > 0 ? (int)pc.cost[L.o_ntr] : 0;   // ... in the reference's g
    const int cstride = MXA + ncc;
    const double* row = vc + b * n_vc;
    const double* lrow = lamc ? lamc + b * (int64_t)(N * cstride + ntc) + (int64_t)k * cstride : nullptr;
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
    // the [x | u] head and the slacks: copied by the N threads of the instance together (uniform trip count, predicated body -
    // no lane-dependent branch around a loop: DESIGN.md 5.1, tools/check_exec_prologue.py)
    for (int i0 = 0; i0 < n_head; i0 += N) {
      const int i = i0 + k;
      if (i < n_head) out[i] = row[i];
    }
#pragma unroll
    for (int i = 0; i < NE; ++i)
      if (k == 0) out[n_head + n_zn + N * DB + i] = row[n_head + i];
#pragma unroll
    for (int i = 0; i < NQ; ++i)      // q_{r,0} = 0: the next warm start reads the pinned value from here
      if (k == 0) out[n_head + n_zn + N * DB + NE + i] = 0.0;
#pragma unroll
    for (int i = 0; i < DD; ++i)
#pragma unroll
      for (int m = 0; m < MXA; ++m) out[n_head + n_zn + k * DB + i * MXA + m] = X[i * MXA + m] / pc.sz[m];
    double zc[DD * NZA1];
    if constexpr (NZA > 0) {
#pragma unroll
      for (int a = 0; a < NZA; ++a) {
        out[n_head + k * NZA + a] = M::z_guess(a);
        if (k == N - 1) out[n_head + N * NZA + a] = M::z_guess(a);
      }
#pragma unroll
      for (int i = 0; i < DD; ++i) {
        dae_solve<M>(X + i * MXA, u, p, zc + i * NZA);
#pragma unroll
        for (int a = 0; a < NZA; ++a) out[n_head + n_zn + k * DB + DN + i * NZA + a] = zc[i * NZA + a];
      }
    }
    if (!lam_g) return;
    const int per = DD * R + DB + MXA + R;                    // rows of an interval in the reference's g
    double* lg = lam_g + b * (int64_t)(N * per + TR) + (int64_t)k * per;
    double lam[MXA], y[DN], F_[DN], mu[DN];
#pragma unroll
    for (int m = 0; m < MXA; ++m) lam[m] = lrow[m];
    {
      // the engine reports the last defect multiplier in the reference's convention (terminal cost on the integrated end
      // state, mpc.py:1682): lambda_ref = lambda + grad V(x_N); the collocation rows need the engine's own lambda.  Evaluated by
      // every thread and applied by the last interval's (a select, no lane-dependent branch)
      const bool last = k == N - 1 && (pc.flags & 1);
      const double* sdN = C::TV ? sdata + b * sd_stride + (int64_t)N * NSD : nullptr;
#pragma unroll
      for (int m = 0; m < MXA; ++m) {
        Jet2 xj[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xj[i] = Jet2(i < MXA ? row[N * MXA + i] : (i < MXA + NE ? row[n_head + (i - MXA)] : 0.0), i == m ? 1.0 : 0.0, 0.0);
        const double gv = term_cost(pc, pr, sdN, xj).a;
        lam[m] -= last ? gv : 0.0;
      }
      // hard terminal rows act on the integrated end state as well (mpc.py:1693-1700: x_end = sum_j D_j x_{N-1,j}): the collocation
      // rows of the last interval see  - D_i sum_r nu_r sign_r grad c_r(x_end)  next to D_i lambda
      if constexpr (F::NTEXPR > 0) {
        const bool lastc = k == N - 1;
#pragma unroll
        for (int m = 0; m < MXA; ++m) {
          Jet2 xe[MXA], ue[MUA > 0 ? MUA : 1], ct[F::NTEXPR];
#pragma unroll
          for (int i = 0; i < MXA; ++i) xe[i] = Jet2(row[N * MXA + i] * pc.sz[i], i == m ? pc.sz[i] : 0.0, 0.0);
#pragma unroll
          for (int q = 0; q < MUA; ++q) ue[q] = Jet2(u[q]);
          F::tcon(xe, ue, p, ct);
          double acc = 0.0;
          for (int r = 0; r < ntc; ++r)
            acc += lamc[b * (int64_t)(N * cstride + ntc) + (int64_t)(N - 1) * cstride + MXA + r] * pc.cost[L.o_trows + r] *
                   pick<F::NTEXPR>(ct, (int)pc.cost[L.o_trowx + r]).a;
          lam[m] -= lastc ? acc : 0.0;
        }
      }
    }
    // multipliers of the engine's rows of this interval: node rows [0, nrow), point i rows [(i + 1) nrow, (i + 2) nrow)
    const double* nu = lrow + MXA + (k == N - 1 ? ntc : 0);
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
      if constexpr (ROWS) {
        // - sum_r nu_{i,r} sign_r grad c_r(X_i): gradient in un-scaled variables, total derivative through z = zeta(X_i, u).
        // d = 1: the node rows see z of this point too (mpc.py:1707 hands them zp_k) - their z-part joins
        Dual<MXA> xd[MXA], ud[MUA > 0 ? MUA : 1], zd[NZA1], ce[NEX];
#pragma unroll
        for (int q = 0; q < MXA; ++q) {
          xd[q] = Dual<MXA>(X[i * MXA + q]);
          xd[q].d[q] = 1.0;
        }
#pragma unroll
        for (int q = 0; q < MUA; ++q) ud[q] = Dual<MXA>(u[q]);
        if constexpr (ZROWS) dae_solve<M>(xd, ud, p, zd);
        F::con(xd, ud, ZROWS ? zd : (const Dual<MXA>*)nullptr, p, ce);
        for (int r = 0; r < nrow; ++r) {
          const double w = nu[(i + 1) * nrow + r] * pc.cost[L.o_rows + r];
          const Dual<MXA> c = pick<NEX>(ce, (int)pc.cost[L.o_rowx + r]);
#pragma unroll
          for (int a = 0; a < MXA; ++a) y[i * MXA + a] -= w * c.d[a];
        }
        if constexpr (ZROWS && D == 1) {
          Dual<MXA> xk[MXA], cn[NEX];
#pragma unroll
          for (int q = 0; q < MXA; ++q) xk[q] = Dual<MXA>(x[q]);
          F::con(xk, ud, zd, p, cn);
          for (int r = 0; r < nrow; ++r) {
            const double w = nu[r] * pc.cost[L.o_rows + r];
            const Dual<MXA> c = pick<NEX>(cn, (int)pc.cost[L.o_rowx + r]);
#pragma unroll
            for (int a = 0; a < MXA; ++a) y[i * MXA + a] -= w * c.d[a];
          }
        }
      }
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
    // ---- rows at the collocation points, at the node, terminal rows: the engine's multipliers at their places ----
    if constexpr (ROWS) {
      for (int q = 0; q < DD * R; ++q) lg[q] = 0.0;                                   // dropped (unbounded) rows
      double* lnode = lg + DD * R + DB + MXA + (k == N - 1 ? TR : 0);
      for (int q = 0; q < R; ++q) lnode[q] = 0.0;
      for (int r = 0; r < nrow; ++r) {
        const int ref = (int)pc.cost[L.o_rref + r];
        if (ref < 0) continue;        // the hidden rows of bounded algebraic states: bounds of variables in the reference, not rows of g
        lnode[ref] = nu[r];
        for (int i = 0; i < DD; ++i) lg[i * R + ref] = nu[(i + 1) * nrow + r];
      }
    }
    if constexpr (NC > 0) {
      if (k == N - 1 && TR > 0) {
        double* lt = lg + DD * R + DB + MXA;
        for (int q = 0; q < TR; ++q) lt[q] = 0.0;
        for (int r = 0; r < ntc; ++r) lt[(int)pc.cost[L.o_trref + r]] = lrow[MXA + r];
        if constexpr (!ROWS) {   // (no stage rows under collocation without FUSED_CON)
        }
      }
    }
    double* lc = lg + DD * R;                                                         // collocation equations
#pragma unroll
    for (int i = 0; i < DD; ++i)
#pragma unroll
      for (int m = 0; m < MXA; ++m) lc[i * (MXA + NZA) + m] = mu[i * MXA + m] * pc.sz[m];
    if constexpr (NZA > 0) {
      // stationarity in z_i:  dt (df/dz)^T mu_i + (dg/dz)^T nu_i + sum_r nu_{i,r} sign_r dc_r/dz = 0   (mu_i: multiplier of
      // the un-scaled collocation row)
#pragma unroll
      for (int i = 0; i < DD; ++i) {
        Dual<NZA1> xd[MXA], ud[MUA > 0 ? MUA : 1], zd[NZA1], fd[MX], gd[NZA1];
#pragma unroll
        for (int q = 0; q < MXA; ++q) xd[q] = Dual<NZA1>(X[i * MXA + q]);
#pragma unroll
        for (int q = 0; q < MUA; ++q) ud[q] = Dual<NZA1>(u[q]);
#pragma unroll
        for (int a = 0; a < NZA; ++a) {
          zd[a] = Dual<NZA1>(zc[i * NZA + a]);
          zd[a].d[a] = 1.0;
        }
        M::ode_z(xd, zd, ud, p, fd);
        M::alg(xd, zd, ud, p, gd);
        double G[NZA1 * NZA1], rhs[NZA1];      // G = (dg/dz)^T, rhs = -dt (df/dz)^T mu_i - sum_r nu_r sign_r dc_r/dz
#pragma unroll
        for (int a = 0; a < NZA; ++a) {
          double acc = 0.0;
#pragma unroll
          for (int m = 0; m < MX; ++m) acc += fd[m].d[a] * mu[i * MXA + m];
          rhs[a] = -pc.dt * acc;
#pragma unroll
          for (int c = 0; c < NZA; ++c) G[a * NZA + c] = gd[c].d[a];
        }
        if constexpr (ZROWS) {
          Dual<NZA1> cz[NEX];
          F::con(xd, ud, zd, p, cz);
          for (int r = 0; r < nrow; ++r) {
            const double w = nu[(i + 1) * nrow + r] * pc.cost[L.o_rows + r];
            const Dual<NZA1> c = pick<NEX>(cz, (int)pc.cost[L.o_rowx + r]);
#pragma unroll
            for (int a = 0; a < NZA; ++a) rhs[a] -= w * c.d[a];
          }
          if constexpr (D == 1) {   // the node rows evaluated with this point's z
            Dual<NZA1> xk[MXA], cn[NEX];
#pragma unroll
            for (int q = 0; q < MXA; ++q) xk[q] = Dual<NZA1>(x[q]);
            F::con(xk, ud, zd, p, cn);
            for (int r = 0; r < nrow; ++r) {
              const double w = nu[r] * pc.cost[L.o_rows + r];
              const Dual<NZA1> c = pick<NEX>(cn, (int)pc.cost[L.o_rowx + r]);
#pragma unroll
              for (int a = 0; a < NZA; ++a) rhs[a] -= w * c.d[a];
            }
          }
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
        for (int a = 0; a < NZA; ++a) lc[i * (MXA + NZA) + MXA + a] = rhs[a];
      }
    }
#pragma unroll
    for (int m = 0; m < MXA; ++m) lg[DD * R + DB + m] = lrow[m];
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
