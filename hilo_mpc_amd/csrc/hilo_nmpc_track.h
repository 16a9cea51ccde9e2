// Policy of the interior-point engine for tracking NMPC with quadratic costs (QuadraticCost, hilo_mpc/util/modeling.py:243-283).
#pragma once
#include "hilo_ocp.h"

namespace hilo {

// Policy: tracking NMPC with quadratic costs.  pc.cost = [Wz | zref | WN | xrefN | Wdu | has_du | Sz | Sdu],
// par = [model parameters | u_old (scaled)].  Sz = Wz + Wz^T, Sdu = Wdu + Wdu^T: the (constant) Hessians of the two terms, written
// by the host next to the weights when the block has room (SYMTAB) - the derivative phase reads its two columns of them once and
// takes gradient AND Hessian entries from those registers (cost_cols).
// SYM_: take the model derivatives from generated symbolic code when the model has it (ModelSym<M>) - the host selects
// SYM_ = false for sub-stepped integration (n_sub > 1), which only the Taylor path covers.
template <class M, bool BIG_ = false, bool SYM_ = true>
struct NmpcTrack {
  using Model = M;
  static constexpr bool SYM = SYM_ && ModelSym<M>::value && !model_has_ext<M>::value;
  static constexpr int NX = M::NX, NU = M::NU, NZ = NX + NU, NPAR = M::NP + M::NU, NSD = 0;
  static constexpr bool FIX_X0 = true;
  static constexpr bool BIG = BIG_;  // false: iterate in LDS; true: per-instance global workspace (long horizons)
  static constexpr int NC = 0, NXV = NX, NX0 = NX, NU0 = NU;  // no inequality rows; plain [x | u] decision vector
  static constexpr bool COOP = model_has_ext<M>::value;  // learned term in the model: lanes share its kernel sum
  static constexpr bool QUAD_COST = true;  // gradient / Hessian of the stage cost in closed form (cost_grad, cost_hess)
  static constexpr int O_WZ = 0, O_ZREF = O_WZ + NZ * NZ, O_WN = O_ZREF + NZ, O_XREFN = O_WN + NX * NX,
                       O_WDU = O_XREFN + NX, O_HASDU = O_WDU + NU * NU, O_SZ = O_HASDU + 1, O_SDU = O_SZ + NZ * NZ;
  static constexpr bool SYMTAB = O_SDU + NU * NU <= OCP_NCOST;     // (hilo_nmpc.hip fills the tables under the same condition)
  static constexpr int O_END = SYMTAB ? O_SDU + NU * NU : O_SZ;
  static constexpr int NCOST = O_END;  // doubles of pc.cost this policy reads (copied to LDS)

  template <class T, class E>
  __device__ __forceinline__ static void dyn(const OcpConst& pc, const double* par, const double*, int, const T* x,
                                             const T* u, T* xn, const E& ext) {
    T xp[NX], up[NU > 0 ? NU : 1], xo[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xp[i] = x[i] * pc.sz[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) up[i] = u[i] * pc.sz[NX + i];
    model_step<M>(pc.order, pc.nsub, xp, up, par, pc.dt, xo, ext);
#pragma unroll
    for (int i = 0; i < NX; ++i) xn[i] = xo[i] * rcp_fast(pc.sz[i]);
  }

  // x+ = Phi(x, u, p) on UN-scaled quantities: the closed-loop helper plant_step_kernel (hilo_nmpc.hip) as a function - the solve
  // kernel advances the plant itself when the caller asks for it (OcpExtra::x_next)
  static constexpr bool PLANT = !COOP;
  __device__ __forceinline__ static void plant(const OcpConst& pc, const double* par, const double* x, const double* u, double* xn) {
    if constexpr (PLANT) {
      double xv[NX], uv[NU > 0 ? NU : 1], pv[M::NP > 0 ? M::NP : 1], xo[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) xv[i] = x[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) uv[i] = u[i];
#pragma unroll
      for (int i = 0; i < M::NP; ++i) pv[i] = par[i];
      model_step<M>(pc.order, pc.nsub, xv, uv, pv, pc.dt, xo);
#pragma unroll
      for (int i = 0; i < NX; ++i) xn[i] = xo[i];
    }
  }

  template <class T>
  __device__ __forceinline__ static T stage_cost(const OcpConst& pc, const double* par, const double*, int k,
                                                 const T* x, const T* u) {
    T z[NZ];
#pragma unroll
    for (int i = 0; i < NX; ++i) z[i] = x[i] - pc.cost[O_ZREF + i];
#pragma unroll
    for (int i = 0; i < NU; ++i) z[NX + i] = u[i] - pc.cost[O_ZREF + NX + i];
    // the weights of two rows are requested from LDS before the first multiply-add (the empty statement with side effects ends
    // the scheduling region; left alone the compiler reads - waits - uses entry by entry, 70 cycles each at one wave per SIMD)
    T acc = T(0.0);
#pragma unroll
    for (int i0 = 0; i0 < NZ; i0 += 2) {
      double w[2][NZ];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < NZ; ++j) w[a][j] = pc.cost[O_WZ + (i0 + a < NZ ? i0 + a : i0) * NZ + j];
      asm volatile("");
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (i0 + a < NZ) {
          T s = T(0.0);
#pragma unroll
          for (int j = 0; j < NZ; ++j) s = s + w[a][j] * z[j];
          acc = acc + z[i0 + a] * s;
        }
      }
    }
    if (k == 0 && pc.cost[O_HASDU] != 0.0) {  // mpc.py:1631-1635: the change penalty only sees u_old in interval 0
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

  // d/dz_i and d2/dz_i dz_j of (z - zref)^T Wz (z - zref) [+ (u - u_old)^T Wdu (u - u_old) in interval 0]
  __device__ __forceinline__ static double cost_grad(const OcpConst& pc, const double* par, const double*, int k, int i, const double* z) {
    double g = 0.0;
    {
      double wr[NZ], wc[NZ], zr[NZ];
#pragma unroll
      for (int j = 0; j < NZ; ++j) { wr[j] = pc.cost[O_WZ + i * NZ + j]; wc[j] = pc.cost[O_WZ + j * NZ + i]; zr[j] = pc.cost[O_ZREF + j]; }
      asm volatile("");   // (every operand requested before the first use, see stage_cost)
#pragma unroll
      for (int j = 0; j < NZ; ++j) g += (wr[j] + wc[j]) * (z[j] - zr[j]);
    }
    if (k == 0 && i >= NX && pc.cost[O_HASDU] != 0.0) {
#pragma unroll
      for (int j = 0; j < NU; ++j)
        g += (pc.cost[O_WDU + (i - NX) * NU + j] + pc.cost[O_WDU + j * NU + (i - NX)]) * (z[NX + j] - par[M::NP + j]);
    }
    return g;
  }
  __device__ __forceinline__ static double cost_hess(const OcpConst& pc, int k, int i, int j) {
    double h = pc.cost[O_WZ + i * NZ + j] + pc.cost[O_WZ + j * NZ + i];
    if (k == 0 && i >= NX && j >= NX && pc.cost[O_HASDU] != 0.0)
      h += pc.cost[O_WDU + (i - NX) * NU + (j - NX)] + pc.cost[O_WDU + (j - NX) * NU + (i - NX)];
    return h;
  }

  // Gradient entries g[c] and Hessian entries ch[r][c] (r >= column; 0 above the diagonal) of the stage cost for the CPL columns
  // c0, c0 + 1, .. of interval k at the point z, from the symmetrised weights: 2 * CPL * NZ / 2 wide LDS reads and the references,
  // all requested before the first use and without a branch - the entry-by-entry forms above cost 66 reads under conditions per
  // lane and evaluation (4.3 k of the derivative phase's 21.6 k clocks on the benchmark problem).  Same products, same order of
  // the sums as cost_grad / cost_hess.
  template <int CPL>
  __device__ __forceinline__ static void cost_cols(const OcpConst& pc, const double* par, int k, int c0, const double* z,
                                                   double* g, double (*ch)[CPL]) {
    static_assert(SYMTAB, "cost_cols needs the symmetrised tables");
    double S[NZ][CPL], zr[NZ], Sd[NU > 0 ? NU : 1][CPL], uo[NU > 0 ? NU : 1];
    int col[CPL], cu[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      col[c] = c0 + c < NZ ? c0 + c : NZ - 1;
      cu[c] = col[c] >= NX ? col[c] - NX : 0;
    }
#pragma unroll
    for (int r = 0; r < NZ; ++r) {
      zr[r] = pc.cost[O_ZREF + r];
#pragma unroll
      for (int c = 0; c < CPL; ++c) S[r][c] = pc.cost[O_SZ + r * NZ + col[c]];
    }
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      uo[a] = par[M::NP + a];
#pragma unroll
      for (int c = 0; c < CPL; ++c) Sd[a][c] = pc.cost[O_SDU + a * NU + cu[c]];
    }
    const bool du = k == 0 && pc.cost[O_HASDU] != 0.0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < NZ; ++r) acc += S[r][c] * (z[r] - zr[r]);
      const bool duc = du && col[c] >= NX;
      double acd = acc;
#pragma unroll
      for (int a = 0; a < NU; ++a) acd += Sd[a][c] * (z[NX + a] - uo[a]);
      g[c] = duc ? acd : acc;
#pragma unroll
      for (int r = 0; r < NZ; ++r) {
        double h = S[r][c];
        if (r >= NX) h = duc ? h + Sd[r >= NX ? r - NX : 0][c] : h;
        ch[r][c] = (c0 + c < NZ && r >= c0 + c) ? h : 0.0;
      }
    }
  }

  // closed forms of the terminal cost (x - xrefN)^T WN (x - xrefN): gradient entry, (constant) Hessian entry
  __device__ __forceinline__ static double term_grad(const OcpConst& pc, int i, const double* x) {
    double g = 0.0;
#pragma unroll
    for (int j = 0; j < NX; ++j) g += (pc.cost[O_WN + i * NX + j] + pc.cost[O_WN + j * NX + i]) * (x[j] - pc.cost[O_XREFN + j]);
    return g;
  }
  __device__ __forceinline__ static double term_hess(const OcpConst& pc, int i, int j) {
    return pc.cost[O_WN + i * NX + j] + pc.cost[O_WN + j * NX + i];
  }

  template <class T>
  __device__ __forceinline__ static T term_cost(const OcpConst& pc, const double*, const double*, const T* x) {
    T z[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) z[i] = x[i] - pc.cost[O_XREFN + i];
    T acc = T(0.0);
#pragma unroll
    for (int i0 = 0; i0 < NX; i0 += 2) {   // two rows of weights in flight, see stage_cost
      double w[2][NX];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < NX; ++j) w[a][j] = pc.cost[O_WN + (i0 + a < NX ? i0 + a : i0) * NX + j];
      asm volatile("");
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        if (i0 + a < NX) {
          T s = T(0.0);
#pragma unroll
          for (int j = 0; j < NX; ++j) s = s + w[a][j] * z[j];
          acc = acc + z[i0 + a] * s;
        }
      }
    }
    return acc;
  }
};

}  // namespace hilo
