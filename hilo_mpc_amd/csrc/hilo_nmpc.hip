// Batched direct-multiple-shooting NMPC: structure-exploiting primal-dual interior point, one workgroup per
// problem instance, all iterates LDS-resident + C ABI.
//
// Replaces, for a pre-discretised model with `integration_method='discrete'`, the solver object the
// reference builds with `ca.nlpsol('solver','ipopt',{'f','x','p','g'})` (hilo_mpc/modules/controller/mpc.py:1778-1787)
// and calls once per step from `NMPC._optimize` (mpc.py:722).  Transcription restated from mpc.py:1455-1787:
//   v = [x_0..x_N | u_0..u_{N-1}] (scaled),  g_k = x_{k+1} - Phi(x_k,u_k) = 0,  J = sum_k l(x_k,u_k) + V(x_N),
//   x_0 pinned (mpc.py:797-802; removed from the variables like IPOPT's make_parameter), box bounds on x, u.
// Algorithm: the interior-point method IPOPT implements (Waechter & Biegler 2006: monotone mu, fraction to the
// boundary, filter line search, inertia correction) with the KKT system solved by a Riccati recursion over the
// horizon instead of a general sparse LDL^T, exact Lagrangian Hessian (K2) by second-order
// univariate Taylor propagation (Jet2) through the Runge-Kutta shooting map + polarisation.
//
// Parallel mapping (this path is fp64-VALU/latency bound, not HBM bound): per iteration
//   derivatives   N * nz(nz+1)/2 independent (stage, direction) Taylor tasks across the lanes
//   Riccati       sequential over stages, each stage's small dense products spread over lanes, LDS-staged blocks
//   reductions    wave shuffles (norms, step lengths, filter tests); decisions are wave-uniform
// HBM is touched only to read (x0, p, warm start) and to write the result.
#include <math.h>
#include <string.h>

#include "hilo_common.h"
#include "hilo_models.h"

namespace hilo {

constexpr int NMPC_MAXNX = 8, NMPC_MAXNU = 4, NMPC_MAXNZ = NMPC_MAXNX + NMPC_MAXNU;
constexpr int NMPC_FILTER = 16;

struct NmpcConst {
  int N, order, nsub, max_iter, acceptable_iter, has_du;
  double dt;
  double Wz[NMPC_MAXNZ * NMPC_MAXNZ], zref[NMPC_MAXNZ], WN[NMPC_MAXNX * NMPC_MAXNX], xrefN[NMPC_MAXNX];
  double Wdu[NMPC_MAXNU * NMPC_MAXNU];
  double lbz[NMPC_MAXNZ], ubz[NMPC_MAXNZ];  // relaxed bounds of a stage's (x,u) slots (scaled); +-inf if none
  double sz[NMPC_MAXNZ];                    // scaling of (x,u)
  // interior-point constants (IPOPT defaults)
  double tol, acceptable_tol, mu_init, kappa_eps, kappa_mu, theta_mu, tau_min, bound_push, bound_frac, s_max,
      kappa_sigma, gamma_theta, gamma_phi, delta_ls, s_theta, s_phi, eta_phi, theta_min_fact, theta_max_fact,
      delta_w_min, delta_w_0, delta_w_max, kappa_w_minus, kappa_w_plus, kappa_w_plus_bar;
};

// ---- block-wide reductions (result broadcast to every lane) ------------------------------------------------
struct OpSum { __device__ static double id() { return 0.0; } __device__ static double f(double a, double b) { return a + b; } };
struct OpMax { __device__ static double id() { return -INFINITY; } __device__ static double f(double a, double b) { return fmax(a, b); } };
struct OpMin { __device__ static double id() { return INFINITY; } __device__ static double f(double a, double b) { return fmin(a, b); } };

template <class Op>
__device__ __forceinline__ double block_reduce(double v, double* scratch) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = Op::f(v, __shfl_xor(v, o, 64));
  const int nw = blockDim.x >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = Op::id();
  for (int w = 0; w < nw; ++w) r = Op::f(r, scratch[w]);
  return r;
}

template <class M>
struct Nmpc {
  static constexpr int NX = M::NX, NU = M::NU, NZ = NX + NU, NP = M::NP, NDIR = NZ * (NZ + 1) / 2;

  // LDS carve-up (doubles)
  struct Lds {
    double *Z, *Zt, *D, *zL, *zU, *dzL, *dzU, *grad, *lam, *lamn, *c, *ct, *AB, *W, *Qd, *P, *pv, *Kg, *kff, *T1,
        *vv, *Mm, *mm, *Rinv, *filt, *red, *par;
  };
  __host__ __device__ static size_t lds_doubles(int N) {
    const size_t S = (size_t)(N + 1) * NZ;
    return 8 * S + 4 * (size_t)N * NX + (size_t)N * NX * NZ + (size_t)N * NZ * NZ + (size_t)N * NDIR +
           (size_t)(N + 1) * NX * NX + (size_t)(N + 1) * NX + (size_t)N * NU * NX + (size_t)N * NU + NX * NZ + NX +
           NZ * NZ + NZ + NU * NU + 2 * NMPC_FILTER + 16 + (NP > 0 ? NP : 1);
  }
  __device__ static Lds carve(double* base, int N) {
    Lds l;
    const size_t S = (size_t)(N + 1) * NZ;
    double* q = base;
    auto take = [&](size_t n) { double* r = q; q += n; return r; };
    l.Z = take(S); l.Zt = take(S); l.D = take(S); l.zL = take(S); l.zU = take(S); l.dzL = take(S); l.dzU = take(S);
    l.grad = take(S);
    l.lam = take((size_t)N * NX); l.lamn = take((size_t)N * NX); l.c = take((size_t)N * NX); l.ct = take((size_t)N * NX);
    l.AB = take((size_t)N * NX * NZ); l.W = take((size_t)N * NZ * NZ); l.Qd = take((size_t)N * NDIR);
    l.P = take((size_t)(N + 1) * NX * NX); l.pv = take((size_t)(N + 1) * NX);
    l.Kg = take((size_t)N * NU * NX); l.kff = take((size_t)N * NU);
    l.T1 = take(NX * NZ); l.vv = take(NX); l.Mm = take(NZ * NZ); l.mm = take(NZ); l.Rinv = take(NU * NU);
    l.filt = take(2 * NMPC_FILTER); l.red = take(16); l.par = take(NP > 0 ? NP : 1);
    return l;
  }

  // a slot (k, i) of the stage-major primal layout is a free variable unless it is x_0 or u_N
  __device__ static bool is_free(int N, int k, int i) { return !((k == 0 && i < NX) || (k == N && i >= NX)); }

  __device__ static void pair_of(int d, int& i, int& j) {  // d >= NZ -> (i < j)
    int r = d - NZ;
    i = 0;
    while (r >= NZ - 1 - i) { r -= NZ - 1 - i; ++i; }
    j = i + 1 + r;
  }
  __device__ static int dir_of(int i, int j) {  // i < j
    return NZ + i * (NZ - 1) - i * (i - 1) / 2 + (j - i - 1);
  }

  // ---- shooting defects at a trial point (values only): ct_k = x_{k+1} - Phi(x_k,u_k) ----------------------
  __device__ static void eval_defects(const NmpcConst& pc, const Lds& l, const double* Zp, double* cp) {
    const int N = pc.N;
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
      double x[NX], u[NU > 0 ? NU : 1], xn[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = Zp[k * NZ + i] * pc.sz[i];
#pragma unroll
      for (int i = 0; i < NU; ++i) u[i] = Zp[k * NZ + NX + i] * pc.sz[NX + i];
      model_step<M>(pc.order, pc.nsub, x, u, l.par, pc.dt, xn);
#pragma unroll
      for (int i = 0; i < NX; ++i) cp[k * NX + i] = Zp[(k + 1) * NZ + i] - xn[i] / pc.sz[i];
    }
  }

  // objective f = sum_k (z-zref)^T Wz (z-zref) [+ (u_0-u_old)^T Wdu (u_0-u_old)] + (x_N-xref)^T WN (x_N-xref)
  __device__ static double eval_objective(const NmpcConst& pc, const Lds& l, const double* Zp, const double* u_old) {
    const int N = pc.N;
    double part = 0.0;
    for (int e = threadIdx.x; e < (N + 1) * NZ; e += blockDim.x) {
      const int k = e / NZ, i = e - k * NZ;
      if (k < N) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NZ; ++j) s += pc.Wz[i * NZ + j] * (Zp[k * NZ + j] - pc.zref[j]);
        part += (Zp[k * NZ + i] - pc.zref[i]) * s;
        if (k == 0 && i >= NX && pc.has_du && u_old) {
          double t = 0.0;
#pragma unroll
          for (int j = 0; j < NU; ++j) t += pc.Wdu[(i - NX) * NU + j] * (Zp[NX + j] - u_old[j]);
          part += (Zp[i] - u_old[i - NX]) * t;
        }
      } else if (i < NX) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NX; ++j) s += pc.WN[i * NX + j] * (Zp[N * NZ + j] - pc.xrefN[j]);
        part += (Zp[N * NZ + i] - pc.xrefN[i]) * s;
      }
    }
    return block_reduce<OpSum>(part, l.red);
  }

  // -mu * sum log(slacks)
  __device__ static double eval_barrier(const NmpcConst& pc, const Lds& l, const double* Zp, double mu) {
    const int N = pc.N;
    double part = 0.0;
    for (int e = threadIdx.x; e < (N + 1) * NZ; e += blockDim.x) {
      const int k = e / NZ, i = e - k * NZ;
      if (!is_free(N, k, i)) continue;
      if (pc.lbz[i] > -INFINITY) part -= log(Zp[e] - pc.lbz[i]);
      if (pc.ubz[i] < INFINITY) part -= log(pc.ubz[i] - Zp[e]);
    }
    return mu * block_reduce<OpSum>(part, l.red);
  }

  // ---- full derivative evaluation at Z: c, AB, W (Lagrangian Hessian blocks), grad --------------------------
  __device__ static void eval_derivs(const NmpcConst& pc, const Lds& l, const double* u_old) {
    const int N = pc.N;
    for (int task = threadIdx.x; task < N * NDIR; task += blockDim.x) {
      const int k = task / NDIR, d = task - k * NDIR;
      int di = d, dj = -1;
      if (d >= NZ) pair_of(d, di, dj);
      if (k == 0 && (di < NX)) {  // x_0 is fixed: directions touching it are never used
        l.Qd[task] = 0.0;
        if (d < NZ) {
#pragma unroll
          for (int m = 0; m < NX; ++m) l.AB[(k * NX + m) * NZ + d] = 0.0;
        }
        if (d != 0) continue;
      }
      Jet2 x[NX], u[NU > 0 ? NU : 1], xn[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const double on = (k == 0 && d == 0) ? 0.0 : ((i == di || i == dj) ? pc.sz[i] : 0.0);
        x[i] = Jet2(l.Z[k * NZ + i] * pc.sz[i], on, 0.0);
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) {
        const double on = (NX + i == di || NX + i == dj) ? pc.sz[NX + i] : 0.0;
        u[i] = Jet2(l.Z[k * NZ + NX + i] * pc.sz[NX + i], on, 0.0);
      }
      model_step<M>(pc.order, pc.nsub, x, u, l.par, pc.dt, xn);
      double q = 0.0;
#pragma unroll
      for (int m = 0; m < NX; ++m) {
        const double is = 1.0 / pc.sz[m];
        if (d == 0) l.c[k * NX + m] = l.Z[(k + 1) * NZ + m] - xn[m].v * is;
        if (d < NZ && !(k == 0 && d < NX)) l.AB[(k * NX + m) * NZ + d] = xn[m].a * is;
        q -= l.lam[k * NX + m] * xn[m].b * is;
      }
      if (!(k == 0 && di < NX)) l.Qd[task] = q;
    }
    __syncthreads();
    // W_k = 2 Wz - sum_m lam_{k+1,m} d2Phi_m  (+ 2 Wdu on the u block of stage 0), by polarisation
    for (int e = threadIdx.x; e < N * NZ * NZ; e += blockDim.x) {
      const int k = e / (NZ * NZ), r = e - k * NZ * NZ, i = r / NZ, j = r - i * NZ;
      const double* Q = l.Qd + k * NDIR;
      double h;
      if (i == j) h = Q[i];
      else {
        const int a = i < j ? i : j, b = i < j ? j : i;
        h = 0.5 * (Q[dir_of(a, b)] - Q[a] - Q[b]);
      }
      h += 2.0 * pc.Wz[i * NZ + j];
      if (k == 0 && pc.has_du && i >= NX && j >= NX) h += 2.0 * pc.Wdu[(i - NX) * NU + (j - NX)];
      l.W[e] = h;
    }
    // grad of the objective w.r.t. every slot
    for (int e = threadIdx.x; e < (N + 1) * NZ; e += blockDim.x) {
      const int k = e / NZ, i = e - k * NZ;
      double g = 0.0;
      if (k < N) {
#pragma unroll
        for (int j = 0; j < NZ; ++j) g += 2.0 * pc.Wz[i * NZ + j] * (l.Z[k * NZ + j] - pc.zref[j]);
        if (k == 0 && i >= NX && pc.has_du && u_old) {
#pragma unroll
          for (int j = 0; j < NU; ++j) g += 2.0 * pc.Wdu[(i - NX) * NU + j] * (l.Z[NX + j] - u_old[j]);
        }
      } else if (i < NX) {
#pragma unroll
        for (int j = 0; j < NX; ++j) g += 2.0 * pc.WN[i * NX + j] * (l.Z[N * NZ + j] - pc.xrefN[j]);
      }
      l.grad[e] = g;
    }
    __syncthreads();
  }

  // dual residual of slot e with multipliers lam: grad + J^T lam - zL + zU
  __device__ static double dual_res(const NmpcConst& pc, const Lds& l, int e) {
    const int N = pc.N, k = e / NZ, i = e - k * NZ;
    double r = l.grad[e] - l.zL[e] + l.zU[e];
    if (i < NX && k >= 1) r += l.lam[(k - 1) * NX + i];
    if (k < N) {
#pragma unroll
      for (int m = 0; m < NX; ++m) r -= l.AB[(k * NX + m) * NZ + i] * l.lam[k * NX + m];
    }
    return r;
  }

  // scaled optimality error pieces (W&B eq. 5): returns dual/s_d and prim; complementarity separately
  __device__ static void opt_error(const NmpcConst& pc, const Lds& l, double& dual_s, double& prim, double& s_c,
                                   double& dual_raw) {
    const int N = pc.N;
    double dmax = 0.0, lsum = 0.0, zsum = 0.0, pmax = 0.0, nb = 0.0;
    for (int e = threadIdx.x; e < (N + 1) * NZ; e += blockDim.x) {
      const int k = e / NZ, i = e - k * NZ;
      if (!is_free(N, k, i)) continue;
      dmax = fmax(dmax, fabs(dual_res(pc, l, e)));
      zsum += fabs(l.zL[e]) + fabs(l.zU[e]);
      nb += (pc.lbz[i] > -INFINITY ? 1.0 : 0.0) + (pc.ubz[i] < INFINITY ? 1.0 : 0.0);
    }
    for (int e = threadIdx.x; e < N * NX; e += blockDim.x) {
      pmax = fmax(pmax, fabs(l.c[e]));
      lsum += fabs(l.lam[e]);
    }
    dmax = block_reduce<OpMax>(dmax, l.red);
    pmax = block_reduce<OpMax>(pmax, l.red);
    lsum = block_reduce<OpSum>(lsum, l.red);
    zsum = block_reduce<OpSum>(zsum, l.red);
    nb = fmax(1.0, block_reduce<OpSum>(nb, l.red));
    const double s_d = fmax(pc.s_max, (lsum + zsum) / (N * NX + nb)) / pc.s_max;
    s_c = fmax(pc.s_max, zsum / nb) / pc.s_max;
    dual_s = dmax / s_d;
    dual_raw = dmax;
    prim = pmax;
  }

  __device__ static double compl_error(const NmpcConst& pc, const Lds& l, double mu) {
    const int N = pc.N;
    double cm = 0.0;
    for (int e = threadIdx.x; e < (N + 1) * NZ; e += blockDim.x) {
      const int k = e / NZ, i = e - k * NZ;
      if (!is_free(N, k, i)) continue;
      if (pc.lbz[i] > -INFINITY) cm = fmax(cm, fabs((l.Z[e] - pc.lbz[i]) * l.zL[e] - mu));
      if (pc.ubz[i] < INFINITY) cm = fmax(cm, fabs((pc.ubz[i] - l.Z[e]) * l.zU[e] - mu));
    }
    return block_reduce<OpMax>(cm, l.red);
  }

  // ---- Riccati factor + solve of the Newton system; returns false when a reduced pivot is not positive ------
  // Hessian block of stage k: W_k + diag(Sigma_k) + delta I ; rhs r = grad - mu/sl + mu/su (slot-wise)
  // `resto` = feasibility-restoration step: H = I, zero gradient (least-norm d with J d = -c)
  __device__ static bool riccati(const NmpcConst& pc, const Lds& l, double mu, double delta, bool resto = false) {
    const int N = pc.N, t = threadIdx.x, T = blockDim.x;
    // terminal: P_N = 2 WN + Sigma + delta, p_N = r_N
    for (int e = t; e < NX * NX + NX; e += T) {
      if (e < NX * NX) {
        const int i = e / NX, j = e - i * NX;
        double v = resto ? 0.0 : 2.0 * pc.WN[e];
        if (i == j) {
          const int s = N * NZ + i;
          v += resto ? 1.0 : delta;
          if (!resto) {
            if (pc.lbz[i] > -INFINITY) v += l.zL[s] / (l.Z[s] - pc.lbz[i]);
            if (pc.ubz[i] < INFINITY) v += l.zU[s] / (pc.ubz[i] - l.Z[s]);
          }
        }
        l.P[N * NX * NX + e] = v;
      } else {
        const int i = e - NX * NX, s = N * NZ + i;
        double v = resto ? 0.0 : l.grad[s];
        if (!resto) {
          if (pc.lbz[i] > -INFINITY) v -= mu / (l.Z[s] - pc.lbz[i]);
          if (pc.ubz[i] < INFINITY) v += mu / (pc.ubz[i] - l.Z[s]);
        }
        l.pv[N * NX + i] = v;
      }
    }
    __syncthreads();
    bool ok = true;
    for (int k = N - 1; k >= 0; --k) {
      const double* Pn = l.P + (k + 1) * NX * NX;
      const double* pn = l.pv + (k + 1) * NX;
      const double* AB = l.AB + k * NX * NZ;
      // T1 = P_{k+1} [A B]  (NX x NZ),  vv = P_{k+1} b + p_{k+1},  b = -c_k
      for (int e = t; e < NX * NZ + NX; e += T) {
        if (e < NX * NZ) {
          const int i = e / NZ, j = e - i * NZ;
          double s = 0.0;
#pragma unroll
          for (int m = 0; m < NX; ++m) s += Pn[i * NX + m] * AB[m * NZ + j];
          l.T1[e] = s;
        } else {
          const int i = e - NX * NZ;
          double s = pn[i];
#pragma unroll
          for (int m = 0; m < NX; ++m) s -= Pn[i * NX + m] * l.c[k * NX + m];
          l.vv[i] = s;
        }
      }
      __syncthreads();
      // Mm = H_k + [A B]^T T1 ; mm = r_k + [A B]^T vv
      for (int e = t; e < NZ * NZ + NZ; e += T) {
        if (e < NZ * NZ) {
          const int i = e / NZ, j = e - i * NZ;
          double s = resto ? 0.0 : l.W[k * NZ * NZ + e];
#pragma unroll
          for (int m = 0; m < NX; ++m) s += AB[m * NZ + i] * l.T1[m * NZ + j];
          if (i == j) {
            const int sl = k * NZ + i;
            s += resto ? 1.0 : delta;
            if (!resto && is_free(N, k, i)) {
              if (pc.lbz[i] > -INFINITY) s += l.zL[sl] / (l.Z[sl] - pc.lbz[i]);
              if (pc.ubz[i] < INFINITY) s += l.zU[sl] / (pc.ubz[i] - l.Z[sl]);
            }
          }
          l.Mm[e] = s;
        } else {
          const int i = e - NZ * NZ, sl = k * NZ + i;
          double s = resto ? 0.0 : l.grad[sl];
          if (!resto && is_free(N, k, i)) {
            if (pc.lbz[i] > -INFINITY) s -= mu / (l.Z[sl] - pc.lbz[i]);
            if (pc.ubz[i] < INFINITY) s += mu / (pc.ubz[i] - l.Z[sl]);
          }
#pragma unroll
          for (int m = 0; m < NX; ++m) s += AB[m * NZ + i] * l.vv[m];
          l.mm[i] = s;
        }
      }
      __syncthreads();
      // Rinv = (M_uu)^-1 via Cholesky (every lane redundantly; tiny) + positivity test
      if constexpr (NU > 0) {
        double Lc[NU * NU];
        bool pd = true;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
          double s = l.Mm[(NX + j) * NZ + NX + j];
#pragma unroll
          for (int q = 0; q < j; ++q) s -= Lc[j * NU + q] * Lc[j * NU + q];
          if (!(s > 0.0)) { pd = false; s = 1.0; }
          const double dd = sqrt(s);
          Lc[j * NU + j] = dd;
#pragma unroll
          for (int i = j + 1; i < NU; ++i) {
            double v = l.Mm[(NX + i) * NZ + NX + j];
#pragma unroll
            for (int q = 0; q < j; ++q) v -= Lc[i * NU + q] * Lc[j * NU + q];
            Lc[i * NU + j] = v / dd;
          }
        }
        ok = ok && pd;
        // K = -Minv_uu M_ux (NU x NX), kff = -Minv_uu m_u : one lane per column of [M_ux | m_u]
        for (int col = t; col < NX + 1; col += T) {
          double y[NU];
#pragma unroll
          for (int a = 0; a < NU; ++a) {
            double s = col < NX ? l.Mm[(NX + a) * NZ + col] : l.mm[NX + a];
#pragma unroll
            for (int q = 0; q < a; ++q) s -= Lc[a * NU + q] * y[q];
            y[a] = s / Lc[a * NU + a];
          }
#pragma unroll
          for (int a = NU - 1; a >= 0; --a) {
            double s = y[a];
#pragma unroll
            for (int q = a + 1; q < NU; ++q) s -= Lc[q * NU + a] * y[q];
            y[a] = s / Lc[a * NU + a];
          }
#pragma unroll
          for (int a = 0; a < NU; ++a) {
            if (col < NX) l.Kg[(k * NU + a) * NX + col] = -y[a];
            else l.kff[k * NU + a] = -y[a];
          }
        }
        __syncthreads();
      }
      // P_k = M_xx + M_xu K (symmetrised), p_k = m_x + M_xu kff
      for (int e = t; e < NX * NX + NX; e += T) {
        if (e < NX * NX) {
          const int i = e / NX, j = e - i * NX;
          double s = 0.5 * (l.Mm[i * NZ + j] + l.Mm[j * NZ + i]);
#pragma unroll
          for (int a = 0; a < NU; ++a)
            s += 0.5 * (l.Mm[i * NZ + NX + a] * l.Kg[(k * NU + a) * NX + j] + l.Mm[j * NZ + NX + a] * l.Kg[(k * NU + a) * NX + i]);
          l.P[k * NX * NX + e] = s;
        } else {
          const int i = e - NX * NX;
          double s = l.mm[i];
#pragma unroll
          for (int a = 0; a < NU; ++a) s += l.Mm[i * NZ + NX + a] * l.kff[k * NU + a];
          l.pv[k * NX + i] = s;
        }
      }
      __syncthreads();
    }
    if (!ok) return false;
    // forward sweep: dx_0 = 0; du_k = K dx_k + kff; dx_{k+1} = A dx_k + B du_k - c_k
    for (int i = t; i < NX; i += T) l.D[i] = 0.0;
    __syncthreads();
    for (int k = 0; k < N; ++k) {
      for (int a = t; a < NU; a += T) {
        double s = l.kff[k * NU + a];
#pragma unroll
        for (int j = 0; j < NX; ++j) s += l.Kg[(k * NU + a) * NX + j] * l.D[k * NZ + j];
        l.D[k * NZ + NX + a] = s;
      }
      __syncthreads();
      for (int i = t; i < NX; i += T) {
        double s = -l.c[k * NX + i];
#pragma unroll
        for (int j = 0; j < NZ; ++j) s += l.AB[(k * NX + i) * NZ + j] * l.D[k * NZ + j];
        l.D[(k + 1) * NZ + i] = s;
      }
      __syncthreads();
    }
    for (int a = t; a < NU; a += T) l.D[N * NZ + NX + a] = 0.0;
    // new equality multipliers: lam_{k+1} = -(P_{k+1} dx_{k+1} + p_{k+1})
    for (int e = t; e < N * NX; e += T) {
      const int k = e / NX, i = e - k * NX;
      double s = l.pv[(k + 1) * NX + i];
#pragma unroll
      for (int j = 0; j < NX; ++j) s += l.P[(k + 1) * NX * NX + i * NX + j] * l.D[(k + 1) * NZ + j];
      l.lamn[e] = -s;
    }
    __syncthreads();
    return true;
  }

  // ---- feasibility restoration, simplified from W&B sec. 3.3 (same statement as oracle/nmpc.py::_restore):
  // least-norm Newton steps on c(w) = 0 with the fraction-to-the-boundary rule and an Armijo search on
  // theta = |c|_1 until theta <= 0.9 theta_start and the point is acceptable to the filter.
  __device__ static bool restore(const NmpcConst& pc, const Lds& l, const double* uo, double mu, double tau,
                                 int nfilt, double theta_max) {
    const int N = pc.N, t = threadIdx.x, T = blockDim.x, SL = (N + 1) * NZ;
    double th = 0.0;
    for (int e = t; e < N * NX; e += T) th += fabs(l.c[e]);
    th = block_reduce<OpSum>(th, l.red);
    const double th_start = th;
    for (int it = 0; it < 50; ++it) {
      riccati(pc, l, mu, 0.0, true);
      double a = 1.0;
      for (int e = t; e < SL; e += T) {
        const int k = e / NZ, i = e - k * NZ;
        if (!is_free(N, k, i)) continue;
        const double d = l.D[e];
        if (pc.lbz[i] > -INFINITY && d < 0.0) a = fmin(a, -tau * (l.Z[e] - pc.lbz[i]) / d);
        if (pc.ubz[i] < INFINITY && d > 0.0) a = fmin(a, tau * (pc.ubz[i] - l.Z[e]) / d);
      }
      double alpha = block_reduce<OpMin>(a, l.red);
      bool ok = false;
      double tht = 0.0;
      while (alpha > 1e-10) {
        for (int e = t; e < SL; e += T) l.Zt[e] = l.Z[e] + alpha * l.D[e];
        __syncthreads();
        eval_defects(pc, l, l.Zt, l.ct);
        __syncthreads();
        tht = 0.0;
        for (int e = t; e < N * NX; e += T) tht += fabs(l.ct[e]);
        tht = block_reduce<OpSum>(tht, l.red);
        if (isfinite(tht) && tht <= (1.0 - 1e-4 * alpha) * th) { ok = true; break; }
        alpha *= 0.5;
      }
      if (!ok) return false;
      for (int e = t; e < SL; e += T) l.Z[e] = l.Zt[e];
      __syncthreads();
      th = tht;
      if (th <= 0.9 * th_start && th <= theta_max) {
        const double ph = eval_objective(pc, l, l.Z, uo) + eval_barrier(pc, l, l.Z, mu);
        bool acc = true;
        for (int q = 0; q < nfilt; ++q)
          if (th >= l.filt[2 * q] && ph >= l.filt[2 * q + 1]) { acc = false; break; }
        if (acc) return true;
      }
      eval_derivs(pc, l, uo);
    }
    return false;
  }
};

template <class M>
__global__ __launch_bounds__(64) void nmpc_solve_kernel(const NmpcConst* __restrict__ pcg, int64_t batch,
                                                        const double* __restrict__ x0, const double* __restrict__ par,
                                                        int64_t par_stride, const double* __restrict__ v0,
                                                        int64_t v0_stride, const double* __restrict__ u_old,
                                                        double* __restrict__ v_opt, double* __restrict__ f_opt,
                                                        double* __restrict__ lam_g, double* __restrict__ u0,
                                                        int32_t* __restrict__ status, int32_t* __restrict__ iters,
                                                        double* __restrict__ kkt) {
  using S = Nmpc<M>;
  constexpr int NX = S::NX, NU = S::NU, NZ = S::NZ, NP = S::NP;
  extern __shared__ double lds_raw[];
  const NmpcConst& pc = *pcg;
  const int N = pc.N, t = threadIdx.x, T = blockDim.x;
  const int64_t b = blockIdx.x;
  if (b >= batch) return;
  typename S::Lds l = S::carve(lds_raw, N);
  const int SL = (N + 1) * NZ;
  const double* uo = (u_old && pc.has_du) ? u_old + b * NU : nullptr;

  // ---- load: x_0 pinned (mpc.py:801-802), warm start in the reference layout [x_0..x_N | u_0..u_{N-1}] ----
  for (int i = t; i < NP; i += T) l.par[i] = par[b * par_stride + i];
  const double* vb = v0 + b * v0_stride;
  for (int e = t; e < SL; e += T) {
    const int k = e / NZ, i = e - k * NZ;
    double v;
    if (i < NX) v = (k == 0) ? x0[b * NX + i] / pc.sz[i] : vb[k * NX + i];
    else v = (k < N) ? vb[(N + 1) * NX + k * NU + (i - NX)] : 0.0;
    if (S::is_free(N, k, i)) {  // IPOPT start: push into the interior (W&B sec. 3.6)
      const double lb = pc.lbz[i], ub = pc.ubz[i];
      const bool hl = lb > -INFINITY, hu = ub < INFINITY;
      if (hl) {
        double pl = pc.bound_push * fmax(1.0, fabs(lb));
        if (hu) pl = fmin(pl, pc.bound_frac * (ub - lb));
        v = fmax(v, lb + pl);
      }
      if (hu) {
        double pu = pc.bound_push * fmax(1.0, fabs(ub));
        if (hl) pu = fmin(pu, pc.bound_frac * (ub - lb));
        v = fmin(v, ub - pu);
      }
      l.zL[e] = hl ? 1.0 : 0.0;
      l.zU[e] = hu ? 1.0 : 0.0;
    } else {
      l.zL[e] = 0.0;
      l.zU[e] = 0.0;
    }
    l.Z[e] = v;
    l.D[e] = 0.0;
  }
  for (int e = t; e < N * NX; e += T) l.lam[e] = 0.0;
  __syncthreads();

  double mu = pc.mu_init, tau = fmax(pc.tau_min, 1.0 - mu);
  double delta_last = 0.0;
  int nfilt = 0, acc_count = 0, it = 0, st = 0;
  double theta_min = 0.0, theta_max = INFINITY;
  double E0 = INFINITY, fval = 0.0;

  for (it = 0;; ++it) {
    S::eval_derivs(pc, l, uo);
    fval = S::eval_objective(pc, l, l.Z, uo);
    double th0 = 0.0;
    for (int e = t; e < N * NX; e += T) th0 += fabs(l.c[e]);
    th0 = block_reduce<OpSum>(th0, l.red);
    if (it == 0) {
      theta_min = pc.theta_min_fact * fmax(1.0, th0);
      theta_max = pc.theta_max_fact * fmax(1.0, th0);
    }
    double dual_s, prim, s_c, dual_raw;
    S::opt_error(pc, l, dual_s, prim, s_c, dual_raw);
    const double c0 = S::compl_error(pc, l, 0.0);
    E0 = fmax(fmax(dual_s, prim), c0 / s_c);
    if (E0 <= pc.tol) { st = HILO_STATUS_SOLVED; break; }
    if (E0 <= pc.acceptable_tol) {
      if (++acc_count >= pc.acceptable_iter) { st = HILO_STATUS_ACCEPTABLE; break; }
    } else acc_count = 0;
    if (it >= pc.max_iter) { st = HILO_STATUS_MAXITER; break; }
    // ---- barrier update (W&B eq. 7) ----
    for (int r = 0; r < 20; ++r) {
      const double Emu = fmax(fmax(dual_s, prim), S::compl_error(pc, l, mu) / s_c);
      if (!(Emu <= pc.kappa_eps * mu && mu > pc.tol / 10 * (1 + 1e-12))) break;
      mu = fmax(pc.tol / 10, fmin(pc.kappa_mu * mu, pow(mu, pc.theta_mu)));
      tau = fmax(pc.tau_min, 1.0 - mu);
      nfilt = 0;
    }
    // ---- search direction with inertia correction (W&B Alg. IC) ----
    double delta = 0.0;
    bool first = true, solved = false;
    for (;;) {
      if (S::riccati(pc, l, mu, delta)) { solved = true; break; }
      if (first) {
        delta = delta_last == 0.0 ? pc.delta_w_0 : fmax(pc.delta_w_min, pc.kappa_w_minus * delta_last);
        first = false;
      } else {
        delta *= delta_last == 0.0 ? pc.kappa_w_plus_bar : pc.kappa_w_plus;
      }
      if (delta > pc.delta_w_max) break;
    }
    if (!solved) { st = HILO_STATUS_RESTORATION_FAILED; break; }
    if (delta > 0.0) delta_last = delta;
    // ---- bound-multiplier steps, fraction to the boundary (W&B eq. 8), directional derivative ----
    double a_p = 1.0, a_z = 1.0, dphi = 0.0;
    for (int e = t; e < SL; e += T) {
      const int k = e / NZ, i = e - k * NZ;
      double dl = 0.0, du = 0.0;
      if (S::is_free(N, k, i)) {
        const double d = l.D[e];
        double gphi = l.grad[e];
        if (pc.lbz[i] > -INFINITY) {
          const double s = l.Z[e] - pc.lbz[i];
          dl = mu / s - l.zL[e] - l.zL[e] / s * d;
          if (d < 0.0) a_p = fmin(a_p, -tau * s / d);
          if (dl < 0.0) a_z = fmin(a_z, -tau * l.zL[e] / dl);
          gphi -= mu / s;
        }
        if (pc.ubz[i] < INFINITY) {
          const double s = pc.ubz[i] - l.Z[e];
          du = mu / s - l.zU[e] + l.zU[e] / s * d;
          if (d > 0.0) a_p = fmin(a_p, tau * s / d);
          if (du < 0.0) a_z = fmin(a_z, -tau * l.zU[e] / du);
          gphi += mu / s;
        }
        dphi += gphi * d;
      }
      l.dzL[e] = dl;
      l.dzU[e] = du;
    }
    a_p = block_reduce<OpMin>(a_p, l.red);
    a_z = block_reduce<OpMin>(a_z, l.red);
    dphi = block_reduce<OpSum>(dphi, l.red);
    // ---- filter line search (W&B Alg. A without second-order correction / restoration phase) ----
    const double phi0 = fval + S::eval_barrier(pc, l, l.Z, mu);
    double alpha = a_p;
    bool accepted = false, armijo = false;
    for (int ls = 0; ls < 60; ++ls) {
      for (int e = t; e < SL; e += T) l.Zt[e] = l.Z[e] + alpha * l.D[e];
      __syncthreads();
      S::eval_defects(pc, l, l.Zt, l.ct);
      __syncthreads();
      double tht = 0.0;
      for (int e = t; e < N * NX; e += T) tht += fabs(l.ct[e]);
      tht = block_reduce<OpSum>(tht, l.red);
      const double pht = S::eval_objective(pc, l, l.Zt, uo) + S::eval_barrier(pc, l, l.Zt, mu);
      bool ok = isfinite(pht) && isfinite(tht) && tht <= theta_max;
      if (ok) {
        for (int q = 0; q < nfilt; ++q) {
          const double tf = l.filt[2 * q], pf = l.filt[2 * q + 1];
          if (tht >= tf && pht - 10 * 2.220446049250313e-16 * fabs(pf) >= pf) { ok = false; break; }
        }
      }
      bool sw = false;
      if (ok) {
        sw = th0 <= theta_min && dphi < 0.0 && alpha * pow(-dphi, pc.s_phi) > pc.delta_ls * pow(th0, pc.s_theta);
        const double rnd = 10 * 2.220446049250313e-16 * fabs(phi0);
        if (sw) ok = pht - phi0 - rnd <= pc.eta_phi * alpha * dphi;
        else ok = tht <= (1 - pc.gamma_theta) * th0 || pht - phi0 - rnd <= -pc.gamma_phi * th0;
      }
      if (ok) { accepted = true; armijo = sw; break; }
      alpha *= 0.5;
      // W&B eq. 23: below alpha_min the line search gives up and the restoration phase is called
      double amin = pc.gamma_theta;
      if (dphi < 0.0) {
        amin = fmin(amin, pc.gamma_phi * th0 / (-dphi));
        if (th0 <= theta_min) amin = fmin(amin, pc.delta_ls * pow(th0, pc.s_theta) / pow(-dphi, pc.s_phi));
      }
      if (alpha < 0.05 * amin) break;
    }
    const bool do_resto = !accepted;
    if (!armijo || do_resto) {  // augment the filter (W&B eq. 22); also done before entering restoration
      if (nfilt == NMPC_FILTER) {
        for (int q = t; q < 2 * (NMPC_FILTER - 1); q += T) l.filt[q] = l.filt[q + 2];
        nfilt = NMPC_FILTER - 1;
        __syncthreads();
      }
      if (t == 0) {
        l.filt[2 * nfilt] = (1 - pc.gamma_theta) * th0;
        l.filt[2 * nfilt + 1] = phi0 - pc.gamma_phi * th0;
      }
      ++nfilt;
      __syncthreads();
    }
    if (do_resto) {
      if (!S::restore(pc, l, uo, mu, tau, nfilt, theta_max)) { st = HILO_STATUS_RESTORATION_FAILED; break; }
      // IPOPT after restoration: equality multipliers reset (constr_mult_reset_threshold = 0), bound multipliers
      // reset to 1 when they exceed bound_mult_reset_threshold = 1000
      double zm = 0.0;
      for (int e = t; e < SL; e += T) zm = fmax(zm, fmax(l.zL[e], l.zU[e]));
      zm = block_reduce<OpMax>(zm, l.red);
      for (int e = t; e < SL; e += T) {
        const int k = e / NZ, i = e - k * NZ;
        if (zm > 1e3 && S::is_free(N, k, i)) {
          l.zL[e] = pc.lbz[i] > -INFINITY ? 1.0 : 0.0;
          l.zU[e] = pc.ubz[i] < INFINITY ? 1.0 : 0.0;
        }
      }
      for (int e = t; e < N * NX; e += T) l.lam[e] = 0.0;
      __syncthreads();
      continue;
    }
    // ---- accept: primal, equality multipliers, bound multipliers (+ W&B eq. 16 safeguard) ----
    for (int e = t; e < SL; e += T) {
      const int k = e / NZ, i = e - k * NZ;
      const double znew = l.Zt[e];
      l.Z[e] = znew;
      if (S::is_free(N, k, i)) {
        if (pc.lbz[i] > -INFINITY) {
          const double s = znew - pc.lbz[i];
          l.zL[e] = fmin(fmax(l.zL[e] + a_z * l.dzL[e], mu / (pc.kappa_sigma * s)), pc.kappa_sigma * mu / s);
        }
        if (pc.ubz[i] < INFINITY) {
          const double s = pc.ubz[i] - znew;
          l.zU[e] = fmin(fmax(l.zU[e] + a_z * l.dzU[e], mu / (pc.kappa_sigma * s)), pc.kappa_sigma * mu / s);
        }
      }
    }
    for (int e = t; e < N * NX; e += T) l.lam[e] += alpha * (l.lamn[e] - l.lam[e]);
    __syncthreads();
  }

  // ---- write back in the reference's layout (mpc.py:1462-1485), u_0 un-scaled (mpc.py:856) ----
  double* vo = v_opt + b * (int64_t)((N + 1) * NX + N * NU);
  for (int e = t; e < SL; e += T) {
    const int k = e / NZ, i = e - k * NZ;
    if (i < NX) vo[k * NX + i] = l.Z[e];
    else if (k < N) vo[(N + 1) * NX + k * NU + (i - NX)] = l.Z[e];
  }
  if (lam_g) {
    // the reference applies the terminal cost to Phi_{N-1} (mpc.py:1682), this solver to x_N; on the feasible
    // set the problems coincide and the multipliers of the last defect differ by grad V(x_N)
    for (int e = t; e < N * NX; e += T) {
      double v = l.lam[e];
      if (e >= (N - 1) * NX) v += l.grad[N * NZ + (e - (N - 1) * NX)];
      lam_g[b * (int64_t)(N * NX) + e] = v;
    }
  }
  for (int a = t; a < NU; a += T) u0[b * NU + a] = l.Z[NX + a] * pc.sz[NX + a];
  if (t == 0) {
    f_opt[b] = fval;
    status[b] = st;
    iters[b] = it;
    if (kkt) kkt[b] = E0;
  }
}

// ---- closed-loop helper for benchmarks / tests: x+ = Phi(x, u) for the true plant = the prediction model ----
template <class M>
__global__ void plant_step_kernel(const NmpcConst* __restrict__ pcg, int64_t batch, const double* __restrict__ x,
                                  const double* __restrict__ u, const double* __restrict__ par, int64_t par_stride,
                                  double* __restrict__ xn) {
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

using namespace hilo;

struct hilo_nmpc {
  int device, model_id, nx, nu, np, N, n_v, n_g;
  NmpcConst host;
  NmpcConst* dev;
  double* v_guess;   // [n_v] device
  double* v_warm;    // [warm_batch][n_v] device: previous solution (mpc.py:725-726)
  int64_t warm_batch;
  int warm_valid;
  size_t lds_bytes;
};

#define HILO_NMPC_MODELS(X)              \
  X(HILO_MODEL_CHEMOSTAT4, Chemostat4)   \
  X(HILO_MODEL_PENDULUM4, Pendulum4)     \
  X(HILO_MODEL_BIOREACTOR3, Bioreactor3)

static int nmpc_model_dims(int id, int* nx, int* nu, int* np, size_t* lds, int N) {
  switch (id) {
#define X(ID, T) case ID: *nx = T::NX; *nu = T::NU; *np = T::NP; *lds = Nmpc<T>::lds_doubles(N) * sizeof(double); return HILO_OK;
    HILO_NMPC_MODELS(X)
#undef X
  }
  return fail(HILO_ENOTSUP, "model id %d has no NMPC instantiation in this build", id);
}

extern "C" void hilo_nmpc_destroy(hilo_nmpc* h) {
  if (!h) return;
  if (h->dev) (void)hipFree(h->dev);
  if (h->v_guess) (void)hipFree(h->v_guess);
  if (h->v_warm) (void)hipFree(h->v_warm);
  delete h;
}

static void copy_or(double* dst, const double* src, int n, double dflt) {
  for (int i = 0; i < n; ++i) dst[i] = src ? src[i] : dflt;
}

extern "C" int hilo_nmpc_create(const hilo_nmpc_desc* d, int device, hilo_nmpc** out) {
  HILO_REQUIRE(d && out, "hilo_nmpc_create: NULL argument");
  HILO_REQUIRE(d->N >= 1 && d->N <= 512, "hilo_nmpc_create: horizon %d out of range [1, 512]", d->N);
  if (d->Nc != 0 && d->Nc != d->N)
    return fail(HILO_ENOTSUP, "control horizon (%d) != prediction horizon (%d) is not supported yet", d->Nc, d->N);
  HILO_REQUIRE(d->dt > 0.0, "hilo_nmpc_create: dt must be positive");
  int nx, nu, np;
  size_t lds;
  int rc = nmpc_model_dims(d->model_id, &nx, &nu, &np, &lds, d->N);
  if (rc) return rc;
  HILO_REQUIRE(nx <= NMPC_MAXNX && nu <= NMPC_MAXNU, "model too large for this build");
  if (lds > 160 * 1024)
    return fail(HILO_ENOTSUP, "horizon %d needs %zu B of LDS per instance (limit 163840)", d->N, lds);
  const int nz = nx + nu;
  hilo_nmpc* h = new hilo_nmpc();
  memset(h, 0, sizeof(*h));
  h->device = device; h->model_id = d->model_id; h->nx = nx; h->nu = nu; h->np = np; h->N = d->N;
  h->n_v = (d->N + 1) * nx + d->N * nu;  // mpc.py:1440
  h->n_g = d->N * nx;                    // mpc.py:1667-1669
  h->lds_bytes = lds;
  NmpcConst& c = h->host;
  c.N = d->N; c.order = d->erk_order >= 1 ? d->erk_order : 4; c.nsub = d->n_sub >= 1 ? d->n_sub : 1;
  c.dt = d->dt;
  c.max_iter = d->max_iter > 0 ? d->max_iter : 3000;
  c.acceptable_iter = d->acceptable_iter > 0 ? d->acceptable_iter : 15;
  double sx[NMPC_MAXNX], su[NMPC_MAXNU];
  copy_or(sx, d->x_scaling, nx, 1.0);
  copy_or(su, d->u_scaling, nu, 1.0);
  for (int i = 0; i < nz; ++i) c.sz[i] = i < nx ? sx[i] : su[i - nx];
  for (int i = 0; i < nz * nz; ++i) c.Wz[i] = d->Wz ? d->Wz[i] : 0.0;
  for (int i = 0; i < nz; ++i) c.zref[i] = d->zref ? d->zref[i] : 0.0;
  for (int i = 0; i < nx * nx; ++i) c.WN[i] = d->WN ? d->WN[i] : 0.0;
  for (int i = 0; i < nx; ++i) c.xrefN[i] = d->xrefN ? d->xrefN[i] : 0.0;
  c.has_du = d->Wdu != nullptr;
  for (int i = 0; i < nu * nu; ++i) c.Wdu[i] = d->Wdu ? d->Wdu[i] : 0.0;
  const double relax = d->bound_relax_factor >= 0.0 ? d->bound_relax_factor : 1e-8;
  for (int i = 0; i < nz; ++i) {
    // bounds arrive in original units; scaled like mpc.py:253-259, then relaxed like IPOPT's bound_relax_factor
    const double* lbs = i < nx ? d->x_lb : d->u_lb;
    const double* ubs = i < nx ? d->x_ub : d->u_ub;
    const int j = i < nx ? i : i - nx;
    double lb = lbs ? lbs[j] / c.sz[i] : -INFINITY, ub = ubs ? ubs[j] / c.sz[i] : INFINITY;
    if (lb > -INFINITY) lb -= relax * fmax(1.0, fabs(lb));
    if (ub < INFINITY) ub += relax * fmax(1.0, fabs(ub));
    HILO_REQUIRE(lb < ub, "hilo_nmpc_create: empty box for variable %d", i);
    c.lbz[i] = lb; c.ubz[i] = ub;
  }
  c.tol = d->tol > 0 ? d->tol : 1e-8;
  c.acceptable_tol = d->acceptable_tol > 0 ? d->acceptable_tol : 1e-6;
  c.mu_init = d->mu_init > 0 ? d->mu_init : 0.1;
  c.kappa_eps = 10.0; c.kappa_mu = 0.2; c.theta_mu = 1.5; c.tau_min = 0.99;
  c.bound_push = 1e-2; c.bound_frac = 1e-2; c.s_max = 100.0; c.kappa_sigma = 1e10;
  c.gamma_theta = 1e-5; c.gamma_phi = 1e-8; c.delta_ls = 1.0; c.s_theta = 1.1; c.s_phi = 2.3; c.eta_phi = 1e-8;
  c.theta_min_fact = 1e-4; c.theta_max_fact = 1e4;
  c.delta_w_min = 1e-20; c.delta_w_0 = 1e-4; c.delta_w_max = 1e40; c.kappa_w_minus = 1.0 / 3; c.kappa_w_plus = 8.0;
  c.kappa_w_plus_bar = 100.0;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc((void**)&h->dev, sizeof(NmpcConst));
  if (e == hipSuccess) e = hipMemcpy(h->dev, &c, sizeof(NmpcConst), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&h->v_guess, sizeof(double) * h->n_v);
  if (e == hipSuccess) {
    // mpc.py:1468-1482: the guess is tiled over the horizon (scaled, mpc.py:255,259)
    double* g = new double[h->n_v];
    for (int k = 0; k <= d->N; ++k)
      for (int i = 0; i < nx; ++i) g[k * nx + i] = (d->x_guess ? d->x_guess[i] : 0.0) / sx[i];
    for (int k = 0; k < d->N; ++k)
      for (int i = 0; i < nu; ++i) g[(d->N + 1) * nx + k * nu + i] = (d->u_guess ? d->u_guess[i] : 0.0) / su[i];
    e = hipMemcpy(h->v_guess, g, sizeof(double) * h->n_v, hipMemcpyHostToDevice);
    delete[] g;
  }
  if (e != hipSuccess) {
    hilo_nmpc_destroy(h);
    return fail(HILO_EHIP, "hilo_nmpc_create: %s", hipGetErrorString(e));
  }
  *out = h;
  return HILO_OK;
}

extern "C" int hilo_nmpc_dims(const hilo_nmpc* h, int* n_v, int* n_g, int* nx, int* nu, int* np) {
  HILO_REQUIRE(h, "hilo_nmpc_dims: NULL handle");
  if (n_v) *n_v = h->n_v;
  if (n_g) *n_g = h->n_g;
  if (nx) *nx = h->nx;
  if (nu) *nu = h->nu;
  if (np) *np = h->np;
  return HILO_OK;
}

extern "C" int hilo_nmpc_reset_warm_start(hilo_nmpc* h) {
  HILO_REQUIRE(h, "hilo_nmpc_reset_warm_start: NULL handle");
  h->warm_valid = 0;
  return HILO_OK;
}

template <class M>
static int nmpc_launch(hilo_nmpc* h, int64_t batch, const double* x0, const double* p, int64_t ps, const double* v0,
                       int64_t v0s, const double* u_old, double* v_opt, double* f_opt, double* lam_g, double* u0,
                       int32_t* status, int32_t* iters, double* kkt, hipStream_t s) {
  if (h->lds_bytes > 64 * 1024)
    HILO_HIP_CHECK(hipFuncSetAttribute((const void*)nmpc_solve_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)h->lds_bytes));
  hipLaunchKernelGGL((nmpc_solve_kernel<M>), dim3((unsigned)batch), dim3(64), h->lds_bytes, s, h->dev, batch, x0, p, ps,
                     v0, v0s, u_old, v_opt, f_opt, lam_g, u0, status, iters, kkt);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

extern "C" int hilo_nmpc_solve(hilo_nmpc* h, int64_t batch, const double* x0, const double* p, int64_t p_stride,
                               const double* v0, const double* u_old, double* v_opt, double* f_opt, double* lam_g,
                               double* u0, int32_t* status, int32_t* iters, double* kkt, void* stream) {
  HILO_REQUIRE(h, "hilo_nmpc_solve: NULL handle");
  HILO_REQUIRE(batch >= 0, "hilo_nmpc_solve: negative batch");
  if (batch == 0) return HILO_OK;
  HILO_REQUIRE(x0 && v_opt && f_opt && u0 && status && iters, "hilo_nmpc_solve: NULL argument");
  HILO_REQUIRE(h->np == 0 || p, "hilo_nmpc_solve: the model has %d parameters but p is NULL (mpc.py:771-780)", h->np);
  HILO_REQUIRE(p_stride == 0 || p_stride >= h->np, "hilo_nmpc_solve: p_stride %lld < np", (long long)p_stride);
  HILO_HIP_CHECK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  // initial guess: explicit v0, else the previous solution (warm start, mpc.py:725-726), else the tiled guess
  const double* vstart = v0;
  int64_t vstride = h->n_v;
  if (!vstart) {
    if (h->warm_valid && h->warm_batch == batch) vstart = h->v_warm;
    else { vstart = h->v_guess; vstride = 0; }
  }
  int rc = HILO_ENOTSUP;
  switch (h->model_id) {
#define X(ID, T) case ID: rc = nmpc_launch<T>(h, batch, x0, p, p_stride, vstart, vstride, u_old, v_opt, f_opt, lam_g, u0, status, iters, kkt, s); break;
    HILO_NMPC_MODELS(X)
#undef X
  }
  if (rc) return rc;
  // keep the solution for the next call (un-shifted, like the reference)
  if (h->warm_batch != batch) {
    if (h->v_warm) HILO_HIP_CHECK(hipFree(h->v_warm));
    h->v_warm = nullptr;
    hipError_t e = hipMalloc((void**)&h->v_warm, sizeof(double) * h->n_v * batch);
    if (e != hipSuccess) return fail(HILO_ENOMEM, "warm-start buffer: %s", hipGetErrorString(e));
    h->warm_batch = batch;
  }
  HILO_HIP_CHECK(hipMemcpyAsync(h->v_warm, v_opt, sizeof(double) * h->n_v * batch, hipMemcpyDeviceToDevice, s));
  h->warm_valid = 1;
  return HILO_OK;
}

extern "C" int hilo_nmpc_plant_step(hilo_nmpc* h, int64_t batch, const double* x, const double* u, const double* p,
                                    int64_t p_stride, double* x_next, void* stream) {
  HILO_REQUIRE(h, "hilo_nmpc_plant_step: NULL handle");
  if (batch <= 0) return HILO_OK;
  HILO_REQUIRE(x && u && x_next && (h->np == 0 || p), "hilo_nmpc_plant_step: NULL argument");
  HILO_HIP_CHECK(hipSetDevice(h->device));
  const unsigned grid = (unsigned)((batch + 255) / 256);
  switch (h->model_id) {
#define X(ID, T) case ID: hipLaunchKernelGGL((plant_step_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, h->dev, batch, x, u, p, p_stride, x_next); break;
    HILO_NMPC_MODELS(X)
#undef X
  }
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}
