// Direct collocation (the reference's default `integration_method`) as an implicit shooting map.
//
// Reference transcription (hilo_mpc/util/modeling.py:1091-1211, hilo_mpc/modules/controller/mpc.py:1307-1372, :1497-1518):
// per interval the d collocation states x_{k,i} are decision variables with
//     dt f(x_{k,i}, u_k) - sum_{j=0..d} C[j,i] x_{k,j} = 0,  x_{k,0} = x_k,      x_{k+1} - sum_j D_j x_{k,j} = 0.
// Here the square collocation system of an interval is solved to round-off inside the shooting map (Newton in the
// Runge-Kutta form  X_i = x + dt sum_j A_ij f(X_j, u),  A = (C_hat^T)^-1, C_hat = C[1:,1:]: same equations, pre-multiplied by
// a constant matrix so that the Newton matrix I - dt (A (x) I) blockdiag(f_x) needs no pivoting), and the map
// x+ = sum_j D_j X_j is differentiated exactly: two further Newton sweeps in second-order Taylor arithmetic on the converged
// point give the first- and second-order coefficients (the k-th sweep fixes the k-th coefficient).  The NLP the
// interior-point engine sees therefore has the reference's KKT points (the collocation states and their multipliers are
// reconstructed on output); the iterates differ from a solver that carries the collocation states as variables, and the
// box of the state is enforced at the shooting nodes, not at the interior collocation points.
#pragma once
#include "hilo_models.h"

namespace hilo {

constexpr int COLL_MAXD = 4;

struct CollData {  // host-computed basis (hilo_mpc_amd/nmpc.py restates modeling.py:1091-1127)
  int d, pad;
  double A[COLL_MAXD * COLL_MAXD];   // Runge-Kutta matrix of the collocation method
  double Dc[COLL_MAXD + 1];          // continuity weights D_0..D_d
  double Bq[COLL_MAXD + 1];          // quadrature weights B_0..B_d = int_0^1 L_i (continuous objective, modeling.py:1195)
};

// ---- the Newton matrix of an interval factored by the lanes of a wave TOGETHER -----------------------------------------------
// One COLUMN of the augmented matrix [Mat | right-hand sides] per lane, all of it in registers (DN doubles): lane c < DN of a
// group owns column c of Mat, the lanes behind them one right-hand side each.  A pivot step sends the pivot column's multipliers
// to every lane of the group (two ds_bpermute_b32 per double) and each lane updates its own column - no lane ever holds the
// matrix, nothing goes through memory.  (A private DN x DN matrix per lane is 3.5 KB for configuration 5's 21 x 21 systems:
// 8.8 KB of scratch per lane and 2.9 TB of traffic per launch, profiles/r05_C5-dae_summary.json.)  No pivoting: the Runge-Kutta
// form of the collocation equations (see below) keeps the matrix within O(dt) of the identity.
__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <int DN>
struct CoopLU {
  // c: this lane's column within its group (>= DN: a right-hand side), gbase: first lane of the group.  Afterwards a matrix lane
  // holds its column of U on and above the diagonal (the diagonal as its RECIPROCAL) and the multipliers of L below it; a
  // right-hand-side lane holds L^-1 b.
  // KEEP_L = false (the factors only serve the right-hand sides of this call): nothing below the diagonal is kept, so every lane
  // applies every update to its column unconditionally - a matrix lane's entries below its diagonal are never read again
  template <bool KEEP_L = true>
  __device__ __forceinline__ static void eliminate(double* col, int c, int gbase) {
#pragma unroll
    for (int k = 0; k < DN; ++k) {
      const double ip = rcp_fast(col[k]);
      double l[DN > 1 ? DN : 1];
#pragma unroll
      for (int i = k + 1; i < DN; ++i) l[i] = lane_bcast(col[i] * ip, gbase + k);
      if constexpr (KEEP_L) {
        if (c > k) {
#pragma unroll
          for (int i = k + 1; i < DN; ++i) col[i] = fma(-l[i], col[k], col[i]);
        } else if (c == k) {
          col[k] = ip;
#pragma unroll
          for (int i = k + 1; i < DN; ++i) col[i] = l[i];
        }
      } else {
        const double ck = col[k];
        col[k] = c == k ? ip : ck;
#pragma unroll
        for (int i = k + 1; i < DN; ++i) col[i] = fma(-l[i], ck, col[i]);
      }
    }
  }
  // The same with the right-hand sides as a SECOND column `ecol` per lane (a lane may own a matrix column and a right-hand side,
  // or only one of them with the other column idle): every lane applies the row operations to its `ecol`.
#ifdef HILO_COLL_NO_LOOKAHEAD
  template <bool KEEP_L = true>
  __device__ __forceinline__ static void eliminate2(double* col, double* ecol, int c, int gbase) {
#pragma unroll
    for (int k = 0; k < DN; ++k) {
      const double ck = col[k], ek = ecol[k];
      const double ip = rcp_fast(ck);
      col[k] = c == k ? ip : ck;
      double l[DN > 1 ? DN : 1];
#pragma unroll
      for (int i = k + 1; i < DN; ++i) l[i] = lane_bcast(col[i] * ip, gbase + k);
#pragma unroll
      for (int i = k + 1; i < DN; ++i) {
        ecol[i] = fma(-l[i], ek, ecol[i]);
        asm volatile("" : "+v"(ecol[i]));
        const double ci = fma(-l[i], ck, col[i]);
        if constexpr (KEEP_L) col[i] = c == k ? l[i] : ci;
        else col[i] = ci;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#else
  template <bool KEEP_L = true>
  __device__ __forceinline__ static void eliminate2(double* col, double* ecol, int c, int gbase) {
    // LOOK-AHEAD: the multipliers of step k + 1 are formed and their broadcasts requested as soon as the matrix columns have
    // taken step k - the right-hand sides' updates of step k then run while those broadcasts are in flight (one wave per SIMD:
    // nothing else hides a ds_bpermute round trip).  Branch-free: a lane left of the pivot updates entries it never reads
    // again, the pivot's own lane keeps its multipliers (KEEP_L).
    double l[DN > 1 ? DN : 1], ln[DN > 1 ? DN : 1];
    double ip = rcp_fast(col[0]);
#pragma unroll
    for (int i = 1; i < DN; ++i) l[i] = lane_bcast(col[i] * ip, gbase);
#pragma unroll
    for (int k = 0; k < DN; ++k) {
      const double ck = col[k], ek = ecol[k];
      col[k] = c == k ? ip : ck;
#pragma unroll
      for (int i = k + 1; i < DN; ++i) {
        const double ci = fma(-l[i], ck, col[i]);
        if constexpr (KEEP_L) col[i] = c == k ? l[i] : ci;
        else col[i] = ci;
      }
      if (k + 1 < DN) {
        ip = rcp_fast(col[k + 1]);
#pragma unroll
        for (int i = k + 2; i < DN; ++i) ln[i] = lane_bcast(col[i] * ip, gbase + k + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = k + 1; i < DN; ++i) {
        ecol[i] = fma(-l[i], ek, ecol[i]);
        // (pinned here: when only one lane's right-hand side is read afterwards - inside that lane's branch - the optimiser sinks
        // the whole update chain into the branch, behind the elimination, and keeps every multiplier alive for it: 210 doubles)
        asm volatile("" : "+v"(ecol[i]));
      }
      __builtin_amdgcn_sched_barrier(0);   // (the unrolled steps are ONE basic block: left alone the scheduler runs all pivot columns
                                           // first and the right-hand sides last, with every multiplier kept in between)
#pragma unroll
      for (int i = k + 2; i < DN; ++i) l[i] = ln[i];
    }
  }
#endif
  // U x = y for the right-hand sides, the factor read from LDS: every matrix lane writes its column (on and above the diagonal;
  // the diagonal is the reciprocal pivot) to `us` (DN x DN doubles of the group), every lane then substitutes its own right-hand
  // side with broadcast reads - one LDS read per entry of U where lane-to-lane broadcasts took two ds_bpermute each.
  __device__ __forceinline__ static void back_substitute2_lds(const double* col, double* ecol, int c,
                                                              __attribute__((address_space(3))) double* us) {
    if (c < DN) {
#pragma unroll
      for (int i = 0; i < DN; ++i) us[c * DN + i] = col[i];
    }
    __syncthreads();
#pragma unroll
    for (int k = DN - 1; k >= 0; --k) {
      double uk[DN];
#pragma unroll
      for (int i = 0; i <= k; ++i) uk[i] = us[k * DN + i];     // the column's reads requested together
      const double xk = ecol[k] * uk[k];
      ecol[k] = xk;
#pragma unroll
      for (int i = 0; i < k; ++i) {
        ecol[i] = fma(-uk[i], xk, ecol[i]);
        asm volatile("" : "+v"(ecol[i]));   // (pinned: see eliminate2)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  __device__ __forceinline__ static void back_substitute2(const double* col, double* ecol, int gbase) {
#pragma unroll
    for (int k = DN - 1; k >= 0; --k) {
      double uk[DN];
#pragma unroll
      for (int i = 0; i <= k; ++i) uk[i] = lane_bcast(col[i], gbase + k);   // column k of U from its owner (uk[k] = 1 / u_kk)
      const double xk = ecol[k] * uk[k];
      ecol[k] = xk;
#pragma unroll
      for (int i = 0; i < k; ++i) {
        ecol[i] = fma(-uk[i], xk, ecol[i]);
        asm volatile("" : "+v"(ecol[i]));   // (pinned: see eliminate2)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // right-hand-side lanes: U x = y in place
  __device__ __forceinline__ static void back_substitute(double* col, int c, int gbase) {
#pragma unroll
    for (int k = DN - 1; k >= 0; --k) {
      double uk[DN];
#pragma unroll
      for (int i = 0; i <= k; ++i) uk[i] = lane_bcast(col[i], gbase + k);   // column k of U from its owner (uk[k] = 1 / u_kk)
      if (c >= DN) {
        const double xk = col[k] * uk[k];
        col[k] = xk;
#pragma unroll
        for (int i = 0; i < k; ++i) col[i] = fma(-uk[i], xk, col[i]);
      }
    }
  }
  // Mat^T y = b with the same factors (Mat^T = U^T L^T): matrix lane c hands in b_c and receives y_c.  Dot-product form - every
  // coefficient a lane needs is in its own column; the unknowns travel lane to lane.
  __device__ __forceinline__ static double solve_transposed(const double* col, double b, int c, int gbase) {
    double s = 0.0, z = 0.0;
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      const double zc = (b - s) * col[i];          // meaningful in lane i: (b_i - sum_{r < i} u_ri z_r) / u_ii
      const double zi = lane_bcast(zc, gbase + i);
      z = c == i ? zc : z;
      s = c > i ? fma(col[i], zi, s) : s;
    }
    double t = 0.0, y = 0.0;
#pragma unroll
    for (int i = DN - 1; i >= 0; --i) {
      const double yc = z - t;                     // meaningful in lane i
      const double yi = lane_bcast(yc, gbase + i);
      y = c == i ? yc : y;
      t = c < i ? fma(col[i], yi, t) : t;
    }
    return y;
  }
};

template <class M, int D>
struct Colloc {
  static constexpr int NX = M::NX, NU = M::NU, DN = D * NX;

  // in-place LU without pivoting of the DN x DN matrix a (row-major)
  __device__ __forceinline__ static void lu(double* a) {
#pragma unroll
    for (int k = 0; k < DN; ++k) {
      const double ip = rcp_fast(a[k * DN + k]);
#pragma unroll
      for (int i = k + 1; i < DN; ++i) {
        const double f = a[i * DN + k] * ip;
        a[i * DN + k] = f;
#pragma unroll
        for (int j = k + 1; j < DN; ++j) a[i * DN + j] -= f * a[k * DN + j];
      }
    }
  }
  template <class AP>
  __device__ __forceinline__ static void lu_solve(AP a, double* b) {
#pragma unroll
    for (int i = 1; i < DN; ++i) {
      double s = b[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= a[i * DN + j] * b[j];
      b[i] = s;
    }
#pragma unroll
    for (int i = DN - 1; i >= 0; --i) {
      double s = b[i];
#pragma unroll
      for (int j = i + 1; j < DN; ++j) s -= a[i * DN + j] * b[j];
      b[i] = s / a[i * DN + i];
    }
  }
  // two right-hand sides at once against PREPARED factors (`prepare`: the diagonal of U stored as its reciprocal): a row of the
  // factors is fetched as one group of independent loads while the previous row's multiply-adds run (the factors sit in LDS or
  // global memory; left to itself the compiler issued load - wait - multiply-add per element: 38 k cycles per solve at one wave
  // per SIMD, measured on configuration 5's 21 x 21 systems)
  template <class AP>
  __device__ __forceinline__ static void lu_solve2_prepared(AP a, double* b, double* c) {
    double row[DN], nxt[DN];
    row[0] = DN > 1 ? a[1 * DN + 0] : a[0];   // (a 1 x 1 system: no forward pass, the backward pass starts with 1 / u_00)
#pragma unroll
    for (int i = 1; i < DN; ++i) {                       // forward: unit lower factor, row i holds a[i][0..i-1]
      if (i + 1 < DN) {
#pragma unroll
        for (int j = 0; j <= i; ++j) nxt[j] = a[(i + 1) * DN + j];
      } else {
#pragma unroll
        for (int j = DN - 1; j < DN; ++j) nxt[j] = a[(DN - 1) * DN + j];   // first row of the backward pass
      }
      __builtin_amdgcn_sched_barrier(0);
      double s = b[i], t = c[i];
#pragma unroll
      for (int j = 0; j < i; ++j) { s -= row[j] * b[j]; t -= row[j] * c[j]; }
      b[i] = s;
      c[i] = t;
#pragma unroll
      for (int j = 0; j < DN; ++j) row[j] = nxt[j];
    }
#pragma unroll
    for (int i = DN - 1; i >= 0; --i) {                  // backward: row i holds a[i][i..DN-1], a[i][i] = 1 / u_ii
      if (i > 0) {
#pragma unroll
        for (int j = i - 1; j < DN; ++j) nxt[j] = a[(i - 1) * DN + j];
      }
      __builtin_amdgcn_sched_barrier(0);
      double s = b[i], t = c[i];
#pragma unroll
      for (int j = i + 1; j < DN; ++j) { s -= row[j] * b[j]; t -= row[j] * c[j]; }
      b[i] = s * row[i];
      c[i] = t * row[i];
#pragma unroll
      for (int j = 0; j < DN; ++j) row[j] = nxt[j];
    }
  }
  // solve a^T y = b with the same factors (a = L U  =>  a^T = U^T L^T)
  __device__ __forceinline__ static void lu_solve_t(const double* a, double* b) {
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      double s = b[i];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= a[j * DN + i] * b[j];
      b[i] = s / a[i * DN + i];
    }
#pragma unroll
    for (int i = DN - 1; i >= 0; --i) {
      double s = b[i];
#pragma unroll
      for (int j = i + 1; j < DN; ++j) s -= a[j * DN + i] * b[j];
      b[i] = s;
    }
  }

  // Newton matrix I - dt (A (x) I) blockdiag(J_j) at the points X, and F_j = f(X_j, u)
  __device__ __forceinline__ static void newton_matrix(const CollData& cd, const double* X, const double* u, const double* p,
                                                       double dt, double* mat, double* F) {
    double J[D][NX * NX];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      Dual<NX> xd[NX], fd[NX];
#pragma unroll
      for (int a = 0; a < NX; ++a) {
        xd[a] = Dual<NX>(X[j * NX + a]);
        xd[a].d[a] = 1.0;
      }
      M::ode(xd, u, p, dt, fd);
#pragma unroll
      for (int m = 0; m < NX; ++m) {
        F[j * NX + m] = fd[m].v;
#pragma unroll
        for (int a = 0; a < NX; ++a) J[j][m * NX + a] = fd[m].d[a];
      }
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int m = 0; m < NX; ++m)
#pragma unroll
        for (int j = 0; j < D; ++j)
#pragma unroll
          for (int a = 0; a < NX; ++a)
            mat[(i * NX + m) * DN + j * NX + a] = ((i == j && m == a) ? 1.0 : 0.0) - dt * cd.A[i * D + j] * J[j][m * NX + a];
  }

  // collocation states of one interval (values); leaves the LU factors of the Newton matrix at the solution in `mat`
  __device__ __forceinline__ static void solve(const CollData& cd, const double* x, const double* u, const double* p, double dt,
                                               double* X, double* mat) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int m = 0; m < NX; ++m) X[i * NX + m] = x[m];
    for (int it = 0; it < 12; ++it) {
      double F[DN], R[DN];
      newton_matrix(cd, X, u, p, dt, mat, F);
      double scale = 1.0;
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int m = 0; m < NX; ++m) {
          double r = X[i * NX + m] - x[m];
#pragma unroll
          for (int j = 0; j < D; ++j) r -= dt * cd.A[i * D + j] * F[j * NX + m];
          R[i * NX + m] = -r;
          scale = fmax(scale, fabs(X[i * NX + m]));
        }
      lu(mat);
      lu_solve(mat, R);
      double dmax = 0.0;
#pragma unroll
      for (int q = 0; q < DN; ++q) {
        X[q] += R[q];
        dmax = fmax(dmax, fabs(R[q]));
      }
      // iterate to round-off: the factors kept in `mat` then belong to a point within 1e-13 of the solution, which is what
      // makes the Taylor sweeps below exact.  A NaN (singular pivot) also leaves the loop.  The exit is WAVE-UNIFORM (every
      // lane iterates until the last one has converged; a converged lane's further steps are round-off): the exit of a
      // lane-dependent loop is a join block, and under this function's register pressure the register allocator of ROCm 7.2
      // put live-range copies in front of that block's EXEC restore (tools/check_exec_prologue.py)
      if (!__any((int)(dmax > 1e-13 * scale))) break;
    }
  }

  // `Xout` (optional, DN entries): the collocation states themselves, with their Taylor coefficients - what the continuous
  // objective integrates the Lagrange term over
  template <class T>
  __device__ __forceinline__ static void step(const CollData& cd, const T* x, const T* u, const double* p, double dt, T* xn,
                                              T* Xout = nullptr) {
    double xv[NX], uv[NU > 0 ? NU : 1], X[DN], mat[DN * DN];
#pragma unroll
    for (int m = 0; m < NX; ++m) xv[m] = value(x[m]);
#pragma unroll
    for (int a = 0; a < NU; ++a) uv[a] = value(u[a]);
    solve(cd, xv, uv, p, dt, X, mat);
    if constexpr (same_type<T, double>::value) {
#pragma unroll
      for (int m = 0; m < NX; ++m) {
        double s = cd.Dc[0] * xv[m];
#pragma unroll
        for (int i = 0; i < D; ++i) s += cd.Dc[i + 1] * X[i * NX + m];
        xn[m] = s;
      }
      if (Xout) {
#pragma unroll
        for (int q = 0; q < DN; ++q) Xout[q] = X[q];
      }
    } else {
      Jet2 XJ[DN];
#pragma unroll
      for (int q = 0; q < DN; ++q) XJ[q] = Jet2(X[q]);
#pragma unroll 1
      for (int rep = 0; rep < 2; ++rep) {
        Jet2 F[DN];
#pragma unroll
        for (int j = 0; j < D; ++j) M::ode(XJ + j * NX, u, p, dt, F + j * NX);
        double ra[DN], rb[DN];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
          for (int m = 0; m < NX; ++m) {
            Jet2 r = XJ[i * NX + m] - x[m];
#pragma unroll
            for (int j = 0; j < D; ++j) r = r - (dt * cd.A[i * D + j]) * F[j * NX + m];
            ra[i * NX + m] = -r.a;
            rb[i * NX + m] = -r.b;
          }
        lu_solve(mat, ra);
        lu_solve(mat, rb);
#pragma unroll
        for (int q = 0; q < DN; ++q) {
          XJ[q].a += ra[q];
          XJ[q].b += rb[q];
        }
      }
#pragma unroll
      for (int m = 0; m < NX; ++m) {
        Jet2 s = cd.Dc[0] * x[m];
#pragma unroll
        for (int i = 0; i < D; ++i) s = s + cd.Dc[i + 1] * XJ[i * NX + m];
        xn[m] = s;
      }
      if (Xout) {
#pragma unroll
        for (int q = 0; q < DN; ++q) Xout[q] = XJ[q];
      }
    }
  }

  // The Taylor part of `step` on PREPARED data: `prep` = [converged collocation states X (DN) | LU factors of the Newton matrix at
  // them (DN x DN)], written once per interval by `prepare` (the values and the factors are the same for every direction of an
  // interval; `step` would repeat the Newton solve per direction).  Same arithmetic, same numbers as `step`.
  template <class PP>
  __device__ __forceinline__ static void prepare(const CollData& cd, const double* x, const double* u, const double* p, double dt,
                                                 PP prep) {
    double X[DN], mat[DN * DN];
    solve(cd, x, u, p, dt, X, mat);
#pragma unroll
    for (int q = 0; q < DN; ++q) prep[q] = X[q];
#pragma unroll
    for (int i = 0; i < DN; ++i)
#pragma unroll
      for (int j = 0; j < DN; ++j) prep[DN + i * DN + j] = i == j ? 1.0 / mat[i * DN + j] : mat[i * DN + j];   // (lu_solve2_prepared)
  }
  template <class PP>
  __device__ __forceinline__ static void step_prepared(const CollData& cd, const Jet2* x, const Jet2* u, const double* p, double dt,
                                                       Jet2* xn, Jet2* Xout, PP prep) {
    Jet2 XJ[DN];
#pragma unroll
    for (int q = 0; q < DN; ++q) XJ[q] = Jet2(prep[q]);
#pragma unroll 1
    for (int rep = 0; rep < 2; ++rep) {
      Jet2 F[DN];
#pragma unroll
      for (int j = 0; j < D; ++j) M::ode(XJ + j * NX, u, p, dt, F + j * NX);
      double ra[DN], rb[DN];
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int m = 0; m < NX; ++m) {
          Jet2 r = XJ[i * NX + m] - x[m];
#pragma unroll
          for (int j = 0; j < D; ++j) r = r - (dt * cd.A[i * D + j]) * F[j * NX + m];
          ra[i * NX + m] = -r.a;
          rb[i * NX + m] = -r.b;
        }
#ifndef HILO_DBG_SKIP_LUSOLVE
      lu_solve2_prepared(prep + DN, ra, rb);
#endif
#pragma unroll
      for (int q = 0; q < DN; ++q) {
        XJ[q].a += ra[q];
        XJ[q].b += rb[q];
      }
    }
#pragma unroll
    for (int m = 0; m < NX; ++m) {
      Jet2 s = cd.Dc[0] * x[m];
#pragma unroll
      for (int i = 0; i < D; ++i) s = s + cd.Dc[i + 1] * XJ[i * NX + m];
      xn[m] = s;
    }
    if (Xout) {
#pragma unroll
      for (int q = 0; q < DN; ++q) Xout[q] = XJ[q];
    }
  }

  // multipliers of the reference's collocation rows G_i = dt f(X_i,u) - sum_j C[j,i] X_j at a KKT point with inactive
  // collocation-state bounds: G = -(C_hat^T (x) I) R with R the Runge-Kutta residual solved above, so
  //   G_X^T mu = (D (x) lambda)   <=>   mu = -(A^T (x) I) Mat^-T (D (x) lambda)
  __device__ static void multipliers(const CollData& cd, const double* X, const double* u, const double* p, double dt,
                                     const double* lam, double* mu) {
    double mat[DN * DN], F[DN], y[DN];
    newton_matrix(cd, X, u, p, dt, mat, F);
    lu(mat);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int a = 0; a < NX; ++a) y[i * NX + a] = cd.Dc[i + 1] * lam[a];
    lu_solve_t(mat, y);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int a = 0; a < NX; ++a) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) s -= cd.A[j * D + i] * y[j * NX + a];
        mu[i * NX + a] = s;
      }
  }
};

}  // namespace hilo
