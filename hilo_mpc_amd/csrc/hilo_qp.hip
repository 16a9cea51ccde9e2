// Batched dense convex QP solver (one wave per instance, everything LDS-resident) + C ABI: the LMPC path.
//
// Replaces the conic solver object of `LMPC.setup` - `ca.conic("solver", 'qpoases', {'h','a'})`
// (hilo_mpc/modules/controller/mpc.py:2268-2276) - called from `LMPC.optimize` as
// `solver(h=H, g=g, a=A, lbx, ubx, lba, uba)` (mpc.py:2374):
//     min 1/2 x^T H x + g^T x   s.t.  lba <= A x <= uba,  lbx <= x <= ubx
// The reference's Aeq is not stage-banded in general (its input block is `kron(B, I_N)`, mpc.py:2243), so
// this path is a general dense solver: Mehrotra predictor-corrector interior point; per iteration one Cholesky of
// H + Sigma (n x n), the Schur complement A (H+Sigma)^-1 A^T (m x m) and its Cholesky, two pairs of triangular
// solves.  Variables with lbx == ubx (the pinned x_0, mpc.py:2361-2362) are substituted.  Rows must be equalities
// (lba == uba), which is all the reference generates (polytopic constraints are a TODO there, mpc.py:2249-2250).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "hilo_common.h"

namespace hilo {

struct QpDims { int n, m, ldn, ldm, max_iter; double tol, reg; };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// the same reductions on the data-parallel primitives (cross-lane moves inside the vector ALU instead of LDS permutes, a sixth of
// the latency): butterfly inside the rows of 16 lanes, then row_bcast15 / row_bcast31; the total is read from lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double qp_dpp(double v, double ident) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <class Op>
__device__ __forceinline__ double qp_wave_reduce(double v, double ident, Op op) {
  v = op(v, qp_dpp<0xB1, 0xf>(v, ident));     // quad_perm [1, 0, 3, 2]
  v = op(v, qp_dpp<0x4E, 0xf>(v, ident));     // quad_perm [2, 3, 0, 1]
  v = op(v, qp_dpp<0x141, 0xf>(v, ident));    // row_half_mirror
  v = op(v, qp_dpp<0x140, 0xf>(v, ident));    // row_mirror: every lane of a row holds the row's total
  v = op(v, qp_dpp<0x142, 0xa>(v, ident));    // row_bcast15 into rows 1 and 3
  v = op(v, qp_dpp<0x143, 0xc>(v, ident));    // row_bcast31 into rows 2 and 3
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double qp_wave_sum(double v) { return qp_wave_reduce(v, 0.0, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ double qp_wave_min(double v) { return qp_wave_reduce(v, INFINITY, [](double a, double b) { return fmin(a, b); }); }
__device__ __forceinline__ double qp_wave_max(double v) { return qp_wave_reduce(v, -INFINITY, [](double a, double b) { return fmax(a, b); }); }

// in-place lower Cholesky of the n x n matrix M (leading dimension ld, odd -> conflict-free column access)
__device__ bool chol_lds(double* M, int n, int ld) {
  const int t = threadIdx.x;
  bool ok = true;
  for (int j = 0; j < n; ++j) {
    const double ajj = M[j * ld + j];
    if (!(ajj > 0.0)) ok = false;
    const double d = sqrt(ajj > 0.0 ? ajj : 1.0), id = 1.0 / d;
    __syncthreads();
    for (int i = j + t; i < n; i += 64) M[i * ld + j] = (i == j) ? d : M[i * ld + j] * id;
    __syncthreads();
    for (int i = j + 1 + t; i < n; i += 64) {
      const double lij = M[i * ld + j];
      for (int k = j + 1; k <= i; ++k) M[i * ld + k] -= lij * M[k * ld + j];
    }
    __syncthreads();
  }
  return ok;
}

// X (n x m, ld ldx) <- L^-1 X, one column per lane
__device__ void trsm_lds(const double* L, int n, int ld, double* X, int m, int ldx) {
  for (int c = threadIdx.x; c < m; c += 64) {
    for (int i = 0; i < n; ++i) {
      double s = X[i * ldx + c];
      for (int j = 0; j < i; ++j) s -= L[i * ld + j] * X[j * ldx + c];
      X[i * ldx + c] = s / L[i * ld + i];
    }
  }
  __syncthreads();
}

// r (length n, in LDS) <- L^-1 r (forward) or L^-T r (backward), column oriented across lanes
__device__ void trsv_lds(const double* L, int n, int ld, double* r, bool transpose) {
  const int t = threadIdx.x;
  if (!transpose) {
    for (int j = 0; j < n; ++j) {
      if (t == 0) r[j] /= L[j * ld + j];
      __syncthreads();
      const double rj = r[j];
      for (int i = j + 1 + t; i < n; i += 64) r[i] -= L[i * ld + j] * rj;
      __syncthreads();
    }
  } else {
    for (int j = n - 1; j >= 0; --j) {
      if (t == 0) r[j] /= L[j * ld + j];
      __syncthreads();
      const double rj = r[j];
      for (int i = t; i < j; i += 64) r[i] -= L[j * ld + i] * rj;
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(64) void qp_solve_kernel(QpDims qd, int64_t batch, const double* __restrict__ Hg, int64_t hs,
                                                      const double* __restrict__ gg, int64_t gs,
                                                      const double* __restrict__ Ag, int64_t as_,
                                                      const double* __restrict__ lbx, const double* __restrict__ ubx,
                                                      int64_t bs, const double* __restrict__ lba,
                                                      const double* __restrict__ uba, int64_t bas,
                                                      double* __restrict__ x_out, double* __restrict__ f_out,
                                                      double* __restrict__ lam_a, double* __restrict__ lam_x,
                                                      int32_t* __restrict__ status, int32_t* __restrict__ iters,
                                                      double* __restrict__ ws, int64_t ws_stride,
                                                      const double* __restrict__ xpin, int npin, int64_t ps) {
  extern __shared__ double lds[];
  const int n = qd.n, m = qd.m, ldn = qd.ldn, ldm = qd.ldm, t = threadIdx.x;
  const int64_t b = blockIdx.x;
  if (b >= batch) return;
  // working set in LDS, or - for QPs beyond 160 KB - in a per-instance workspace in global memory (same code, flat pointers)
  double* q = ws ? ws + b * ws_stride : lds;
  auto take = [&](size_t k) { double* r = q; q += k; return r; };
  double* Hf = take((size_t)n * ldn);   // H with fixed rows/cols removed (identity there)
  double* Mc = take((size_t)n * ldn);   // H + Sigma -> its Cholesky factor
  double* Af = take((size_t)m * ldn);   // A with fixed columns zeroed
  double* X = take((size_t)n * ldm);    // L^-1 Af^T
  double* S = take((size_t)m * ldm);    // Schur complement -> its Cholesky factor
  double *x = take(n), *gf = take(n), *l = take(n), *u = take(n), *zl = take(n), *zu = take(n), *r1 = take(n),
         *dx = take(n), *dzl = take(n), *dzu = take(n), *base = take(n), *xfix = take(n);
  double *y = take(m), *bf = take(m), *rp = take(m), *dy = take(m);
  const double* H = Hg + b * hs;
  const double* A = Ag + b * as_;
  const double* g = gg + b * gs;

  // ---- load, substitute fixed variables (lbx == ubx) ----
  int bad_rows = 0;
  for (int i = t; i < n; i += 64) {
    double lo, up;
    qp_bounds(lbx, ubx, bs, xpin, npin, ps, b, i, lo, up);
    const bool fx = lo == up;
    l[i] = fx ? NAN : lo;  // NAN marks a fixed variable
    u[i] = up;
    xfix[i] = fx ? lo : 0.0;
  }
  for (int r = t; r < m; r += 64)
    if (lba[b * bas + r] != uba[b * bas + r]) bad_rows = 1;
  bad_rows = __any(bad_rows);
  __syncthreads();
  for (int e = t; e < n * n; e += 64) {
    const int i = e / n, j = e - i * n;
    const bool fi = isnan(l[i]), fj = isnan(l[j]);
    Hf[i * ldn + j] = (fi || fj) ? (i == j ? 1.0 : 0.0) : H[e];
  }
  for (int e = t; e < m * n; e += 64) {
    const int r = e / n, j = e - r * n;
    Af[r * ldn + j] = isnan(l[j]) ? 0.0 : A[e];
  }
  for (int i = t; i < n; i += 64) {
    double s = g[i];
    for (int j = 0; j < n; ++j) s += H[i * n + j] * xfix[j];
    gf[i] = isnan(l[i]) ? 0.0 : s;
  }
  for (int r = t; r < m; r += 64) {
    double s = uba[b * bas + r];
    for (int j = 0; j < n; ++j) s -= A[r * n + j] * xfix[j];
    bf[r] = s;
    y[r] = 0.0;
  }
  // starting point: strictly inside the box, unit multipliers
  double nb_part = 0.0;
  for (int i = t; i < n; i += 64) {
    const bool fx = isnan(l[i]);
    const bool hl = !fx && l[i] > -INFINITY, hu = !fx && u[i] < INFINITY;
    double xi = 0.0;
    if (hl && hu) xi = 0.5 * (l[i] + u[i]);
    else if (hl) xi = fmax(0.0, l[i] + 1.0);
    else if (hu) xi = fmin(0.0, u[i] - 1.0);
    x[i] = fx ? 0.0 : xi;
    zl[i] = hl ? 1.0 : 0.0;
    zu[i] = hu ? 1.0 : 0.0;
    nb_part += (hl ? 1.0 : 0.0) + (hu ? 1.0 : 0.0);
  }
  const double nb = fmax(1.0, wave_sum(nb_part));
  double gmax = 0.0;
  for (int i = t; i < n; i += 64) gmax = fmax(gmax, fabs(gf[i]));
  gmax = wave_max(gmax);
  __syncthreads();

  int st = HILO_STATUS_MAXITER, it = 0;
  double phi_min = INFINITY;
  if (bad_rows) st = HILO_STATUS_OTHER;
  for (it = 0; !bad_rows && it < qd.max_iter; ++it) {
    // residuals: base = -(Hf x + gf + Af^T y), rd = -base - zl + zu, rp = Af x - bf, mu
    double rdmax = 0.0, mupart = 0.0, nonfinite = 0.0;   // fmax() drops NaN: count non-finite residuals explicitly
    for (int i = t; i < n; i += 64) {
      double s = gf[i];
      for (int j = 0; j < n; ++j) s += Hf[i * ldn + j] * x[j];
      for (int r = 0; r < m; ++r) s += Af[r * ldn + i] * y[r];
      const bool fx = isnan(l[i]);
      if (fx) s = 0.0;
      base[i] = -s;
      rdmax = fmax(rdmax, fabs(s - zl[i] + zu[i]));
      nonfinite += isfinite(s - zl[i] + zu[i]) ? 0.0 : 1.0;
      if (!fx) {
        if (l[i] > -INFINITY) mupart += (x[i] - l[i]) * zl[i];
        if (u[i] < INFINITY) mupart += (u[i] - x[i]) * zu[i];
      }
    }
    double rpmax = 0.0;
    for (int r = t; r < m; r += 64) {
      double s = -bf[r];
      for (int j = 0; j < n; ++j) s += Af[r * ldn + j] * x[j];
      rp[r] = s;
      rpmax = fmax(rpmax, fabs(s));
      nonfinite += isfinite(s) ? 0.0 : 1.0;
    }
    rdmax = wave_max(rdmax);
    rpmax = wave_max(rpmax);
    const double mu = wave_sum(mupart) / nb;
    nonfinite = wave_sum(nonfinite);
    if (nonfinite > 0.0 || !isfinite(rdmax) || !isfinite(rpmax) || !isfinite(mu)) { st = HILO_STATUS_INFEASIBLE; break; }
    const double phi = fmax(fmax(rdmax / (1.0 + gmax), rpmax), mu);
    if (phi <= qd.tol) { st = HILO_STATUS_SOLVED; break; }
    // infeasible QP (the reference's qpOASES reports it; a predictor-corrector iteration diverges instead): the termination rule
    // of OOQP (Gertz & Wright, ACM TOMS 29, 2003) - the merit has grown to 1e4 times its smallest value so far
    phi_min = fmin(phi_min, phi);
    if (phi >= 1.0e4 * phi_min) { st = HILO_STATUS_INFEASIBLE; break; }
    // M = Hf + Sigma + reg; factor
    for (int e = t; e < n * n; e += 64) {
      const int i = e / n, j = e - i * n;
      double v = Hf[i * ldn + j];
      if (i == j && !isnan(l[i])) {
        v += qd.reg;
        if (l[i] > -INFINITY) v += zl[i] / (x[i] - l[i]);
        if (u[i] < INFINITY) v += zu[i] / (u[i] - x[i]);
      }
      Mc[i * ldn + j] = v;
    }
    __syncthreads();
    if (!chol_lds(Mc, n, ldn)) { st = HILO_STATUS_OTHER; break; }
    for (int e = t; e < n * m; e += 64) {
      const int i = e / m, c = e - i * m;
      X[i * ldm + c] = Af[c * ldn + i];
    }
    __syncthreads();
    trsm_lds(Mc, n, ldn, X, m, ldm);
    for (int e = t; e < m * m; e += 64) {
      const int a = e / m, c = e - a * m;
      double s = (a == c) ? qd.reg : 0.0;
      for (int i = 0; i < n; ++i) s += X[i * ldm + a] * X[i * ldm + c];
      S[a * ldm + c] = s;
    }
    __syncthreads();
    if (m > 0 && !chol_lds(S, m, ldm)) { st = HILO_STATUS_OTHER; break; }

    double sigma_mu = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
      // r1 = base (+ centering/corrector terms in the second pass)
      for (int i = t; i < n; i += 64) {
        double s = base[i];
        if (pass == 1 && !isnan(l[i])) {
          if (l[i] > -INFINITY) s += (sigma_mu - dx[i] * dzl[i]) / (x[i] - l[i]);
          if (u[i] < INFINITY) s -= (sigma_mu + dx[i] * dzu[i]) / (u[i] - x[i]);
        }
        r1[i] = s;
      }
      __syncthreads();
      trsv_lds(Mc, n, ldn, r1, false);                       // t = L^-1 r1
      for (int a = t; a < m; a += 64) {
        double s = rp[a];
        for (int i = 0; i < n; ++i) s += X[i * ldm + a] * r1[i];
        dy[a] = s;
      }
      __syncthreads();
      if (m > 0) {
        trsv_lds(S, m, ldm, dy, false);
        trsv_lds(S, m, ldm, dy, true);
      }
      for (int i = t; i < n; i += 64) {
        double s = r1[i];
        for (int a = 0; a < m; ++a) s -= X[i * ldm + a] * dy[a];
        r1[i] = s;
      }
      __syncthreads();
      trsv_lds(Mc, n, ldn, r1, true);                        // dx = L^-T (t - X dy)
      // bound-multiplier steps, step lengths
      double ap = 1.0, ad = 1.0;
      const double tau = pass == 0 ? 1.0 : fmax(0.995, 1.0 - mu);
      for (int i = t; i < n; i += 64) {
        const bool fx = isnan(l[i]);
        const double d = fx ? 0.0 : r1[i];
        double dl = 0.0, du = 0.0;
        if (!fx) {
          // second pass: the affine products dx_aff*dz_aff are still in dx/dzl/dzu
          const double cl = pass == 1 ? dx[i] * dzl[i] : 0.0, cu = pass == 1 ? -dx[i] * dzu[i] : 0.0;
          if (l[i] > -INFINITY) {
            const double s = x[i] - l[i];
            dl = (sigma_mu - cl) / s - zl[i] - zl[i] / s * d;
            if (d < 0.0) ap = fmin(ap, -tau * s / d);
            if (dl < 0.0) ad = fmin(ad, -tau * zl[i] / dl);
          }
          if (u[i] < INFINITY) {
            const double s = u[i] - x[i];
            du = (sigma_mu - cu) / s - zu[i] + zu[i] / s * d;
            if (d > 0.0) ap = fmin(ap, tau * s / d);
            if (du < 0.0) ad = fmin(ad, -tau * zu[i] / du);
          }
        }
        dx[i] = d; dzl[i] = dl; dzu[i] = du;  // own entries only: no cross-lane hazard
      }
      ap = wave_min(ap);
      ad = wave_min(ad);
      __syncthreads();
      if (pass == 0) {
        double mpart = 0.0;
        for (int i = t; i < n; i += 64) {
          if (isnan(l[i])) continue;
          if (l[i] > -INFINITY) mpart += (x[i] - l[i] + ap * dx[i]) * (zl[i] + ad * dzl[i]);
          if (u[i] < INFINITY) mpart += (u[i] - x[i] - ap * dx[i]) * (zu[i] + ad * dzu[i]);
        }
        const double mu_aff = wave_sum(mpart) / nb;
        const double sg = mu > 0.0 ? (mu_aff / mu) : 0.0;
        sigma_mu = sg * sg * sg * mu;
      } else {
        for (int i = t; i < n; i += 64) {
          double xi = x[i] + ap * dx[i];
          // near convergence (mu ~ 1e-13) the fraction-to-the-boundary step can round a slack to exactly zero: keep every
          // slack at least a few ulps wide so that z / slack stays finite
          if (l[i] > -INFINITY) xi = fmax(xi, l[i] + 4.0e-16 * fmax(1.0, fabs(l[i])));
          if (u[i] < INFINITY) xi = fmin(xi, u[i] - 4.0e-16 * fmax(1.0, fabs(u[i])));
          x[i] = xi;
          zl[i] += ad * dzl[i];
          zu[i] += ad * dzu[i];
        }
        for (int a = t; a < m; a += 64) y[a] += ad * dy[a];
        __syncthreads();
      }
    }
  }

  // ---- outputs (CasADi conic sign convention: H x + g + A^T lam_a + lam_x = 0) ----
  double fpart = 0.0;
  for (int i = t; i < n; i += 64) {
    const bool fx = isnan(l[i]);
    x[i] = fx ? xfix[i] : x[i];
  }
  __syncthreads();
  for (int i = t; i < n; i += 64) {
    double hx = 0.0, aty = 0.0;
    for (int j = 0; j < n; ++j) hx += H[i * n + j] * x[j];
    for (int r = 0; r < m; ++r) aty += A[r * n + i] * y[r];
    fpart += x[i] * (0.5 * hx + g[i]);
    x_out[b * n + i] = x[i];
    if (lam_x) lam_x[b * n + i] = isnan(l[i]) ? -(hx + g[i] + aty) : zu[i] - zl[i];
  }
  if (lam_a)
    for (int a = t; a < m; a += 64) lam_a[b * m + a] = y[a];
  fpart = wave_sum(fpart);
  if (t == 0) {
    f_out[b] = fpart;
    status[b] = st;
    iters[b] = it;
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 3: the same Mehrotra iteration with the FACTORISATIONS IN REGISTERS (n, m <= NP <= 64).
// The first kernel factors column by column through LDS (a store / barrier / load round trip per column and per substitution
// step: ~50 dependent LDS round trips per factorisation, 8 triangular solves per iteration) and loses to a CPU core on the
// 32-variable QP of BASELINE configuration 1.  Here
//   * lane i holds ROW i of the matrix (H + Sigma, then the Schur complement) in registers; the right-looking Cholesky
//     broadcasts the pivot and the multipliers with v_readlane - no memory on the critical path;
//   * the inverse factors are formed once per iteration: lane c forward-substitutes COLUMN c of L^-1 (lanes n .. n+m-1: the
//     columns of X = L^-1 A^T) entirely in registers against L in LDS (uniform-address reads), so that every later
//     triangular solve is a matrix-vector product with L^-1 (one pass, no per-column barriers);
//   * both Mehrotra passes reuse L^-1, X = L^-1 A^T, S = X^T X and S's inverse factor.
// Same algorithm, constants, start and termination as qp_solve_kernel (which stays the path for larger or workspace-resident QPs).
__device__ __forceinline__ double qp_read_lane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double qp_rsq(double x) {   // 1 / sqrt(x): v_rsq_f64 + two Newton steps
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  y = y * fma(-hx * y, y, 1.5);
  return y;
}

// this lane's column of L^-1 B: col[i] = (b[i] - sum_{j < i} L[i][j] col[j]) / L[i][i];  L (NP x NP lower, identity padded) and
// the reciprocal diagonal in LDS (uniform-address reads), the column in registers, the right-hand side entry by entry
// An LDS base address held in ONE vector register and opaque to the optimiser: the accesses of an unrolled block become
// `ds_read base offset:imm`, and reads of data that does not change between iterations (H, A) are not hoisted out of the iteration
// loop.  (Left to itself the compiler forms every address as a scalar outside the loop - 600 of them, spilled to vector-register
// lanes - and parks 120 hoisted doubles in AGPRs; under that register pressure the scheduler serialises every LDS read.)
using qp_lds_cd = const __attribute__((address_space(3))) double*;
__device__ __forceinline__ qp_lds_cd qp_vbase(const double* p) {
  qp_lds_cd q = (qp_lds_cd)p;
  asm volatile("" : "+v"(q));
  return q;
}

// a[K0 .. K1) must be in registers here (the compiler waits for the loads that produce them) - eight at a time
template <int K0, int K1, int N>
__device__ __forceinline__ void qp_pin(double (&a)[N]) {
  if constexpr (K1 - K0 >= 8) {
    asm volatile("" : "+v"(a[K0]), "+v"(a[K0 + 1]), "+v"(a[K0 + 2]), "+v"(a[K0 + 3]), "+v"(a[K0 + 4]), "+v"(a[K0 + 5]),
                 "+v"(a[K0 + 6]), "+v"(a[K0 + 7]));
    qp_pin<K0 + 8, K1>(a);
  } else if constexpr (K1 - K0 >= 1) {
    asm volatile("" : "+v"(a[K0]));
    qp_pin<K0 + 1, K1>(a);
  }
}
// Lane i holds row i of a symmetric positive definite NP x NP matrix (a smaller matrix is padded with the identity): in-place
// lower Cholesky factor, row i of L in lane i; `dinv` = reciprocal of this lane's diagonal entry.  Every array index is a constant:
// the rows stay in registers.  Column J per step, right-looking:
//   critical path   pivot = row J of lane J (v_readlane), l = row[J] / sqrt(pivot), and the one update the NEXT column waits for,
//                   row[J + 1] -= l * L[J + 1][J] (v_readlane);
//   everything else the column l goes through a 64-entry LDS buffer and comes back as broadcast reads, L[k][J] for k >= J + 2; those
//                   reads are in flight during the next column's critical path and their multiply-adds run after it
//                   (LDS operations of a wave complete in issue order: the buffer needs no second copy and no wait between the
//                   write and the reads).
using qp_lds_d = __attribute__((address_space(3))) double*;
template <int J, int NP>
__device__ __forceinline__ void qp_chol_cols(double (&row)[NP], double& dinv, bool& ok, qp_lds_cd cb, qp_lds_d mine, double lprev,
                                             double (&prev)[NP]) {
  if constexpr (J < NP) {
    const double piv = qp_read_lane(row[J], J);
    const bool pos = piv > 0.0;
    ok = ok && pos;
    const double rs = qp_rsq(pos ? piv : 1.0);
    const double lj = row[J] * rs;            // L[i][J] for the lanes i >= J (lane J: sqrt(pivot))
    row[J] = lj;
    dinv = (int)threadIdx.x == J ? rs : dinv;
    double nxt[NP];
    if constexpr (J + 1 < NP) {
      *mine = lj;
      row[J + 1] -= lj * qp_read_lane(lj, J + 1);
#pragma unroll
      for (int k = J + 2; k < NP; ++k) nxt[k] = cb[k];
    }
    asm volatile("");                          // (ends the scheduling region: the reads above are issued before what follows)
    if constexpr (J >= 1) {
      qp_pin<J + 1, NP>(prev);
#pragma unroll
      for (int k = J + 1; k < NP; ++k) row[k] -= lprev * prev[k];   // L[i][J-1] L[k][J-1]; meaningful for the lanes i >= k
    }
    qp_chol_cols<J + 1, NP>(row, dinv, ok, cb, mine, lj, nxt);
  }
}
template <int NP>
__device__ __forceinline__ bool qp_chol_rows(double (&row)[NP], double& dinv, double* colbuf) {
  bool ok = true;
  dinv = 1.0;
  double none[NP];
  qp_lds_d mine = (qp_lds_d)(colbuf + threadIdx.x);
  asm volatile("" : "+v"(mine));
  qp_chol_cols<0, NP>(row, dinv, ok, qp_vbase(colbuf), mine, 0.0, none);
  return ok;
}

// row I of the forward substitution; `cur` = L[I][0 .. I) already requested from LDS.  Software pipeline written out by hand: the
// reads of row I + 1 are issued (a statement with side effects ends the scheduling region, so they stay in front of it) before the
// multiply-adds of row I run.  (The compiler's own order issues one read, waits for it, uses it.)
template <int I, int NP, class Rhs>
__device__ __forceinline__ void qp_fsub_rows(qp_lds_cd L, int ld, qp_lds_cd dinv, Rhs& rhs, double (&col)[NP], double (&cur)[NP]) {
  if constexpr (I < NP) {
    qp_pin<0, I>(cur);
    double nxt[NP];
    if constexpr (I + 1 < NP) {
#pragma unroll
      for (int j = 0; j < I + 1; ++j) nxt[j] = L[(I + 1) * ld + j];
    }
    const double di = dinv[I];
    double s0 = rhs(I), s1 = 0.0;               // two chains: the latency of one multiply-add hides behind the other
    asm volatile("");
#pragma unroll
    for (int j = 0; j + 1 < I; j += 2) {
      s0 -= cur[j] * col[j];
      s1 -= cur[j + 1] * col[j + 1];
    }
    if constexpr (I % 2 == 1) s0 -= cur[I - 1] * col[I - 1];
    col[I] = (s0 + s1) * di;
    // pin the row here: without it the arithmetic is sunk behind the next barrier into the block that stores the column
    asm volatile("" : "+v"(col[I]));
    qp_fsub_rows<I + 1, NP>(L, ld, dinv, rhs, col, nxt);
  }
}
template <int NP, class Rhs>
__device__ __forceinline__ void qp_fsub_col(const double* Lg, int ld, const double* dinvg, Rhs rhs, double (&col)[NP]) {
  double cur[NP];
  qp_fsub_rows<0, NP>(qp_vbase(Lg), ld, qp_vbase(dinvg), rhs, col, cur);
}

// sum_j row[j] v[j] over a compile-time length (padded operands, zeros in the padding): every LDS read is issued before the
// first multiply-add (the empty statement with side effects ends the scheduling region)
template <int K>
__device__ __forceinline__ double qp_dot(const double* rowg, const double* vg) {
  static_assert(K % 2 == 0, "even padded length");
  const qp_lds_cd v = qp_vbase(vg), row = qp_vbase(rowg);
  double a[K], w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { a[j] = row[j]; w[j] = v[j]; }
  asm volatile("");
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int j = 0; j < K; j += 2) {
    s0 += a[j] * w[j];
    s1 += a[j + 1] * w[j + 1];
  }
  return s0 + s1;
}
// sum_i M[i][t] v[i]: a column walk (consecutive lanes read consecutive addresses)
template <int K>
__device__ __forceinline__ double qp_dot_col(const double* colg, int ld, const double* vg) {
  static_assert(K % 2 == 0, "even padded length");
  const qp_lds_cd v = qp_vbase(vg), col = qp_vbase(colg);
  double a[K], w[K];
#pragma unroll
  for (int i = 0; i < K; ++i) { a[i] = col[i * ld]; w[i] = v[i]; }
  asm volatile("");
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int i = 0; i < K; i += 2) {
    s0 += a[i] * w[i];
    s1 += a[i + 1] * w[i + 1];
  }
  return s0 + s1;
}
// row[c] += sum_i X[i][a] X[i][c]  (a = this lane's column; the rows of X broadcast): rows i, i + 1 per trip, the reads of the next
// row in flight while the current one is used
template <int NP, int MP>
__device__ __forceinline__ void qp_gram_row(const double* Xg, int ld, int a, double (&row)[MP]) {
  static_assert(NP % 2 == 0, "two rows per trip");
  qp_lds_cd xr = qp_vbase(Xg);
  double p[MP], q[MP], pa, qa;
#pragma unroll
  for (int c = 0; c < MP; ++c) p[c] = xr[c];
  pa = xr[a];
  for (int i = 0; i < NP; i += 2) {
    qp_pin<0, MP>(p);
    asm volatile("" : "+v"(pa));
#pragma unroll
    for (int c = 0; c < MP; ++c) q[c] = xr[ld + c];
    qa = xr[ld + a];
    asm volatile("");
#pragma unroll
    for (int c = 0; c < MP; ++c) row[c] += pa * p[c];
    const int nx = (i + 2 < NP) ? 2 * ld : 0;            // (the last trip re-reads a row it does not use)
    qp_pin<0, MP>(q);
    asm volatile("" : "+v"(qa));
#pragma unroll
    for (int c = 0; c < MP; ++c) p[c] = xr[nx + c];
    pa = xr[nx + a];
    asm volatile("");
#pragma unroll
    for (int c = 0; c < MP; ++c) row[c] += qa * q[c];
    xr += 2 * ld;
  }
}

// Developer build (-DHILO_QP_PROF, tools/dbg/qp_prof.py): clock ticks per section of the iteration, instance 0
#ifdef HILO_QP_PROF
__device__ long long g_qp_prof[16];
#define QP_TICK(i)                                                  \
  do {                                                              \
    const long long tk_ = (long long)__builtin_readcyclecounter();  \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_qp_prof[i] += tk_ - tq_; \
    tq_ = (long long)__builtin_readcyclecounter();                  \
  } while (0)
#define QP_TICK0() long long tq_ = (long long)__builtin_readcyclecounter()
#else
#define QP_TICK(i)
#define QP_TICK0()
#endif

template <int NP, int MP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void qp_solve_reg_kernel(
    QpDims qd, int64_t batch, const double* __restrict__ Hg, int64_t hs, const double* __restrict__ gg, int64_t gs,
    const double* __restrict__ Ag, int64_t as_, const double* __restrict__ lbx, const double* __restrict__ ubx, int64_t bs,
    const double* __restrict__ lba, const double* __restrict__ uba, int64_t bas, double* __restrict__ x_out,
    double* __restrict__ f_out, double* __restrict__ lam_a, double* __restrict__ lam_x, int32_t* __restrict__ status,
    int32_t* __restrict__ iters, const double* __restrict__ xpin, int npin, int64_t ps) {
  static_assert(NP <= 64 && MP <= 64 && MP <= NP, "one row / column per lane");
  extern __shared__ double lds[];
  constexpr int ldn = NP + 1, ldm = MP + 1;     // odd pitches: conflict-free row-per-lane and column walks
  const int n = qd.n, m = qd.m, t = threadIdx.x;
  const int64_t b = blockIdx.x;
  if (b >= batch) return;
  double* q = lds;
  auto take = [&](size_t k) { double* r = q; q += k; return r; };
  double* Hf = take((size_t)NP * ldn);  // H with fixed rows / columns replaced by the identity (rows / columns >= n: identity)
  double* Af = take((size_t)MP * ldn);  // A with fixed columns zeroed (rows >= m, columns >= n: zero)
  double* Lm = take((size_t)NP * ldn);  // H + Sigma -> L = its factor -> (in place) L^-1
  double* X = take((size_t)NP * ldm);   // L^-1 A^T
  double* Ls = take((size_t)MP * ldm);  // Schur complement X^T X -> its factor -> its inverse factor
  double *x = take(NP), *gf = take(NP), *l = take(NP), *u = take(NP), *zl = take(NP), *zu = take(NP), *r1 = take(NP),
         *dx = take(NP), *dzl = take(NP), *dzu = take(NP), *base = take(NP), *xfix = take(NP), *tv = take(NP), *dinv = take(NP);
  double *y = take(MP), *bf = take(MP), *rp = take(MP), *dy = take(MP), *wv = take(MP), *dinvs = take(MP);
  double* cbuf = take(64);              // column broadcast buffer of the factorisations
  const double* H = Hg + b * hs;
  const double* A = Ag + b * as_;
  const double* g = gg + b * gs;
  for (int e = t; e < 14 * NP + 6 * MP; e += 64) x[e] = 0.0;   // every vector, padding included (the padding stays zero)
  __syncthreads();

  // ---- load, substitute fixed variables (lbx == ubx): as qp_solve_kernel; padding = identity / zero ----
  int bad_rows = 0;
  if (t < NP) {
    const bool in = t < n;
    double lo = 0.0, up = 0.0;
    if (in) qp_bounds(lbx, ubx, bs, xpin, npin, ps, b, t, lo, up);
    const bool fx = lo == up;          // padding slots count as fixed at 0
    l[t] = fx ? NAN : lo;
    u[t] = up;
    xfix[t] = fx ? lo : 0.0;
  }
  for (int r = t; r < m; r += 64)
    if (lba[b * bas + r] != uba[b * bas + r]) bad_rows = 1;
  bad_rows = __any(bad_rows);
  __syncthreads();
  for (int e = t; e < NP * NP; e += 64) {
    const int i = e / NP, j = e - i * NP;
    const bool in = i < n && j < n;
    const bool fi = isnan(l[i]), fj = isnan(l[j]);
    Hf[i * ldn + j] = (!in || fi || fj) ? (i == j ? 1.0 : 0.0) : H[i * n + j];
  }
  for (int e = t; e < MP * NP; e += 64) {
    const int r = e / NP, j = e - r * NP;
    Af[r * ldn + j] = (r < m && j < n && !isnan(l[j])) ? A[r * n + j] : 0.0;
  }
  if (t < NP) {
    double s = 0.0;
    if (t < n) {
      s = g[t];
      for (int j = 0; j < n; ++j) s += H[t * n + j] * xfix[j];
    }
    gf[t] = isnan(l[t]) ? 0.0 : s;
  }
  if (t < MP) {
    double s = 0.0;
    if (t < m) {
      s = uba[b * bas + t];
      for (int j = 0; j < n; ++j) s -= A[t * n + j] * xfix[j];
    }
    bf[t] = s;
    y[t] = 0.0;
  }
  double nb_part = 0.0;
  if (t < NP) {
    const int i = t;
    const bool fx = isnan(l[i]);
    const bool hl = !fx && l[i] > -INFINITY, hu = !fx && u[i] < INFINITY;
    double xi = 0.0;
    if (hl && hu) xi = 0.5 * (l[i] + u[i]);
    else if (hl) xi = fmax(0.0, l[i] + 1.0);
    else if (hu) xi = fmin(0.0, u[i] - 1.0);
    x[i] = fx ? 0.0 : xi;
    zl[i] = hl ? 1.0 : 0.0;
    zu[i] = hu ? 1.0 : 0.0;
    nb_part += (hl ? 1.0 : 0.0) + (hu ? 1.0 : 0.0);
  }
  const double nb = fmax(1.0, qp_wave_sum(nb_part));
  double gmax = 0.0;
  if (t < NP) gmax = fabs(gf[t]);
  gmax = qp_wave_max(gmax);
  __syncthreads();

  // H (after the substitution of the fixed variables) diagonal - the LMPC's block of diagonal weights (mpc.py:2307-2330 with
  // diagonal Q, R, P): H + Sigma is diagonal as well, its factor and L^-1 A^T are one scaling per row, no factorisation
  bool hdiag;
  {
    double off = 0.0;
    const qp_lds_cd hrow = qp_vbase(Hf + (t < NP ? t : NP - 1) * ldn);
#pragma unroll
    for (int j = 0; j < NP; ++j) off = fmax(off, (j == (t < NP ? t : NP - 1)) ? 0.0 : fabs(hrow[j]));
    hdiag = __all((int)(off == 0.0));
  }

  int st = HILO_STATUS_MAXITER, it = 0;
  double phi_min = INFINITY;
  QP_TICK0();
  if (bad_rows) st = HILO_STATUS_OTHER;
  for (it = 0; !bad_rows && it < qd.max_iter; ++it) {
    QP_TICK(0);
    // residuals: base = -(Hf x + gf + Af^T y), rd = -base - zl + zu, rp = Af x - bf, mu
    double rdmax = 0.0, mupart = 0.0, nonfinite = 0.0;
    const int tr = t < NP ? t : NP - 1, tm = t < MP ? t : MP - 1;   // lanes beyond the padded sizes mirror the last row
    const double lag = gf[tr] + qp_dot<NP>(Hf + tr * ldn, x) + qp_dot_col<MP>(Af + tr, ldn, y);
    const double ax = qp_dot<NP>(Af + tm * ldn, x);
    if (t < m) rp[t] = ax - bf[t];
    if (t < n) {
      const int i = t;
      double s = lag;
      const bool fx = isnan(l[i]);
      if (fx) s = 0.0;
      base[i] = -s;
      rdmax = fabs(s - zl[i] + zu[i]);
      nonfinite += isfinite(s - zl[i] + zu[i]) ? 0.0 : 1.0;
      if (!fx) {
        if (l[i] > -INFINITY) mupart += (x[i] - l[i]) * zl[i];
        if (u[i] < INFINITY) mupart += (u[i] - x[i]) * zu[i];
      }
    }
    double rpmax = 0.0;
    if (t < m) {
      const double s = rp[t];
      rpmax = fabs(s);
      nonfinite += isfinite(s) ? 0.0 : 1.0;
    }
    rdmax = qp_wave_max(rdmax);
    rpmax = qp_wave_max(rpmax);
    const double mu = qp_wave_sum(mupart) / nb;
    nonfinite = qp_wave_sum(nonfinite);
    if (nonfinite > 0.0 || !isfinite(rdmax) || !isfinite(rpmax) || !isfinite(mu)) { st = HILO_STATUS_INFEASIBLE; break; }
    const double phi = fmax(fmax(rdmax / (1.0 + gmax), rpmax), mu);
    if (phi <= qd.tol) { st = HILO_STATUS_SOLVED; break; }
    // infeasible QP (the reference's qpOASES reports it; a predictor-corrector iteration diverges instead): the termination rule
    // of OOQP (Gertz & Wright, ACM TOMS 29, 2003) - the merit has grown to 1e4 times its smallest value so far
    phi_min = fmin(phi_min, phi);
    if (phi >= 1.0e4 * phi_min) { st = HILO_STATUS_INFEASIBLE; break; }

    QP_TICK(1);
    double dil = 0.0;                              // hdiag: this lane's entry of the diagonal L^-1
    if (hdiag) {
      double d = Hf[tr * ldn + tr];
      if (!isnan(l[tr])) {
        d += qd.reg;
        const double lo = l[tr], up = u[tr], xv = x[tr];
        if (lo > -INFINITY) d += zl[tr] / (xv - lo);
        if (up < INFINITY) d += zu[tr] / (up - xv);
      }
      if (!__all((int)(d > 0.0 && isfinite(d)))) { st = HILO_STATUS_OTHER; break; }
      dil = qp_rsq(d);
      if (t < NP) {
        const qp_lds_cd acol = qp_vbase(Af + t);
#pragma unroll
        for (int r = 0; r < MP; ++r) X[t * ldm + r] = acol[r * ldn] * dil;
      }
      __syncthreads();
    QP_TICK(2);
    } else {
    // ---- M = Hf + Sigma + reg: row i in lane i (lanes >= NP mirror the last row), factored in registers ----
    bool okf;
    {
      double row[NP];
      const qp_lds_cd hrow = qp_vbase(Hf + tr * ldn);
#pragma unroll
      for (int j = 0; j < NP; ++j) row[j] = hrow[j];
      double dg = 0.0;
      if (!isnan(l[tr])) {
        dg = qd.reg;
        const double lo = l[tr], up = u[tr], xv = x[tr];
        if (lo > -INFINITY) dg += zl[tr] / (xv - lo);
        if (up < INFINITY) dg += zu[tr] / (up - xv);
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) row[j] += (j == tr) ? dg : 0.0;
      double di;
      okf = qp_chol_rows<NP>(row, di, cbuf);
      if (t < NP) {
        dinv[t] = di;
#pragma unroll
        for (int j = 0; j < NP; ++j) Lm[t * ldn + j] = row[j];   // (entries above the diagonal: never read)
      }
    }
    if (!okf) { st = HILO_STATUS_OTHER; break; }
    __syncthreads();
    QP_TICK(2);
    // ---- columns of X = L^-1 A^T (MP of them) and of L^-1 (NP), 64 per pass; the pass with the columns of L^-1 runs last and
    // overwrites L once every lane has finished reading it ----
    constexpr int NCOL = NP + MP;
#pragma unroll
    for (int c0 = ((NCOL - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
      const int c = c0 + t;
      const bool isL = c < NP, isX = c >= NP && c < NCOL;
      const qp_lds_cd arow = qp_vbase(Af + (size_t)(isX ? c - NP : 0) * ldn);
      double col[NP];
      const double wx = isX ? 1.0 : 0.0, wl = isL ? 1.0 : 0.0;      // branch-free right-hand side: a row of Af or a unit vector
      qp_fsub_col<NP>(Lm, ldn, dinv, [&](int i) { return fma(arow[i], wx, i == c ? wl : 0.0); }, col);
      if (isX) {
#pragma unroll
        for (int i = 0; i < NP; ++i) X[i * ldm + (c - NP)] = col[i];
      }
      if (c0 == 0) {
        __syncthreads();
        if (isL) {
#pragma unroll
          for (int i = 0; i < NP; ++i) Lm[i * ldn + c] = col[i];   // (zeros above the diagonal)
        }
      }
    }
    __syncthreads();
    }
    QP_TICK(3);
    // ---- Schur complement S = X^T X + reg: row a accumulated in lane a (the rows of X broadcast), factored in registers ----
    bool oks;
    {
      double row[MP];
#pragma unroll
      for (int c = 0; c < MP; ++c) row[c] = (c == tm) ? (tm < m ? qd.reg : 1.0) : 0.0;    // padding rows: identity
      qp_gram_row<NP, MP>(X, ldm, tm, row);
      double di;
    QP_TICK(4);
      oks = qp_chol_rows<MP>(row, di, cbuf);
      if (t < MP) {
        dinvs[t] = di;
#pragma unroll
        for (int j = 0; j < MP; ++j) Ls[t * ldm + j] = row[j];
      }
    }
    if (!oks) { st = HILO_STATUS_OTHER; break; }
    __syncthreads();
    {
    QP_TICK(5);
      double col[MP];
      qp_fsub_col<MP>(Ls, ldm, dinvs, [&](int i) { return (i == tm) ? 1.0 : 0.0; }, col);
      __syncthreads();
      if (t < MP) {
#pragma unroll
        for (int i = 0; i < MP; ++i) Ls[i * ldm + t] = col[i];
      }
    }
    __syncthreads();

    QP_TICK(6);
    double sigma_mu = 0.0;
    for (int pass = 0; pass < 2; ++pass) {
      if (t < n) {
        const int i = t;
        double s = base[i];
        if (pass == 1 && !isnan(l[i])) {
          if (l[i] > -INFINITY) s += (sigma_mu - dx[i] * dzl[i]) / (x[i] - l[i]);
          if (u[i] < INFINITY) s -= (sigma_mu + dx[i] * dzu[i]) / (u[i] - x[i]);
        }
        r1[i] = s;
      }
      __syncthreads();
      {                                              // tv = L^-1 r1  (L^-1: zeros above the diagonal)
        const double s = hdiag ? dil * r1[tr] : qp_dot<NP>(Lm + tr * ldn, r1);
        if (t < n) tv[t] = s;
      }
      __syncthreads();
      {                                              // dy0 = rp + X^T tv
        const double s = qp_dot_col<NP>(X + tm, ldm, tv);
        if (t < m) dy[t] = rp[t] + s;
      }
      __syncthreads();
      {                                              // wv = Ls^-1 dy0
        const double s = qp_dot<MP>(Ls + tm * ldm, dy);
        if (t < m) wv[t] = s;
      }
      __syncthreads();
      {                                              // dy = Ls^-T wv
        const double s = qp_dot_col<MP>(Ls + tm, ldm, wv);
        __syncthreads();
        if (t < m) dy[t] = s;
      }
      __syncthreads();
      {                                              // tv <- tv - X dy
        const double s = qp_dot<MP>(X + tr * ldm, dy);
        if (t < n) tv[t] -= s;
      }
      __syncthreads();
      {                                              // dx = L^-T tv
        const double s = hdiag ? dil * tv[tr] : qp_dot_col<NP>(Lm + tr, ldn, tv);
        if (t < n) r1[t] = s;
      }
      __syncthreads();
      // bound-multiplier steps, step lengths (as qp_solve_kernel)
      double ap = 1.0, ad = 1.0;
      const double tau = pass == 0 ? 1.0 : fmax(0.995, 1.0 - mu);
      if (t < n) {
        const int i = t;
        const bool fx = isnan(l[i]);
        const double d = fx ? 0.0 : r1[i];
        double dl = 0.0, du = 0.0;
        if (!fx) {
          const double cl = pass == 1 ? dx[i] * dzl[i] : 0.0, cu = pass == 1 ? -dx[i] * dzu[i] : 0.0;
          if (l[i] > -INFINITY) {
            const double s = x[i] - l[i];
            dl = (sigma_mu - cl) / s - zl[i] - zl[i] / s * d;
            if (d < 0.0) ap = fmin(ap, -tau * s / d);
            if (dl < 0.0) ad = fmin(ad, -tau * zl[i] / dl);
          }
          if (u[i] < INFINITY) {
            const double s = u[i] - x[i];
            du = (sigma_mu - cu) / s - zu[i] + zu[i] / s * d;
            if (d > 0.0) ap = fmin(ap, tau * s / d);
            if (du < 0.0) ad = fmin(ad, -tau * zu[i] / du);
          }
        }
        dx[i] = d; dzl[i] = dl; dzu[i] = du;
      }
      ap = qp_wave_min(ap);
      ad = qp_wave_min(ad);
      __syncthreads();
      if (pass == 0) {
        double mpart = 0.0;
        if (t < n && !isnan(l[t])) {
          const int i = t;
          if (l[i] > -INFINITY) mpart += (x[i] - l[i] + ap * dx[i]) * (zl[i] + ad * dzl[i]);
          if (u[i] < INFINITY) mpart += (u[i] - x[i] - ap * dx[i]) * (zu[i] + ad * dzu[i]);
        }
        const double mu_aff = qp_wave_sum(mpart) / nb;
        const double sg = mu > 0.0 ? (mu_aff / mu) : 0.0;
        sigma_mu = sg * sg * sg * mu;
      } else {
        if (t < n) {
          const int i = t;
          double xi = x[i] + ap * dx[i];
          if (l[i] > -INFINITY) xi = fmax(xi, l[i] + 4.0e-16 * fmax(1.0, fabs(l[i])));
          if (u[i] < INFINITY) xi = fmin(xi, u[i] - 4.0e-16 * fmax(1.0, fabs(u[i])));
          x[i] = xi;
          zl[i] += ad * dzl[i];
          zu[i] += ad * dzu[i];
        }
        if (t < m) y[t] += ad * dy[t];
        __syncthreads();
      }
    }
    QP_TICK(7);
  }

  // ---- outputs (CasADi conic sign convention: H x + g + A^T lam_a + lam_x = 0) ----
  double fpart = 0.0;
  if (t < n) x[t] = isnan(l[t]) ? xfix[t] : x[t];
  __syncthreads();
  if (t < n) {
    const int i = t;
    double hx = 0.0, aty = 0.0;
    for (int j = 0; j < n; ++j) hx += H[i * n + j] * x[j];
    for (int r = 0; r < m; ++r) aty += A[r * n + i] * y[r];
    fpart += x[i] * (0.5 * hx + g[i]);
    x_out[b * n + i] = x[i];
    if (lam_x) lam_x[b * n + i] = isnan(l[i]) ? -(hx + g[i] + aty) : zu[i] - zl[i];
  }
  if (lam_a && t < m) lam_a[b * m + t] = y[t];
  fpart = qp_wave_sum(fpart);
  if (t == 0) {
    f_out[b] = fpart;
    status[b] = st;
    iters[b] = it;
  }
}

}  // namespace hilo

#include "hilo_qp_ocp.h"


using namespace hilo;

struct hilo_qp {
  int device, n, m;
  QpDims qd;
  size_t lds_bytes;      // working set per instance
  bool big;              // working set in global memory
  int fast_np, fast_mp;  // register-resident kernel qp_solve_reg_kernel<NP, MP> (n <= NP, m <= MP), 0 = the LDS-column kernel
  size_t fast_lds;
  double* ws;
  int64_t ws_batch;
  int ocp_nx, ocp_nu, ocp_N;   // stage structure declared with hilo_qp_set_stages (0: none): the Riccati kernel of hilo_qp_ocp.h
};

extern "C" int hilo_qp_create(int n, int m, int device, hilo_qp** out) {
  HILO_REQUIRE(out, "hilo_qp_create: NULL argument");
  HILO_REQUIRE(n >= 1 && m >= 0, "hilo_qp_create: need n >= 1, m >= 0 (got %d, %d)", n, m);
  hilo_qp* h = new hilo_qp();
  h->device = device; h->n = n; h->m = m;
  QpDims& q = h->qd;
  q.n = n; q.m = m;
  q.ldn = n | 1;                 // odd leading dimensions: conflict-free column walks in LDS
  q.ldm = (m > 0 ? m : 1) | 1;
  q.max_iter = 100; q.tol = 1e-12; q.reg = 1e-12;
  h->lds_bytes = sizeof(double) * ((size_t)2 * n * q.ldn + (size_t)m * q.ldn + (size_t)n * q.ldm + (size_t)(m > 0 ? m : 1) * q.ldm +
                                   12 * (size_t)n + 4 * (size_t)(m > 0 ? m : 1));
  h->big = h->lds_bytes > 160 * 1024;   // e.g. LMPC with nx=2, nu=1 beyond N = 22
  // small dense QPs (the LMPC of BASELINE configuration 1: n = 32, m = 20): factorisations in registers, padded dimensions
  h->fast_np = n <= 32 ? 32 : (n <= 64 ? 64 : 0);
  h->fast_mp = h->fast_np == 32 ? (m <= 24 ? 24 : (m <= 32 ? 32 : 0)) : (h->fast_np == 64 ? (m <= 48 ? 48 : (m <= 64 ? 64 : 0)) : 0);
  if (!h->fast_mp) h->fast_np = 0;
  {
    const size_t NP = h->fast_np, MP = h->fast_mp;
    h->fast_lds = sizeof(double) * (2 * NP * (NP + 1) + MP * (NP + 1) + NP * (MP + 1) + MP * (MP + 1) + 14 * NP + 6 * MP + 64);
  }
  if (h->fast_lds > 160 * 1024 || getenv("HILO_QP_LDS_COLUMNS")) h->fast_np = 0;
  h->ws = nullptr;
  h->ws_batch = 0;
  h->ocp_nx = h->ocp_nu = h->ocp_N = 0;
  *out = h;
  return HILO_OK;
}

extern "C" void hilo_qp_destroy(hilo_qp* h) {
  if (!h) return;
  if (h->ws) (void)hipFree(h->ws);
  delete h;
}

extern "C" int hilo_qp_set_options(hilo_qp* h, double tol, int max_iter) {
  HILO_REQUIRE(h, "hilo_qp_set_options: NULL handle");
  if (tol > 0) h->qd.tol = tol;
  if (max_iter > 0) h->qd.max_iter = max_iter;
  return HILO_OK;
}

// The QP has the LMPC's stage shape (hilo_qp_ocp.h; mpc.py:2198-2266 with a block-diagonal input block): hilo_qp_solve then
// takes its Newton steps by a Riccati recursion over the stages.  Sizes without an instantiation keep the dense kernels
// (returns HILO_OK either way; `*used` tells which).  N = 0 withdraws the declaration.
#define HILO_QP_OCP_SIZES(X) X(1, 1) X(2, 1) X(2, 2) X(3, 1) X(3, 2) X(4, 1) X(4, 2)
extern "C" int hilo_qp_set_stages(hilo_qp* h, int nx, int nu, int N, int* used) {
  HILO_REQUIRE(h, "hilo_qp_set_stages: NULL handle");
  if (used) *used = 0;
  h->ocp_nx = h->ocp_nu = h->ocp_N = 0;
  if (N == 0) return HILO_OK;
  HILO_REQUIRE(nx >= 1 && nu >= 1 && N >= 1, "hilo_qp_set_stages: need nx, nu, N >= 1");
  HILO_REQUIRE(h->n == (N + 1) * nx + N * nu && h->m == N * nx,
               "hilo_qp_set_stages: n = %d, m = %d do not match (N+1) nx + N nu = %d, N nx = %d", h->n, h->m,
               (N + 1) * nx + N * nu, N * nx);
  bool have = false;
#define X(NXV, NUV) have = have || (nx == NXV && nu == NUV);
  HILO_QP_OCP_SIZES(X)
#undef X
  if (!have || N + 1 > 64 || getenv("HILO_QP_DENSE")) return HILO_OK;
  h->ocp_nx = nx; h->ocp_nu = nu; h->ocp_N = N;
  if (used) *used = 1;
  return HILO_OK;
}

extern "C" int hilo_qp_solve(hilo_qp* h, int64_t batch, const double* H, int64_t h_stride, const double* g,
                             int64_t g_stride, const double* A, int64_t a_stride, const double* lbx, const double* ubx,
                             int64_t bx_stride, const double* lba, const double* uba, int64_t ba_stride, double* x,
                             double* f, double* lam_a, double* lam_x, int32_t* status, int32_t* iters, void* stream) {
  return hilo_qp_solve_pinned(h, batch, H, h_stride, g, g_stride, A, a_stride, lbx, ubx, bx_stride, nullptr, 0, 0, lba, uba, ba_stride, x,
                              f, lam_a, lam_x, status, iters, stream);
}

extern "C" int hilo_qp_solve_pinned(hilo_qp* h, int64_t batch, const double* H, int64_t h_stride, const double* g,
                                    int64_t g_stride, const double* A, int64_t a_stride, const double* lbx, const double* ubx,
                                    int64_t bx_stride, const double* xpin, int npin, int64_t ps, const double* lba,
                                    const double* uba, int64_t ba_stride, double* x, double* f, double* lam_a, double* lam_x,
                                    int32_t* status, int32_t* iters, void* stream) {
  HILO_REQUIRE(h, "hilo_qp_solve: NULL handle");
  HILO_REQUIRE(batch >= 0, "hilo_qp_solve: negative batch");
  if (batch == 0) return HILO_OK;
  HILO_REQUIRE(H && g && lbx && ubx && x && f && status && iters, "hilo_qp_solve: NULL argument");
  HILO_REQUIRE(h->m == 0 || (A && lba && uba), "hilo_qp_solve: the problem has %d rows but A / lba / uba is NULL", h->m);
  HILO_REQUIRE(bx_stride >= h->n || (xpin && bx_stride == 0),
               "hilo_qp_solve: lbx/ubx are per instance (x_0 is pinned through them, mpc.py:2361-2362) unless pinned values are given");
  HILO_REQUIRE(!xpin || (npin > 0 && npin <= h->n && ps >= npin), "hilo_qp_solve_pinned: %d pinned variables with stride %lld", npin,
               (long long)ps);
  if (!xpin) { npin = 0; ps = 0; }
  HILO_HIP_CHECK(hipSetDevice(h->device));
  if (h->ocp_N) {
    const int N = h->ocp_N;
#define X(NXV, NUV)                                                                                                                 \
  if (h->ocp_nx == NXV && h->ocp_nu == NUV) {                                                                                        \
    if (N + 1 <= 16)                                                                                                                 \
      hipLaunchKernelGGL((qp_ocp_kernel<NXV, NUV, 16>), dim3((unsigned)((batch + 3) / 4)), dim3(64), 0, (hipStream_t)stream, h->qd,  \
                         N, batch, H, h_stride, g, g_stride, A, a_stride, lbx, ubx, bx_stride, lba, uba, ba_stride, x, f, lam_a,     \
                         lam_x, status, iters, xpin, npin, ps);                                                                     \
    else                                                                                                                             \
      hipLaunchKernelGGL((qp_ocp_kernel<NXV, NUV, 64>), dim3((unsigned)batch), dim3(64), 0, (hipStream_t)stream, h->qd, N, batch,    \
                         H, h_stride, g, g_stride, A, a_stride, lbx, ubx, bx_stride, lba, uba, ba_stride, x, f, lam_a, lam_x,        \
                         status, iters, xpin, npin, ps);                                                                            \
  }
    HILO_QP_OCP_SIZES(X)
#undef X
    HILO_HIP_CHECK(hipGetLastError());
    return HILO_OK;
  }
  if (h->fast_np) {
#define HILO_QP_FAST(NPV, MPV)                                                                                                     \
  if (h->fast_np == NPV && h->fast_mp == MPV) {                                                                                     \
    if (h->fast_lds > 64 * 1024)                                                                                                    \
      HILO_HIP_CHECK(hipFuncSetAttribute((const void*)qp_solve_reg_kernel<NPV, MPV>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                         (int)h->fast_lds));                                                                       \
    hipLaunchKernelGGL((qp_solve_reg_kernel<NPV, MPV>), dim3((unsigned)batch), dim3(64), h->fast_lds, (hipStream_t)stream, h->qd,   \
                       batch, H, h_stride, g, g_stride, A, a_stride, lbx, ubx, bx_stride, lba, uba, ba_stride, x, f, lam_a, lam_x, \
                       status, iters, xpin, npin, ps);                                                                            \
  }
    HILO_QP_FAST(32, 24)
    HILO_QP_FAST(32, 32)
    HILO_QP_FAST(64, 48)
    HILO_QP_FAST(64, 64)
#undef HILO_QP_FAST
    HILO_HIP_CHECK(hipGetLastError());
    return HILO_OK;
  }
  if (h->big) {
    if (h->ws_batch != batch) {
      if (h->ws) HILO_HIP_CHECK(hipFree(h->ws));
      h->ws = nullptr;
      hipError_t e = hipMalloc((void**)&h->ws, h->lds_bytes * (size_t)batch);
      if (e != hipSuccess) return fail(HILO_ENOMEM, "QP workspace (%zu B per instance): %s", h->lds_bytes, hipGetErrorString(e));
      h->ws_batch = batch;
    }
  } else if (h->lds_bytes > 64 * 1024) {
    HILO_HIP_CHECK(hipFuncSetAttribute((const void*)qp_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)h->lds_bytes));
  }
  hipLaunchKernelGGL(qp_solve_kernel, dim3((unsigned)batch), dim3(64), h->big ? 0 : h->lds_bytes, (hipStream_t)stream, h->qd,
                     batch, H, h_stride, g, g_stride, A, a_stride, lbx, ubx, bx_stride, lba, uba, ba_stride, x, f, lam_a, lam_x,
                     status, iters, h->big ? h->ws : (double*)nullptr, (int64_t)(h->lds_bytes / sizeof(double)), xpin, npin, ps);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

#ifdef HILO_QPO_PROF
extern "C" int hilo_debug_qpo_prof(unsigned long long* out) {
  HILO_HIP_CHECK(hipDeviceSynchronize());
  HILO_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(hilo::hilo_qpo_prof), sizeof(unsigned long long) * 16));
  return HILO_OK;
}
#endif
#ifdef HILO_QP_PROF
extern "C" int hilo_qp_debug_prof(long long* out16) {
  HILO_HIP_CHECK(hipDeviceSynchronize());
  HILO_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(hilo::g_qp_prof), sizeof(long long) * 16));
  long long z[16] = {0};
  HILO_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(hilo::g_qp_prof), z, sizeof(z)));
  return HILO_OK;
}
#endif
