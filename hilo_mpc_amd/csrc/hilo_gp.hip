// Exact Gaussian-process inference and prediction on the GPU + C ABI.
//
// Reference semantics (hilo_mpc/modules/machine_learning/gp/):
//   kernel.py:97-205   Kernel.__call__ / get_covariance_matrix  (X is n_features x n_obs; hyper-parameters as logs)
//   mean.py:90-116     Mean.__call__
//   inference.py:172-221 ExactInference.get_posterior (no jitter; upper Cholesky; alpha; LML; mean; var)
//   gp.py:699-718      GaussianProcess.predict (var += sn2 unless noise_free)
// The reference re-evaluates kernel matrix, Cholesky and both solves on EVERY predict call (gp.py:709); here
// K + sn2 I is factored once at create() on the device and kept resident together with alpha and L^-1, so a
// query costs one pass over X_train (k*), a dot product with alpha and, for the variance, one triangular
// matrix-vector product with L^-1 staged per query tile in LDS.
#include <math.h>

#include <stdlib.h>
#include <string.h>

#include "hilo_common.h"

namespace hilo {

// ------------------------------------------------------------------------------------------------
// kernel / mean program interpreter (opcodes: include/hilo_hip.h).  Programs are wave-uniform, so all
// control flow below is scalar.
// ------------------------------------------------------------------------------------------------
constexpr int GP_STACK = 8;
constexpr int GP_PACK_HDR = 5;  // = hilo_models.h GP2_HDR
constexpr int HILO_K_XX_BEGIN = 19;  // internal marker: evaluate following nodes at (x, x) until the POWER node

__device__ __forceinline__ double d2_active(const double* node, int na, const double* Mdiag, const double* x,
                                            int64_t sx, const double* xb, int64_t sxb) {
  double s = 0.0;
  for (int k = 0; k < na; ++k) {
    const int d = (int)node[2 + k];
    const double df = x[d * sx] - xb[d * sxb];
    s += df * Mdiag[k] * df;
  }
  return s;
}
__device__ __forceinline__ double dot_active(const double* node, int na, const double* a, int64_t sa,
                                             const double* b, int64_t sb) {
  double s = 0.0;
  for (int k = 0; k < na; ++k) {
    const int d = (int)node[2 + k];
    s += a[d * sa] * b[d * sb];
  }
  return s;
}

__device__ double eval_kernel(const double* __restrict__ prog, int len, const double* x, int64_t sx,
                              const double* xbar, int64_t sxbar) {
  double st[GP_STACK];
  int sp = 0;
  int pos = 0;
  int xx_depth = 0;
  while (pos < len) {
    const double* node = prog + pos;
    const int op = (int)node[0];
    const int na = (int)node[1];
    const int npar = (int)node[2 + na];
    const double* par = node + 3 + na;
    pos += 3 + na + npar;
    const double* xb = xx_depth ? x : xbar;
    const int64_t sxb = xx_depth ? sx : sxbar;
    double v = 0.0;
    switch (op) {
      case HILO_K_CONST:
        v = par[0];
        break;
      case HILO_K_GAMMAEXP: {  // kernel.py:696: exp(2 log s - alpha d2^(p/2))
        const double d2 = d2_active(node, na, par + 3, x, sx, xb, sxb);
        const double e = par[2] == 1.0 ? d2 : pow(d2, par[2]);
        v = par[0] * exp(-par[1] * e);
        break;
      }
      case HILO_K_MATERN: {  // kernel.py:810-821
        const int nc = (int)par[2];
        const double d = sqrt(par[1] * par[1] * d2_active(node, na, par + 3 + nc, x, sx, xb, sxb));
        double f = 1.0 + d * par[3];
        for (int k = 1; k < nc; ++k) f = 1.0 + d * par[3 + k] * f;
        v = par[0] * exp(-d) * f;
        break;
      }
      case HILO_K_RQ: {  // kernel.py:997
        const double d2 = d2_active(node, na, par + 2, x, sx, xb, sxb);
        v = par[0] * pow(1.0 + 0.5 * d2 / par[1], -par[1]);
        break;
      }
      case HILO_K_PP: {  // kernel.py:1076-1100
        const int q = (int)par[1];
        const double j = par[2];
        const double d2 = d2_active(node, na, par + 3, x, sx, xb, sxb);
        const double d = sqrt(d2);
        double f = 1.0;
        if (q == 1) f = (j + 1) * d + 1;
        else if (q == 2) f = (j * j + 4 * j + 3) / 3 * d2 + (j + 2) * d + 1;
        else if (q == 3)
          f = (j * j * j + 9 * j * j + 23 * j + 15) / 15 * d * d * d + (6 * j * j + 36 * j + 45) / 15 * d2 + (j + 3) * d + 1;
        const double cs = d < 1.0 ? pow(fmax(1.0 - d, 0.0), j + q) : 0.0;
        v = par[0] * cs * f;
        break;
      }
      case HILO_K_POLY: {  // kernel.py:1229
        const double b = dot_active(node, na, x, sx, xb, sxb) + par[1];
        const int deg = (int)par[2];
        double r = 1.0;
        for (int k = 0; k < deg; ++k) r *= b;
        v = par[0] * r;
        break;
      }
      case HILO_K_NN: {  // kernel.py:1320-1327
        const double num = 1.0 + dot_active(node, na, x, sx, xb, sxb);
        const double den1 = sqrt(par[1] + 1.0 + dot_active(node, na, x, sx, x, sx));
        const double den2 = sqrt(par[1] + 1.0 + dot_active(node, na, xb, sxb, xb, sxb));
        v = par[0] * asin(num / (den1 * den2));
        break;
      }
      case HILO_K_PERIODIC: {  // kernel.py:1413-1418 (one active dimension)
        const int d = (int)node[2];
        const double arg = sin(M_PI * (x[d * sx] - xb[d * sxb]) / par[2]) / par[1];
        v = exp(par[0] - 2.0 * arg * arg);
        break;
      }
      case HILO_K_XX_BEGIN:
        ++xx_depth;
        continue;
      case HILO_K_SUM:
        sp -= 2;
        v = st[sp] + st[sp + 1];
        break;
      case HILO_K_PRODUCT:
        sp -= 2;
        v = st[sp] * st[sp + 1];
        break;
      case HILO_K_POWER:  // kernel.py:1651-1660
        --sp;
        --xx_depth;
        v = pow(st[sp], par[0]);
        break;
      default:
        v = nan("");
    }
    st[sp++] = v;
  }
  return st[0];
}

__device__ double eval_mean(const double* __restrict__ prog, int len, const double* x, int64_t sx) {
  double st[GP_STACK];
  int sp = 0, pos = 0;
  while (pos < len) {
    const double* node = prog + pos;
    const int op = (int)node[0];
    const int na = (int)node[1];
    const int npar = (int)node[2 + na];
    const double* par = node + 3 + na;
    pos += 3 + na + npar;
    double v = 0.0;
    switch (op) {
      case HILO_M_CONST:
        v = par[0];
        break;
      case HILO_M_POLY: {  // mean.py:450: (M^T x + offset)^p
        double b = par[0];
        for (int k = 0; k < na; ++k) b += par[2 + k] * x[(int)node[2 + k] * sx];
        const int deg = (int)par[1];
        double r = 1.0;
        for (int k = 0; k < deg; ++k) r *= b;
        v = r;
        break;
      }
      case HILO_M_SUM:
        sp -= 2;
        v = st[sp] + st[sp + 1];
        break;
      case HILO_M_PRODUCT:
        sp -= 2;
        v = st[sp] * st[sp + 1];
        break;
      case HILO_M_POWER:
        --sp;
        v = pow(st[sp], par[0]);
        break;
      case HILO_M_SCALE:
        --sp;
        v = par[0] * st[sp];
        break;
      default:
        v = nan("");
    }
    st[sp++] = v;
  }
  return st[0];
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// K[i][j] = k(X1[:,i], X2[:,j]) (+ diag_add on the diagonal); feature-major inputs, unit-stride lanes along j
__global__ void kmat_kernel(const double* __restrict__ prog, int len, int nf, int64_t n1, const double* __restrict__ X1,
                            int64_t n2, const double* __restrict__ X2, double diag_add, double* __restrict__ K) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n2) return;
  for (int64_t i = blockIdx.y; i < n1; i += gridDim.y) {   // rows strided over gridDim.y (at most 65535 blocks in y)
    double v = eval_kernel(prog, len, X1 + i, n1, X2 + j, n2);
    if (i == j) v += diag_add;
    K[i * n2 + j] = v;
  }
}

__global__ void mean_kernel(const double* __restrict__ prog, int len, int64_t n, const double* __restrict__ X,
                            double* __restrict__ mu) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mu[i] = eval_mean(prog, len, X + i, n);
}

// Single-workgroup dense factorisation suite for the (setup-time) posterior, all in global memory (L2 resident):
//   A -> L (lower Cholesky, in place; the reference's upper factor is L^T), ym = y - m(X),
//   alpha = L^-T (L^-1 ym), lml, Linv = L^-1.
constexpr int FACT_TPB = 1024;
__global__ __launch_bounds__(FACT_TPB) void gp_factor_kernel(int n, double* __restrict__ A, const double* __restrict__ y,
                                                             const double* __restrict__ mu, double* __restrict__ alpha,
                                                             double* __restrict__ Linv, double* __restrict__ out /*[2]: lml, info*/) {
  const int t = threadIdx.x;
  __shared__ double red[FACT_TPB];
  __shared__ int bad;
  if (t == 0) bad = 0;
  __syncthreads();
  // --- right-looking Cholesky ---------------------------------------------------------------
  for (int j = 0; j < n; ++j) {
    const double ajj = A[(int64_t)j * n + j];
    if (t == 0 && !(ajj > 0.0)) bad = j + 1;
    const double d = sqrt(ajj);
    const double id = 1.0 / d;
    __syncthreads();
    for (int i = j + t; i < n; i += FACT_TPB) A[(int64_t)i * n + j] = (i == j) ? d : A[(int64_t)i * n + j] * id;
    __syncthreads();
    // trailing update of the lower triangle: A[i][k] -= L[i][j] L[k][j], j < k <= i
    const int m = n - j - 1;
    const int64_t tot = (int64_t)m * (m + 1) / 2;
    for (int64_t e = t; e < tot; e += FACT_TPB) {
      // e -> (r, c) with 0 <= c <= r < m (row-major packed lower)
      int r = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
      while ((int64_t)r * (r + 1) / 2 > e) --r;
      while ((int64_t)(r + 1) * (r + 2) / 2 <= e) ++r;
      const int c = (int)(e - (int64_t)r * (r + 1) / 2);
      const int i = j + 1 + r, k = j + 1 + c;
      A[(int64_t)i * n + k] -= A[(int64_t)i * n + j] * A[(int64_t)k * n + j];
    }
    __syncthreads();
  }
  // --- alpha = L^-T L^-1 (y - mu) (inference.py:207-208) ---------------------------------------------
  for (int i = t; i < n; i += FACT_TPB) alpha[i] = y[i] - mu[i];
  __syncthreads();
  double quad = 0.0;  // (y-m)^T alpha accumulated as |L^-1 ym|^2 is NOT what the reference computes; see below
  for (int j = 0; j < n; ++j) {  // forward: column oriented
    if (t == 0) alpha[j] /= A[(int64_t)j * n + j];
    __syncthreads();
    const double aj = alpha[j];
    for (int i = j + 1 + t; i < n; i += FACT_TPB) alpha[i] -= A[(int64_t)i * n + j] * aj;
    __syncthreads();
  }
  for (int j = n - 1; j >= 0; --j) {  // backward with L^T: alpha_j /= L_jj; alpha_i -= L[j][i] alpha_j for i < j
    if (t == 0) alpha[j] /= A[(int64_t)j * n + j];
    __syncthreads();
    const double aj = alpha[j];
    for (int i = t; i < j; i += FACT_TPB) alpha[i] -= A[(int64_t)j * n + i] * aj;
    __syncthreads();
  }
  // --- LML = -1/2 (y-m) alpha - sum log diag L - n/2 log 2 pi (inference.py:210) -----------------------
  double part = 0.0;
  for (int i = t; i < n; i += FACT_TPB) part += -0.5 * (y[i] - mu[i]) * alpha[i] - log(A[(int64_t)i * n + i]);
  red[t] = part;
  __syncthreads();
  for (int s = FACT_TPB / 2; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  if (t == 0) {
    out[0] = red[0] - 0.5 * n * log(2.0 * M_PI);
    out[1] = (double)bad;
  }
  (void)quad;
  (void)Linv;
}

// Blocked version of the above for training sets whose panel fits the LDS (n <= 512): right-looking Cholesky in 16-wide block
// columns - diagonal block factored and inverted by one wave in LDS, panel L_ij = A_ij L_jj^-T by all threads into an LDS panel,
// trailing update A_IK -= P_I P_K^T tile by tile on v_mfma_f64_16x16x4 (4 per 16 x 16 x 16 tile, the waves of the workgroup
// share the tiles) - then alpha by blocked forward / backward substitution with the stored inverses of the diagonal blocks, and
// the log marginal likelihood.  13 block steps instead of 200 column steps (3 barriers each) for n = 200.
typedef double gpf_v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(FACT_TPB) void gp_factor_blocked_kernel(int n, double* __restrict__ A, const double* __restrict__ y,
                                                                     const double* __restrict__ mu, double* __restrict__ alpha,
                                                                     double* __restrict__ out /*[2]: lml, info*/) {
  extern __shared__ double fsm[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwave = FACT_TPB / 64;
  const int T = (n + 15) / 16, np_ = T * 16;
  double* D = fsm;                       // [16][17] diagonal block (then its Cholesky factor)
  double* Dinv = D + 16 * 17;            // [T][16][16] inverses of the diagonal factors (lower triangular, row-major)
  double* P = Dinv + (size_t)T * 256;    // [np_][16] panel of the current block column (rows below the diagonal block)
  double* r = P + (size_t)np_ * 16;      // [np_] right-hand side of the alpha solves
  __shared__ int bad;
  __shared__ double red[FACT_TPB / 64];
  if (t == 0) bad = 0;
  __syncthreads();
  for (int J = 0; J < T; ++J) {
    const int j0 = 16 * J;
    // (1) diagonal block: load, factor, invert - one wave, lane i < 16 owns row i
    if (wave == 0) {
      for (int e = lane; e < 256; e += 64) {
        const int i = e >> 4, c = e & 15, gi = j0 + i, gc = j0 + c;
        D[i * 17 + c] = (gi < n && gc < n) ? A[(int64_t)gi * n + gc] : (i == c ? 1.0 : 0.0);   // identity padding
      }
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      for (int c = 0; c < 16; ++c) {
        const double acc = D[c * 17 + c];
        if (lane == 0 && !(acc > 0.0) && bad == 0) bad = j0 + c + 1;
        const double d = sqrt(acc > 0.0 ? acc : 1.0), id = 1.0 / d;
        __builtin_amdgcn_wave_barrier();
        if (lane < 16 && lane >= c) D[lane * 17 + c] = lane == c ? d : D[lane * 17 + c] * id;
        __builtin_amdgcn_wave_barrier();
        if (lane < 16 && lane > c) {
          const double lic = D[lane * 17 + c];
          for (int k = c + 1; k <= lane; ++k) D[lane * 17 + k] -= lic * D[k * 17 + c];
        }
        __builtin_amdgcn_wave_barrier();
      }
      // inverse of the lower-triangular factor: lane c < 16 builds column c by forward substitution
      double* Di = Dinv + (size_t)J * 256;
      if (lane < 16) {
        const int c = lane;
        double col[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          double sv = i == c ? 1.0 : 0.0;
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (k < i && k >= c) sv -= D[i * 17 + k] * col[k];
          col[i] = i >= c ? sv / D[i * 17 + i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) Di[i * 16 + c] = col[i];
      }
      for (int e = lane; e < 256; e += 64) {   // the factor back to global memory
        const int i = e >> 4, c = e & 15, gi = j0 + i, gc = j0 + c;
        if (gi < n && gc < n && c <= i) A[(int64_t)gi * n + gc] = D[i * 17 + c];
      }
    }
    __syncthreads();
    // (2) panel: L_ij = A_ij L_jj^-T for the rows below the block  (entry (row, c) = sum_{k <= c} A[row][j0 + k] Linv[c][k])
    const int r0 = j0 + 16, nrow = np_ - r0;
    {
      const double* Di = Dinv + (size_t)J * 256;
      for (int e = t; e < nrow * 16; e += FACT_TPB) {
        const int rr = e >> 4, c = e & 15, gi = r0 + rr;
        double sv = 0.0;
        if (gi < n) {
          for (int k = 0; k <= c; ++k) {
            const int gk = j0 + k;
            if (gk < n) sv += A[(int64_t)gi * n + gk] * Di[c * 16 + k];
          }
        }
        P[rr * 16 + c] = sv;
      }
    }
    __syncthreads();
    for (int e = t; e < nrow * 16; e += FACT_TPB) {
      const int rr = e >> 4, c = e & 15, gi = r0 + rr, gc = j0 + c;
      if (gi < n && gc < n) A[(int64_t)gi * n + gc] = P[rr * 16 + c];
    }
    // (3) trailing update on the matrix cores: tiles (I >= K) of the rows / columns below the block
    const int TT = nrow / 16, npair = TT * (TT + 1) / 2;
    const int q = lane & 15, g = lane >> 4;
    for (int pr = wave; pr < npair; pr += nwave) {
      int I = (int)((sqrt(8.0 * (double)pr + 1.0) - 1.0) * 0.5);
      while (I * (I + 1) / 2 > pr) --I;
      while ((I + 1) * (I + 2) / 2 <= pr) ++I;
      const int K = pr - I * (I + 1) / 2;
      gpf_v4d acc;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int gi = r0 + 16 * I + 4 * rg + g, gc = r0 + 16 * K + q;
        acc[rg] = (gi < n && gc < n) ? A[(int64_t)gi * n + gc] : 0.0;
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const double a = -P[(16 * I + q) * 16 + 4 * kb + g];      // A operand: -P_I[row q][4 kb + g]
        const double b = P[(16 * K + q) * 16 + 4 * kb + g];       // B operand: P_K^T[4 kb + g][col q]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int gi = r0 + 16 * I + 4 * rg + g, gc = r0 + 16 * K + q;
        if (gi < n && gc < n && gc <= gi) A[(int64_t)gi * n + gc] = acc[rg];
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  // --- alpha = L^-T L^-1 (y - mu) (inference.py:207-208), blocked ------------------------------------------------
  for (int i = t; i < np_; i += FACT_TPB) r[i] = i < n ? y[i] - mu[i] : 0.0;
  __syncthreads();
  for (int J = 0; J < T; ++J) {          // forward: z_J = L_JJ^-1 r_J;  r_I -= L_IJ z_J for I > J
    const int j0 = 16 * J;
    const double* Di = Dinv + (size_t)J * 256;
    double zv = 0.0;
    if (t < 16) {
      for (int k = 0; k <= t; ++k) zv += Di[t * 16 + k] * r[j0 + k];
    }
    __syncthreads();
    if (t < 16) r[j0 + t] = zv;
    __syncthreads();
    for (int i = j0 + 16 + t; i < n; i += FACT_TPB) {
      double sv = 0.0;
      for (int k = 0; k < 16; ++k)
        if (j0 + k < n) sv += A[(int64_t)i * n + j0 + k] * r[j0 + k];
      r[i] -= sv;
    }
    __syncthreads();
  }
  for (int J = T - 1; J >= 0; --J) {     // backward: x_J = L_JJ^-T r_J;  r_I -= L_JI^T x_J for I < J
    const int j0 = 16 * J;
    const double* Di = Dinv + (size_t)J * 256;
    double xv = 0.0;
    if (t < 16) {
      for (int k = t; k < 16; ++k) xv += Di[k * 16 + t] * r[j0 + k];
    }
    __syncthreads();
    if (t < 16) r[j0 + t] = xv;
    __syncthreads();
    for (int i = t; i < j0; i += FACT_TPB) {
      double sv = 0.0;
      for (int k = 0; k < 16; ++k)
        if (j0 + k < n) sv += A[(int64_t)(j0 + k) * n + i] * r[j0 + k];
      r[i] -= sv;
    }
    __syncthreads();
  }
  // --- LML = -1/2 (y-m) alpha - sum log diag L - n/2 log 2 pi (inference.py:210) -----------------------
  double part = 0.0;
  for (int i = t; i < n; i += FACT_TPB) {
    alpha[i] = r[i];
    part += -0.5 * (y[i] - mu[i]) * r[i] - log(A[(int64_t)i * n + i]);
  }
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  if (t == 0) {
    double tot = 0.0;
    for (int w = 0; w < nwave; ++w) tot += red[w];
    out[0] = tot - 0.5 * n * log(2.0 * M_PI);
    out[1] = (double)bad;
  }
}

// Linv = L^-1, needed by the predictive variance only (built on first use after a fit): one wave per column c, forward
// substitution with the dot products spread over the lanes; row-major full matrix, zeros above the diagonal
__global__ __launch_bounds__(64) void gp_linv_kernel(int n, const double* __restrict__ L, double* __restrict__ Linv, int lp) {
  extern __shared__ double col[];   // column c of L^-1 while it is built
  const int c = blockIdx.x, lane = threadIdx.x;
  for (int i = c; i < n; ++i) {
    double s = 0.0;
    for (int j = c + lane; j < i; j += 64) s += L[(int64_t)i * n + j] * col[j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) col[i] = ((i == c ? 1.0 : 0.0) - s) / L[(int64_t)i * n + i];
    __syncthreads();
  }
  for (int i = lane; i < n; i += 64) Linv[(int64_t)i * lp + c] = i < c ? 0.0 : col[i];   // row pitch lp, zero padded
}

// L^-1 in MFMA-operand order for the register-resident predict kernel: block (row tile it, k-block kb <= last block of the
// tile) holds its 64 A-operand values in lane order, lane 16 g + q <-> L^-1[16 it + q][4 kb + g]; blocks of one row tile are
// consecutive, tile it starts at block 2 it (it + 1).  One wave-wide load of an operand is then 512 contiguous bytes (four
// cache lines) instead of sixteen 32-byte pieces of sixteen rows.
__global__ void gp_linv_swizzle_kernel(int nt, const double* __restrict__ Linv, int lp, double* __restrict__ Ls) {
  const int blk = blockIdx.x, lane = threadIdx.x, q = lane & 15, g = lane >> 4;
  int it = 0;
  while (2 * (it + 1) * (it + 2) <= blk) ++it;
  const int kb = blk - 2 * it * (it + 1);
  if (it < nt) Ls[(int64_t)blk * 64 + lane] = Linv[(int64_t)(16 * it + q) * lp + 4 * kb + g];
}

// d LML / d theta_j = 1/2 tr((alpha alpha^T - K_y^-1) dK_y / d theta_j) (Rasmussen & Williams eq. 5.9): the factorisation of
// the handle is reused for every hyper-parameter, dK_y/d theta_j comes from central differences of the (cheap) covariance
// function itself - the two perturbed kernel programs of each theta_j are evaluated element by element.
//   Amat[a][b] = alpha_a alpha_b - sum_k Linv[k][a] Linv[k][b]
__global__ void gp_amat_kernel(int n, const double* __restrict__ alpha, const double* __restrict__ Linv, int lp, double* __restrict__ Amat) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * n) return;
  const int a = e / n, b = e - a * n;
  double s = 0.0;
  for (int k = a > b ? a : b; k < n; ++k) s += Linv[(int64_t)k * lp + a] * Linv[(int64_t)k * lp + b];
  Amat[e] = alpha[a] * alpha[b] - s;
}
__global__ __launch_bounds__(256) void gp_grad_kernel(int n, int nf, int klen, const double* __restrict__ progs /*[nt][2][klen]*/,
                                                      const double* __restrict__ dsn2 /*[nt]: d sn2 / d theta*/,
                                                      const double* __restrict__ inv2h /*[nt]*/, const double* __restrict__ X,
                                                      const double* __restrict__ Amat, double* __restrict__ grad) {
  const int j = blockIdx.y;
  const double* pp = progs + (int64_t)j * 2 * klen;
  __shared__ double red[256];
  double part = 0.0;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n * n; e += gridDim.x * 256) {
    const int a = e / n, b = e - a * n;
    double dk = (eval_kernel(pp, klen, X + a, n, X + b, n) - eval_kernel(pp + klen, klen, X + a, n, X + b, n)) * inv2h[j];
    if (a == b) dk += dsn2[j];
    part += Amat[e] * dk;
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(grad + j, 0.5 * red[0]);
}

// Prediction for a tile of Q queries per workgroup (gp.py:699-718, inference.py:212-217).
//   phase 1: K*[i][q] = k(X[:,i], x_q) into LDS
//   phase 2: mean_q = m(x_q) + sum_i K*[i][q] alpha_i
//   phase 3: var_q = k(x_q,x_q) - |Linv K*[:,q]|^2 (+ sn2)
constexpr int PRED_TPB = 256;
__global__ __launch_bounds__(PRED_TPB) void gp_predict_kernel(const double* __restrict__ kprog, int klen,
                                                              const double* __restrict__ mprog, int mlen, int nf, int n,
                                                              const double* __restrict__ Xt, const double* __restrict__ alpha,
                                                              const double* __restrict__ Linv, int lp, double sn2_add, int64_t m,
                                                              const double* __restrict__ Xq, int Q, double* __restrict__ mean,
                                                              double* __restrict__ var) {
  extern __shared__ double lds[];  // K* tile [n][Q] then reduction scratch [PRED_TPB]
  double* Ks = lds;
  double* red = lds + (int64_t)n * Q;
  const int t = threadIdx.x;
  const int64_t q0 = (int64_t)blockIdx.x * Q;
  const int nq = (int)((m - q0) < Q ? (m - q0) : Q);
  for (int e = t; e < n * Q; e += PRED_TPB) {
    const int i = e / Q, q = e - i * Q;
    Ks[e] = q < nq ? eval_kernel(kprog, klen, Xt + i, n, Xq + q0 + q, m) : 0.0;
  }
  __syncthreads();
  const int q = t % Q, g = t / Q, G = PRED_TPB / Q;
  {
    double s = 0.0;
    for (int i = g; i < n; i += G) s += Ks[i * Q + q] * alpha[i];
    red[t] = s;
    __syncthreads();
    if (g == 0 && q < nq) {
      double tot = eval_mean(mprog, mlen, Xq + q0 + q, m);
      for (int k = 0; k < G; ++k) tot += red[k * Q + q];
      mean[q0 + q] = tot;
    }
    __syncthreads();
  }
  if (var) {
    double ss = 0.0;
    for (int i = g; i < n; i += G) {
      const double* Li = Linv + (int64_t)i * lp;
      double v = 0.0;
      for (int j = 0; j <= i; ++j) v += Li[j] * Ks[j * Q + q];
      ss += v * v;
    }
    red[t] = ss;
    __syncthreads();
    if (g == 0 && q < nq) {
      double tot = 0.0;
      for (int k = 0; k < G; ++k) tot += red[k * Q + q];
      const double kss = eval_kernel(kprog, klen, Xq + q0 + q, m, Xq + q0 + q, m);
      var[q0 + q] = kss - tot + sn2_add;
    }
  }
}


// Prediction on the f64 matrix cores: the variance needs V = L^-1 K* (n x n lower-triangular times n x Q), a GEMM whose
// column sums of squares are subtracted from k(x*, x*) (inference.py:215-217).  One workgroup = Q = 16 W queries, W waves;
// wave w owns the 16 query columns of tile w.
//   phase 1: K* tile [n_pad][Q] into LDS (rows >= n zero), mean = m(x*) + K*^T alpha
//   phase 2: per 16-row tile `it` of L^-1: acc (16 x 16) = sum over k-blocks of 4 <= the tile's last row (the factor is lower
//            triangular, later blocks are zero) of v_mfma_f64_16x16x4: A-operand L^-1[16 it + q][4 kb + g] straight from
//            global memory (L2-resident, one element per lane), B-operand K*[4 kb + g][16 w + q] from LDS; lane = 16 g + q
//            holds rows 4 r + g of column q in accumulator register r: ss_q += sum_r acc_r^2, combined over g at the end.
typedef double gp_v4d __attribute__((ext_vector_type(4)));
template <int W>
__global__ __launch_bounds__(64 * W) void gp_predict_mfma_kernel(const double* __restrict__ kprog, int klen,
                                                                 const double* __restrict__ mprog, int mlen, int nf, int n,
                                                                 const double* __restrict__ Xt, const double* __restrict__ alpha,
                                                                 const double* __restrict__ Linv, int lp, double sn2_add, int64_t m,
                                                                 const double* __restrict__ Xq, double* __restrict__ mean,
                                                                 double* __restrict__ var) {
  constexpr int Q = 16 * W, TPB = 64 * W;
  // row pitch Q + 16 doubles: the four row groups g of a B-operand read then fall on disjoint halves of the 64 LDS banks
  constexpr int P = Q + 16;
  extern __shared__ double lds[];   // K* tile [n_pad][P] | partial sums [TPB] | kernel program | X_train [nf][n_pad] | queries [nf][Q]
  const int n_pad = (n + 15) & ~15;
  double* Ks = lds;
  double* red = lds + (int64_t)(n_pad + 16) * P;   // 16 spare zero rows: the k-loop runs in groups of eight blocks
  double* prog_s = red + TPB;
  double* Xt_s = prog_s + klen;
  double* Xq_s = Xt_s + (int64_t)nf * n_pad;
  const int t = threadIdx.x;
  const int64_t q0 = (int64_t)blockIdx.x * Q;
  const int nq = (int)((m - q0) < Q ? (m - q0) : Q);
  // operands of the covariance function once into LDS: every one of the n_pad x Q evaluations below reads them from there
  for (int e = t; e < klen; e += TPB) prog_s[e] = kprog[e];
  for (int e = t; e < nf * n_pad; e += TPB) {
    const int d = e / n_pad, i = e - d * n_pad;
    Xt_s[e] = i < n ? Xt[(int64_t)d * n + i] : 0.0;
  }
  for (int e = t; e < nf * Q; e += TPB) {
    const int d = e / Q, q = e - d * Q;
    Xq_s[e] = q < nq ? Xq[(int64_t)d * m + q0 + q] : 0.0;
  }
  __syncthreads();
  // the common case - one squared-exponential node (kernel.py:696 with p = 2) - is evaluated inline; everything else goes
  // through the interpreter
  const int na0 = (int)prog_s[1];
  const bool se = (int)prog_s[0] == HILO_K_GAMMAEXP && klen == 3 + na0 + (int)prog_s[2 + na0] && prog_s[3 + na0 + 2] == 1.0;
  if (se) {
    const double sf2 = prog_s[3 + na0], al = prog_s[3 + na0 + 1];
    const double* Md = prog_s + 3 + na0 + 3;
    for (int e = t; e < (n_pad + 16) * Q; e += TPB) {
      const int i = e / Q, q = e - i * Q;
      double d2 = 0.0;
      for (int k = 0; k < na0; ++k) {
        const int d = (int)prog_s[2 + k];
        const double df = Xt_s[d * n_pad + (i < n_pad ? i : 0)] - Xq_s[d * Q + q];
        d2 += df * Md[k] * df;
      }
      Ks[i * P + q] = (i < n && q < nq) ? sf2 * exp(-al * d2) : 0.0;
    }
  } else {
    for (int e = t; e < (n_pad + 16) * Q; e += TPB) {
      const int i = e / Q, q = e - i * Q;
      Ks[i * P + q] = (i < n && q < nq) ? eval_kernel(prog_s, klen, Xt_s + i, n_pad, Xq_s + q, Q) : 0.0;
    }
  }
  __syncthreads();
  {  // mean: thread (g, q) sums rows g, g + G, ...
    const int q = t % Q, g = t / Q;
    constexpr int G = TPB / Q;
    double s = 0.0;
    for (int i = g; i < n; i += G) s += Ks[i * P + q] * alpha[i];
    red[t] = s;
    __syncthreads();
    if (g == 0 && q < nq) {
      double tot = eval_mean(mprog, mlen, Xq + q0 + q, m);
#pragma unroll
      for (int k = 0; k < G; ++k) tot += red[k * Q + q];
      mean[q0 + q] = tot;
    }
  }
  if (!var) return;
  const int lane = t & 63, w = t >> 6, q = lane & 15, g = lane >> 4;
  const int ntile = n_pad / 16;
  double ss = 0.0;
  const double* Kw = Ks + 16 * w + q;
  // L^-1 is stored zero padded (rows up to n_pad, row pitch lp = n_pad + 16) and the K* tile carries 16 spare zero rows, so the
  // k-loop needs no bounds: eight unconditional operand pairs per trip (A from global / L2, B from LDS), eight MFMAs.
  for (int it = 0; it < ntile; ++it) {
    gp_v4d acc = {0.0, 0.0, 0.0, 0.0};
    const double* Lrow = Linv + (int64_t)(16 * it + q) * lp + g;
    const double* Kg = Kw + g * P;
    const int nk = 16 * (it + 1);                  // columns up to the last row of this tile; the rest of the row is zero
    for (int k0 = 0; k0 < nk; k0 += 32) {
      double a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = Lrow[k0 + 4 * u];
        b[u] = Kg[(k0 + 4 * u) * P];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) ss += acc[r] * acc[r];
  }
  // combine the four row groups g of a column (lanes q, 16 + q, 32 + q, 48 + q)
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  if (g == 0 && 16 * w + q < nq) {
    const int64_t qi = q0 + 16 * w + q;
    const double kss = eval_kernel(prog_s, klen, Xq_s + 16 * w + q, Q, Xq_s + 16 * w + q, Q);
    var[qi] = kss - ss + sn2_add;
  }
}

// Register-resident variant for n <= 256 (the sizes the reference's exact GP is used at): lane (g, q) of a wave evaluates the
// entries K*[4 kb + g][q] of its 16 query columns itself and KEEPS them - they are exactly the B operands of every k-block -
// so the K* tile never exists in LDS, a workgroup needs only the staged training inputs there, and several workgroups share
// a CU (the MFMA chains of one hide the operand fetches of the others).  NT = row tiles of 16 (compile-time: the operand
// registers are indexed statically, both loops are fully unrolled).
template <int NT>
__global__ __launch_bounds__(256) void gp_predict_reg_kernel(const double* __restrict__ kprog, int klen,
                                                             const double* __restrict__ mprog, int mlen, int nf, int n,
                                                             const double* __restrict__ Xt, const double* __restrict__ alpha,
                                                             const double* __restrict__ Ls, double sn2_add, int64_t m,
                                                             const double* __restrict__ Xq, double* __restrict__ mean,
                                                             double* __restrict__ var) {
  constexpr int n_pad = 16 * NT, NKB = 4 * NT;
  extern __shared__ double lds[];   // kernel program | X_train [nf][n_pad] | alpha [n_pad]
  double* prog_s = lds;
  double* Xt_s = prog_s + klen;
  double* al_s = Xt_s + (int64_t)nf * n_pad;
  const int t = threadIdx.x;
  for (int e = t; e < klen; e += 256) prog_s[e] = kprog[e];
  for (int e = t; e < nf * n_pad; e += 256) {
    const int d = e / n_pad, i = e - d * n_pad;
    Xt_s[e] = i < n ? Xt[(int64_t)d * n + i] : 0.0;
  }
  for (int e = t; e < n_pad; e += 256) al_s[e] = e < n ? alpha[e] : 0.0;
  __syncthreads();
  const int lane = t & 63, w = t >> 6, q = lane & 15, g = lane >> 4;
  const int64_t qi = ((int64_t)blockIdx.x * 4 + w) * 16 + q;
  const bool qv = qi < m;
  const int64_t qc = qv ? qi : m - 1;
  const int na0 = (int)prog_s[1];
  const bool se = (int)prog_s[0] == HILO_K_GAMMAEXP && klen == 3 + na0 + (int)prog_s[2 + na0] && prog_s[3 + na0 + 2] == 1.0;
  double kreg[NKB];
  double msum = 0.0;
  if (se) {   // one squared-exponential node (kernel.py:696 with p = 2), inline
    const double sf2 = prog_s[3 + na0], al = prog_s[3 + na0 + 1];
    const double* Md = prog_s + 3 + na0 + 3;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) kreg[kb] = 0.0;
    for (int k = 0; k < na0; ++k) {
      const int d = (int)prog_s[2 + k];
      const double xq = Xq[(int64_t)d * m + qc], Mk = Md[k];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const double df = Xt_s[d * n_pad + 4 * kb + g] - xq;
        kreg[kb] += df * Mk * df;
      }
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const double v = (4 * kb + g < n) ? sf2 * exp(-al * kreg[kb]) : 0.0;
      kreg[kb] = v;
      msum += v * al_s[4 * kb + g];
    }
  } else {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const int i = 4 * kb + g;
      const double v = i < n ? eval_kernel(prog_s, klen, Xt_s + i, n_pad, Xq + qc, m) : 0.0;
      kreg[kb] = v;
      msum += v * al_s[i];
    }
  }
  msum += __shfl_xor(msum, 16, 64);
  msum += __shfl_xor(msum, 32, 64);
  if (g == 0 && qv) mean[qi] = eval_mean(mprog, mlen, Xq + qi, m) + msum;
  if (!var) return;
  double ss = 0.0;
  // A operands (L^-1 from global / L2) in chunks of eight k-blocks, double buffered by hand: the loads of the next chunk are
  // issued before the MFMAs of the current one and fenced there (left alone, the scheduler serialises load -> wait -> MFMA
  // through one register pair).  Both loops unroll completely, every index below is a constant.
  // Row tiles are processed in PAIRS (two independent accumulator chains that share the B operands: one chain alone leaves
  // the matrix pipe idle for the accumulator latency - measured 193 cycles per dependent f64 MFMA against ~100 issued from
  // independent chains, tools/dbg/mfma_peak.hip).
  double abuf[2][2][8];
  const double* Lq = Ls + lane;   // operand-order copy of L^-1 (gp_linv_swizzle_kernel): block (it, kb) at 2 it (it + 1) + kb
#define GP_LOAD(IT, C, BUF)                                                                                        \
  {                                                                                                                \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                \
      BUF[0][u] = (8 * (C) + u < 4 * ((IT) + 1)) ? Lq[(2 * (IT) * ((IT) + 1) + 8 * (C) + u) * 64] : 0.0;           \
      BUF[1][u] = ((IT) + 1 < NT && 8 * (C) + u < 4 * ((IT) + 2)) ? Lq[(2 * ((IT) + 1) * ((IT) + 2) + 8 * (C) + u) * 64] : 0.0; \
    }                                                                                                              \
  }
  GP_LOAD(0, 0, abuf[0])
  int cur = 0;
#pragma unroll
  for (int it = 0; it < NT; it += 2) {
    gp_v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    const int nkb0 = 4 * (it + 1), nkb1 = (it + 1 < NT) ? 4 * (it + 2) : 0, nkb = nkb1 > nkb0 ? nkb1 : nkb0, nch = (nkb + 7) / 8;
#pragma unroll
    for (int c = 0; c < nch; ++c) {
      if (c + 1 < nch) GP_LOAD(it, c + 1, abuf[cur ^ 1])
      else if (it + 2 < NT) GP_LOAD(it + 2, 0, abuf[cur ^ 1])
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (8 * c + u < nkb0) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[cur][0][u], kreg[8 * c + u], acc0, 0, 0, 0);
        if (8 * c + u < nkb1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(abuf[cur][1][u], kreg[8 * c + u], acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      cur ^= 1;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) ss += acc0[r] * acc0[r] + acc1[r] * acc1[r];
  }
#undef GP_LOAD
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  if (g == 0 && qv) var[qi] = eval_kernel(prog_s, klen, Xq + qi, m, Xq + qi, m) - ss + sn2_add;
}

// host-side validation of a program: returns 0 when well formed, sets max stack depth
static int check_prog(const double* prog, int len, bool is_mean, int nf) {
  int pos = 0, sp = 0;
  while (pos < len) {
    if (pos + 2 > len) return fail(HILO_EINVAL, "program truncated at %d", pos);
    const int op = (int)prog[pos], na = (int)prog[pos + 1];
    if (na < 0 || pos + 3 + na > len) return fail(HILO_EINVAL, "bad active-dims count at %d", pos);
    for (int k = 0; k < na; ++k) {
      const int d = (int)prog[pos + 2 + k];
      if (d < 0 || d >= nf) return fail(HILO_EINVAL, "active dimension %d out of range (input space dimension %d)", d, nf);
    }
    const int npar = (int)prog[pos + 2 + na];
    if (npar < 0 || pos + 3 + na + npar > len) return fail(HILO_EINVAL, "bad parameter count at %d", pos);
    pos += 3 + na + npar;
    const bool binary = is_mean ? (op == HILO_M_SUM || op == HILO_M_PRODUCT) : (op == HILO_K_SUM || op == HILO_K_PRODUCT);
    const bool unary = is_mean ? (op == HILO_M_POWER || op == HILO_M_SCALE) : (op == HILO_K_POWER);
    if (!is_mean && op == HILO_K_XX_BEGIN) continue;
    if (binary) sp -= 1;
    else if (!unary) sp += 1;
    if (sp < 1 || sp > GP_STACK) return fail(HILO_EINVAL, "program stack depth %d out of range", sp);
  }
  if (sp != 1) return fail(HILO_EINVAL, "program leaves %d values on the stack", sp);
  return HILO_OK;
}

}  // namespace hilo

using namespace hilo;

struct hilo_gp {
  int device, nf, n, klen, mlen;
  int lp;           // row pitch of Linv: n rounded up to 16, plus 16 (zero padded: the MFMA predict kernel reads whole blocks)
  int linv_valid;   // L^-1 of the current factorisation has been built (predictive variance, gradient)
  double sn2, lml;
  double *X, *y, *kprog, *mprog, *L, *Linv, *LinvS, *alpha, *mu, *out;
  double *h_kprog, *h_mprog;  // host copies of the programs (gp_pack_se2)
};

static int upload(double** d, const double* h, size_t count) {
  hipError_t e = hipMalloc((void**)d, (count ? count : 1) * sizeof(double));
  if (e != hipSuccess) return fail(HILO_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e));
  if (count && h) HILO_HIP_CHECK(hipMemcpy(*d, h, count * sizeof(double), hipMemcpyHostToDevice));
  return HILO_OK;
}

extern "C" void hilo_gp_destroy(hilo_gp* gp) {
  if (!gp) return;
  double* ptrs[] = {gp->X, gp->y, gp->kprog, gp->mprog, gp->L, gp->Linv, gp->LinvS, gp->alpha, gp->mu, gp->out};
  for (double* p : ptrs)
    if (p) (void)hipFree(p);
  delete[] gp->h_kprog;
  delete[] gp->h_mprog;
  delete gp;
}

// K + sn2 I -> L, alpha, LML with the handle's current kernel program and noise variance
static int gp_factorize(hilo_gp* gp) {
  const int n = gp->n;
  hipStream_t s = 0;
  gp->linv_valid = 0;
  hipLaunchKernelGGL(kmat_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, gp->kprog, gp->klen, gp->nf, (int64_t)n, gp->X,
                     (int64_t)n, gp->X, gp->sn2, gp->L);
  hipLaunchKernelGGL(mean_kernel, dim3((n + 255) / 256), dim3(256), 0, s, gp->mprog, gp->mlen, (int64_t)n, gp->X, gp->mu);
  {
    const int T = (n + 15) / 16;
    const size_t lds = sizeof(double) * (16 * 17 + (size_t)T * 256 + (size_t)T * 16 * 16 + (size_t)T * 16);
    if (lds <= 150 * 1024 && !getenv("HILO_GP_FACTOR_UNBLOCKED")) {
      if (lds > 64 * 1024)
        HILO_HIP_CHECK(hipFuncSetAttribute((const void*)gp_factor_blocked_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(gp_factor_blocked_kernel, dim3(1), dim3(FACT_TPB), lds, s, n, gp->L, gp->y, gp->mu, gp->alpha, gp->out);
    } else {
      hipLaunchKernelGGL(gp_factor_kernel, dim3(1), dim3(FACT_TPB), 0, s, n, gp->L, gp->y, gp->mu, gp->alpha, gp->Linv, gp->out);
    }
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  double res[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(res, gp->out, sizeof(res), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(HILO_EHIP, "GP factorisation failed: %s", hipGetErrorString(e));
  if (res[1] != 0.0)   // its own code: a trial point of a hyper-parameter fit may well be indefinite (hilo_gp_refit)
    return fail(HILO_ENOTPD, "K + sn2 I is not positive definite (pivot %d); the reference adds no jitter "
                             "(inference.py:206)", (int)res[1]);
  gp->lml = res[0];
  return HILO_OK;
}

static int gp_ensure_linv(hilo_gp* gp) {
  if (gp->linv_valid) return HILO_OK;
  HILO_HIP_CHECK(hipMemsetAsync(gp->Linv, 0, sizeof(double) * (size_t)((gp->n + 15) & ~15) * gp->lp, (hipStream_t)0));
  hipLaunchKernelGGL(gp_linv_kernel, dim3(gp->n), dim3(64), sizeof(double) * gp->n, (hipStream_t)0, gp->n, gp->L, gp->Linv, gp->lp);
  {
    const int nt = (gp->n + 15) / 16;
    hipLaunchKernelGGL(gp_linv_swizzle_kernel, dim3(2 * nt * (nt + 1)), dim3(64), 0, (hipStream_t)0, nt, gp->Linv, gp->lp, gp->LinvS);
  }
  HILO_HIP_CHECK(hipGetLastError());
  HILO_HIP_CHECK(hipStreamSynchronize(0));
  gp->linv_valid = 1;
  return HILO_OK;
}

extern "C" int hilo_gp_create(int device, int nf, int n, const double* X_host, const double* y_host,
                              const double* kprog_host, int klen, const double* mprog_host, int mlen,
                              double noise_variance, hilo_gp** out) {
  HILO_REQUIRE(out && X_host && y_host && kprog_host && mprog_host, "hilo_gp_create: NULL argument");
  HILO_REQUIRE(nf >= 1 && n >= 1, "hilo_gp_create: need nf >= 1 and n >= 1 (got %d, %d)", nf, n);
  HILO_REQUIRE(noise_variance >= 0.0, "hilo_gp_create: negative noise variance");
  int rc = check_prog(kprog_host, klen, false, nf);
  if (rc) return rc;
  rc = check_prog(mprog_host, mlen, true, nf);
  if (rc) return rc;
  HILO_HIP_CHECK(hipSetDevice(device));
  hilo_gp* gp = new hilo_gp();
  memset(gp, 0, sizeof(*gp));
  gp->device = device; gp->nf = nf; gp->n = n; gp->klen = klen; gp->mlen = mlen;
  // inference.py:199: noise_variance = exp(2 * log_sigma_n) with log_sigma_n = log(noise_variance)/2 (kernel.py:127-130)
  gp->sn2 = noise_variance > 0.0 ? exp(2.0 * (log(noise_variance) / 2.0)) : 0.0;
  gp->h_kprog = new double[klen];
  gp->h_mprog = new double[mlen];
  memcpy(gp->h_kprog, kprog_host, sizeof(double) * klen);
  memcpy(gp->h_mprog, mprog_host, sizeof(double) * mlen);
  const size_t nn = (size_t)n * n;
#define UP(field, src, cnt) if ((rc = upload(&gp->field, src, cnt))) { hilo_gp_destroy(gp); return rc; }
  UP(X, X_host, (size_t)nf * n) UP(y, y_host, n) UP(kprog, kprog_host, klen) UP(mprog, mprog_host, mlen)
  gp->lp = ((n + 15) & ~15) + 16;
  UP(L, nullptr, nn) UP(Linv, nullptr, (size_t)((n + 15) & ~15) * gp->lp) UP(alpha, nullptr, n) UP(mu, nullptr, n) UP(out, nullptr, 2)
  { const int nt = (n + 15) / 16; UP(LinvS, nullptr, (size_t)2 * nt * (nt + 1) * 64) }
#undef UP
  rc = gp_factorize(gp);
  if (rc) { hilo_gp_destroy(gp); return rc; }
  *out = gp;
  return HILO_OK;
}

// Posterior mean of a two-feature squared-exponential GP in the layout the model functors read (hilo_models.h GpExt):
//   [n, sf2, bias, M_0, M_1, (X_0i, X_1i, alpha_i) * n];  mean(x*) = bias + sum_i alpha_i sf2 exp(-d2_i / 2)
// (inference.py:211-213 with kernel.py:696 at alpha = 1/2, gamma = 2 and mean.py:280-305)
int hilo::gp_pack_se2(const hilo_gp* gp, double** d_pack) {
  HILO_REQUIRE(gp && d_pack, "gp_pack_se2: NULL argument");
  const double* k = gp->h_kprog;
  const double* m = gp->h_mprog;
  const bool se = gp->nf == 2 && gp->klen == 10 && (int)k[0] == HILO_K_GAMMAEXP && (int)k[1] == 2 && (int)k[2] == 0 &&
                  (int)k[3] == 1 && (int)k[4] == 5 && k[6] == 0.5 && k[7] == 1.0;
  const bool cm = gp->mlen == 4 && (int)m[0] == HILO_M_CONST && (int)m[1] == 0;
  if (!se || !cm)
    return fail(HILO_ENOTSUP, "a GP inside a model must have a squared-exponential kernel over its two features and a "
                              "constant or zero mean in this build");
  const int n = gp->n;
  double* X = new double[2 * (size_t)n];
  double* a = new double[n];
  double* pack = new double[GP_PACK_HDR + 3 * (size_t)n];
  hipError_t e = hipSetDevice(gp->device);
  if (e == hipSuccess) e = hipMemcpy(X, gp->X, sizeof(double) * 2 * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(a, gp->alpha, sizeof(double) * n, hipMemcpyDeviceToHost);
  pack[0] = n; pack[1] = k[5]; pack[2] = m[3]; pack[3] = k[8]; pack[4] = k[9];
  for (int i = 0; i < n; ++i) {
    pack[GP_PACK_HDR + 3 * i] = X[i];
    pack[GP_PACK_HDR + 3 * i + 1] = X[n + i];
    pack[GP_PACK_HDR + 3 * i + 2] = a[i];
  }
  if (e == hipSuccess) e = hipMalloc((void**)d_pack, sizeof(double) * (GP_PACK_HDR + 3 * (size_t)n));
  if (e == hipSuccess) e = hipMemcpy(*d_pack, pack, sizeof(double) * (GP_PACK_HDR + 3 * (size_t)n), hipMemcpyHostToDevice);
  delete[] X; delete[] a; delete[] pack;
  if (e != hipSuccess) return fail(HILO_EHIP, "gp_pack_se2: %s", hipGetErrorString(e));
  return HILO_OK;
}

// Posterior mean of a squared-exponential GP with a constant / zero mean over ANY number of features, in the layout the
// run-time compiled models read (hilo_models.h::gp_se_mean): [n, na, sf2, bias, (active dim) * na, M * na, (X_{ad_k, i} * na,
// alpha_i) * n | sn2, L^-1 (n <= GP_VAR_MAX: gp_se_var)].  `Model.substitute_from(gp)` of a model written as expressions (dynamic_model.py:3040-3125).
int hilo::gp_pack_se(const hilo_gp* gp, double** d_pack) {
  HILO_REQUIRE(gp && d_pack, "gp_pack_se: NULL argument");
  const double* k = gp->h_kprog;
  const double* m = gp->h_mprog;
  const int na = (int)k[1];
  const bool se = (int)k[0] == HILO_K_GAMMAEXP && na >= 1 && na <= 8 && gp->klen == 3 + na + 3 + na && (int)k[2 + na] == 3 + na &&
                  k[3 + na + 1] == 0.5 && k[3 + na + 2] == 1.0;
  const bool cm = gp->mlen == 4 && (int)m[0] == HILO_M_CONST && (int)m[1] == 0;
  if (!cm)
    return fail(HILO_ENOTSUP, "a GP inside a run-time compiled model must have a constant or zero mean");
  const int n = gp->n, nf = gp->nf;
  if (!se) {
    // any other kernel is compiled into the model source (hilo_mpc_amd/codegen.py::gp_helper_source); what the model reads here is
    // the same table with ALL features as columns: [n, nf, NaN, bias, 0..nf-1, 1.., rows (X_0..X_{nf-1}, alpha)].  The signal
    // variance slot holds NaN: gp_se_mean on this table (a squared-exponential function called for another kernel) cannot go unnoticed.
    if (nf < 1 || nf > 8) return fail(HILO_ENOTSUP, "a GP inside a run-time compiled model takes 1 to 8 features (got %d)", nf);
    const size_t len = 4 + 2 * (size_t)nf + (size_t)n * (nf + 1);
    double* X = new double[(size_t)nf * n];
    double* a = new double[n];
    double* pack = new double[len];
    hipError_t e = hipSetDevice(gp->device);
    if (e == hipSuccess) e = hipMemcpy(X, gp->X, sizeof(double) * nf * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(a, gp->alpha, sizeof(double) * n, hipMemcpyDeviceToHost);
    pack[0] = n; pack[1] = nf; pack[2] = NAN; pack[3] = m[3];
    for (int q = 0; q < nf; ++q) { pack[4 + q] = q; pack[4 + nf + q] = 1.0; }
    for (int i = 0; i < n; ++i) {
      double* r = pack + 4 + 2 * nf + (size_t)i * (nf + 1);
      for (int q = 0; q < nf; ++q) r[q] = X[(size_t)q * n + i];
      r[nf] = a[i];
    }
    if (e == hipSuccess) e = hipMalloc((void**)d_pack, sizeof(double) * len);
    if (e == hipSuccess) e = hipMemcpy(*d_pack, pack, sizeof(double) * len, hipMemcpyHostToDevice);
    delete[] X; delete[] a; delete[] pack;
    if (e != hipSuccess) return fail(HILO_EHIP, "gp_pack_se: %s", hipGetErrorString(e));
    return HILO_OK;
  }
  // tail for the posterior variance (hilo_models.h::gp_se_var): [sn2, L^-1 row-major n x n], small training sets only
  const bool with_var = n <= 256;            // = hilo_models.h::GP_VAR_MAX (the k* array of gp_se_var; covers configuration 4's 200 points)
  const size_t head = 4 + 2 * (size_t)na + (size_t)n * (na + 1);
  const size_t len = head + (with_var ? 1 + (size_t)n * n : 0);
  if (with_var) {
    int rc = gp_ensure_linv(const_cast<hilo_gp*>(gp));
    if (rc) return rc;
  }
  double* X = new double[(size_t)nf * n];
  double* a = new double[n];
  double* pack = new double[len];
  hipError_t e = hipSetDevice(gp->device);
  if (e == hipSuccess) e = hipMemcpy(X, gp->X, sizeof(double) * nf * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(a, gp->alpha, sizeof(double) * n, hipMemcpyDeviceToHost);
  if (with_var) {
    pack[head] = gp->sn2;
    if (e == hipSuccess)
      e = hipMemcpy2D(pack + head + 1, sizeof(double) * n, gp->Linv, sizeof(double) * gp->lp, sizeof(double) * n, n,
                      hipMemcpyDeviceToHost);
  }
  pack[0] = n; pack[1] = na; pack[2] = k[3 + na]; pack[3] = m[3];
  for (int q = 0; q < na; ++q) { pack[4 + q] = k[2 + q]; pack[4 + na + q] = k[3 + na + 3 + q]; }
  for (int i = 0; i < n; ++i) {
    double* r = pack + 4 + 2 * na + (size_t)i * (na + 1);
    for (int q = 0; q < na; ++q) r[q] = X[(size_t)(int)k[2 + q] * n + i];
    r[na] = a[i];
  }
  if (e == hipSuccess) e = hipMalloc((void**)d_pack, sizeof(double) * len);
  if (e == hipSuccess) e = hipMemcpy(*d_pack, pack, sizeof(double) * len, hipMemcpyHostToDevice);
  delete[] X; delete[] a; delete[] pack;
  if (e != hipSuccess) return fail(HILO_EHIP, "gp_pack_se: %s", hipGetErrorString(e));
  return HILO_OK;
}

extern "C" int hilo_gp_log_marginal_likelihood(hilo_gp* gp, double* lml_host) {
  HILO_REQUIRE(gp && lml_host, "hilo_gp_log_marginal_likelihood: NULL argument");
  *lml_host = gp->lml;
  return HILO_OK;
}

// New hyper-parameters on the same training data: only the kernel program and the noise variance are uploaded, every buffer
// of the handle is reused (what `GaussianProcess.fit_model` calls per objective value; gp.py:660-697).  On HILO_ENOTPD the
// handle keeps the programs of the failed trial point but no valid factorisation: refit again before predicting.
extern "C" int hilo_gp_refit(hilo_gp* gp, const double* kprog_host, int klen, double noise_variance) {
  HILO_REQUIRE(gp && kprog_host, "hilo_gp_refit: NULL argument");
  HILO_REQUIRE(klen == gp->klen, "hilo_gp_refit: the kernel program changed its length (%d -> %d): the kernel structure is fixed, "
                                 "only its hyper-parameters may change", gp->klen, klen);
  HILO_REQUIRE(noise_variance >= 0.0, "hilo_gp_refit: negative noise variance");
  int rc = check_prog(kprog_host, klen, false, gp->nf);
  if (rc) return rc;
  HILO_HIP_CHECK(hipSetDevice(gp->device));
  memcpy(gp->h_kprog, kprog_host, sizeof(double) * klen);
  HILO_HIP_CHECK(hipMemcpy(gp->kprog, kprog_host, sizeof(double) * klen, hipMemcpyHostToDevice));
  gp->sn2 = noise_variance > 0.0 ? exp(2.0 * (log(noise_variance) / 2.0)) : 0.0;
  return gp_factorize(gp);
}

// New hyper-parameters of the MEAN function (same program length: the structure is fixed); takes effect with the next
// hilo_gp_refit, which re-evaluates the mean on the training inputs before it factorises.
extern "C" int hilo_gp_set_mean_program(hilo_gp* gp, const double* mprog_host, int mlen) {
  HILO_REQUIRE(gp && mprog_host, "hilo_gp_set_mean_program: NULL argument");
  HILO_REQUIRE(mlen == gp->mlen, "hilo_gp_set_mean_program: the mean program changed its length (%d -> %d)", gp->mlen, mlen);
  int rc = check_prog(mprog_host, mlen, true, gp->nf);
  if (rc) return rc;
  HILO_HIP_CHECK(hipSetDevice(gp->device));
  memcpy(gp->h_mprog, mprog_host, sizeof(double) * mlen);
  HILO_HIP_CHECK(hipMemcpy(gp->mprog, mprog_host, sizeof(double) * mlen, hipMemcpyHostToDevice));
  return HILO_OK;
}

// Gradient of the log marginal likelihood with respect to n_theta hyper-parameters at the handle's current point, by the
// trace formula 1/2 tr((alpha alpha^T - K_y^-1) dK_y/dtheta_j) on the device.  For each theta_j the caller passes
// the kernel programs at theta + h_j e_j and theta - h_j e_j ([n_theta][2][klen], host) and the noise variances there
// ([n_theta][2], host): dK_y/dtheta_j is their central difference, evaluated element-wise next to the trace.
extern "C" int hilo_gp_lml_gradient(hilo_gp* gp, int n_theta, const double* kprogs_pm_host, const double* noise_pm_host,
                                    const double* h_host, double* grad_host) {
  HILO_REQUIRE(gp && kprogs_pm_host && noise_pm_host && h_host && grad_host, "hilo_gp_lml_gradient: NULL argument");
  HILO_REQUIRE(n_theta >= 1 && n_theta <= 64, "hilo_gp_lml_gradient: n_theta out of range");
  HILO_HIP_CHECK(hipSetDevice(gp->device));
  int rc = gp_ensure_linv(gp);
  if (rc) return rc;
  const int n = gp->n, klen = gp->klen;
  for (int j = 0; j < 2 * n_theta; ++j)
    if ((rc = check_prog(kprogs_pm_host + (size_t)j * klen, klen, false, gp->nf))) return rc;
  double *d_progs = nullptr, *d_aux = nullptr, *d_A = nullptr;
  double aux[3 * 64];
  for (int j = 0; j < n_theta; ++j) {
    HILO_REQUIRE(h_host[j] > 0.0, "hilo_gp_lml_gradient: step %d must be positive", j);
    aux[j] = (noise_pm_host[2 * j] - noise_pm_host[2 * j + 1]) / (2.0 * h_host[j]);   // d sn2 / d theta_j
    aux[64 + j] = 1.0 / (2.0 * h_host[j]);
    aux[128 + j] = 0.0;
  }
  if ((rc = upload(&d_progs, kprogs_pm_host, (size_t)2 * n_theta * klen))) return rc;
  if ((rc = upload(&d_aux, aux, 3 * 64))) { (void)hipFree(d_progs); return rc; }
  if ((rc = upload(&d_A, nullptr, (size_t)n * n))) { (void)hipFree(d_progs); (void)hipFree(d_aux); return rc; }
  hipLaunchKernelGGL(gp_amat_kernel, dim3((n * n + 255) / 256), dim3(256), 0, (hipStream_t)0, n, gp->alpha, gp->Linv, gp->lp, d_A);
  int gx = (n * n + 255) / 256;
  gx = gx > 64 ? 64 : gx;
  hipLaunchKernelGGL(gp_grad_kernel, dim3(gx, n_theta), dim3(256), 0, (hipStream_t)0, n, gp->nf, klen, d_progs, d_aux, d_aux + 64, gp->X,
                     d_A, d_aux + 128);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(grad_host, d_aux + 128, sizeof(double) * n_theta, hipMemcpyDeviceToHost);
  (void)hipFree(d_progs); (void)hipFree(d_aux); (void)hipFree(d_A);
  if (e != hipSuccess) return fail(HILO_EHIP, "hilo_gp_lml_gradient: %s", hipGetErrorString(e));
  return HILO_OK;
}

extern "C" int hilo_gp_predict(hilo_gp* gp, int64_t m, const double* Xq, int noise_free, double* mean, double* var,
                               void* stream) {
  HILO_REQUIRE(gp, "hilo_gp_predict: NULL handle");
  HILO_REQUIRE(m >= 0, "hilo_gp_predict: negative query count");
  if (m == 0) return HILO_OK;
  HILO_REQUIRE(Xq && mean, "hilo_gp_predict: NULL argument");
  HILO_HIP_CHECK(hipSetDevice(gp->device));
  if (var) {
    int rcl = gp_ensure_linv(gp);   // L^-1 is built on first use after a (re)fit
    if (rcl) return rcl;
  }
  const double sn2 = noise_free ? 0.0 : gp->sn2;
  const int n_pad = (gp->n + 15) & ~15;
  if (n_pad <= 256 && !getenv("HILO_GP_PREDICT_VALU") && !getenv("HILO_GP_PREDICT_LDS")) {
    const size_t lds_r = (gp->klen + (size_t)(gp->nf + 1) * n_pad) * sizeof(double);
    const unsigned grid = (unsigned)((m + 63) / 64);
    switch (n_pad / 16) {
#define R(NT) case NT: hipLaunchKernelGGL((gp_predict_reg_kernel<NT>), dim3(grid), dim3(256), lds_r, (hipStream_t)stream, gp->kprog, \
                                         gp->klen, gp->mprog, gp->mlen, gp->nf, gp->n, gp->X, gp->alpha, gp->LinvS, sn2, m, Xq, \
                                         mean, var); break;
      R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15) R(16)
#undef R
    }
    HILO_HIP_CHECK(hipGetLastError());
    return HILO_OK;
  }
  // f64 MFMA path: the K* tile of Q = 16 W queries must fit the LDS next to the reduction scratch
  int W = 4;
  auto lds_of = [&](int w) { return ((size_t)(n_pad + 16) * (16 * w + 16) + 64 * w + gp->klen + (size_t)gp->nf * (n_pad + 16 * w)) * sizeof(double); };
  while (W > 1 && lds_of(W) > 150 * 1024) W >>= 1;
  const size_t lds_m = lds_of(W);
  if (lds_m <= 150 * 1024 && !getenv("HILO_GP_PREDICT_VALU")) {
    const unsigned grid = (unsigned)((m + 16 * W - 1) / (16 * W));
#define LAUNCH_MFMA(WW)                                                                                                         \
    {                                                                                                                           \
      if (lds_m > 64 * 1024)                                                                                                    \
        HILO_HIP_CHECK(hipFuncSetAttribute((const void*)gp_predict_mfma_kernel<WW>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           (int)lds_m));                                                                        \
      hipLaunchKernelGGL((gp_predict_mfma_kernel<WW>), dim3(grid), dim3(64 * WW), lds_m, (hipStream_t)stream, gp->kprog,         \
                         gp->klen, gp->mprog, gp->mlen, gp->nf, gp->n, gp->X, gp->alpha, gp->Linv, gp->lp, sn2, m, Xq, mean, var); \
    }
    if (W == 4) LAUNCH_MFMA(4) else if (W == 2) LAUNCH_MFMA(2) else LAUNCH_MFMA(1)
#undef LAUNCH_MFMA
    HILO_HIP_CHECK(hipGetLastError());
    return HILO_OK;
  }
  int Q = 32;
  while (Q > 1 && ((size_t)gp->n * Q + PRED_TPB) * sizeof(double) > 128 * 1024) Q >>= 1;
  const size_t lds = ((size_t)gp->n * Q + PRED_TPB) * sizeof(double);
  if (lds > 160 * 1024) return fail(HILO_ENOTSUP, "n = %d training points exceed the LDS-resident predict tile", gp->n);
  if (lds > 64 * 1024)
    HILO_HIP_CHECK(hipFuncSetAttribute((const void*)gp_predict_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned grid = (unsigned)((m + Q - 1) / Q);
  hipLaunchKernelGGL(gp_predict_kernel, dim3(grid), dim3(PRED_TPB), lds, (hipStream_t)stream, gp->kprog, gp->klen,
                     gp->mprog, gp->mlen, gp->nf, gp->n, gp->X, gp->alpha, gp->Linv, gp->lp, sn2, m, Xq, Q,
                     mean, var);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

extern "C" int hilo_gp_kernel_matrix(int device, int nf, const double* kprog_host, int klen, int64_t n1, const double* X1,
                                     int64_t n2, const double* X2, double* K, void* stream) {
  HILO_REQUIRE(kprog_host, "hilo_gp_kernel_matrix: NULL argument");
  int rc = check_prog(kprog_host, klen, false, nf);
  if (rc) return rc;
  if (n1 == 0 || n2 == 0) return HILO_OK;
  HILO_REQUIRE(X1 && X2 && K, "hilo_gp_kernel_matrix: NULL argument");
  HILO_HIP_CHECK(hipSetDevice(device));
  double* dprog = nullptr;
  if ((rc = upload(&dprog, kprog_host, klen))) return rc;
  hipLaunchKernelGGL(kmat_kernel, dim3((unsigned)((n2 + 255) / 256), (unsigned)(n1 < 32768 ? n1 : 32768)), dim3(256), 0, (hipStream_t)stream, dprog,
                     klen, nf, n1, X1, n2, X2, 0.0, K);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(dprog);
  if (e != hipSuccess) return fail(HILO_EHIP, "kernel matrix launch failed: %s", hipGetErrorString(e));
  return HILO_OK;
}

extern "C" int hilo_gp_mean(int device, int nf, const double* mprog_host, int mlen, int64_t n, const double* X, double* mu,
                            void* stream) {
  HILO_REQUIRE(mprog_host, "hilo_gp_mean: NULL argument");
  int rc = check_prog(mprog_host, mlen, true, nf);
  if (rc) return rc;
  if (n == 0) return HILO_OK;
  HILO_REQUIRE(X && mu, "hilo_gp_mean: NULL argument");
  HILO_HIP_CHECK(hipSetDevice(device));
  double* dprog = nullptr;
  if ((rc = upload(&dprog, mprog_host, mlen))) return rc;
  hipLaunchKernelGGL(mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dprog, mlen, n, X, mu);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  (void)hipFree(dprog);
  if (e != hipSuccess) return fail(HILO_EHIP, "mean launch failed: %s", hipGetErrorString(e));
  return HILO_OK;
}
