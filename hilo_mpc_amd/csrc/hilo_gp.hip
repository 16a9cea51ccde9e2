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
  const int64_t i = blockIdx.y;
  if (j >= n2) return;
  double v = eval_kernel(prog, len, X1 + i, n1, X2 + j, n2);
  if (i == j) v += diag_add;
  K[i * n2 + j] = v;
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
  // --- Linv = L^-1 : column c by forward substitution, one column per thread --------------------------
  for (int c = t; c < n; c += FACT_TPB) {
    for (int i = 0; i < n; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      if (i >= c) {
        for (int j = c; j < i; ++j) s -= A[(int64_t)i * n + j] * Linv[(int64_t)j * n + c];
        s /= A[(int64_t)i * n + i];
      } else {
        s = 0.0;
      }
      Linv[(int64_t)i * n + c] = s;
    }
  }
}

// Prediction for a tile of Q queries per workgroup (gp.py:699-718, inference.py:212-217).
//   phase 1: K*[i][q] = k(X[:,i], x_q) into LDS
//   phase 2: mean_q = m(x_q) + sum_i K*[i][q] alpha_i
//   phase 3: var_q = k(x_q,x_q) - |Linv K*[:,q]|^2 (+ sn2)
constexpr int PRED_TPB = 256;
__global__ __launch_bounds__(PRED_TPB) void gp_predict_kernel(const double* __restrict__ kprog, int klen,
                                                              const double* __restrict__ mprog, int mlen, int nf, int n,
                                                              const double* __restrict__ Xt, const double* __restrict__ alpha,
                                                              const double* __restrict__ Linv, double sn2_add, int64_t m,
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
      const double* Li = Linv + (int64_t)i * n;
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
  double sn2, lml;
  double *X, *y, *kprog, *mprog, *L, *Linv, *alpha, *mu, *out;
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
  double* ptrs[] = {gp->X, gp->y, gp->kprog, gp->mprog, gp->L, gp->Linv, gp->alpha, gp->mu, gp->out};
  for (double* p : ptrs)
    if (p) (void)hipFree(p);
  delete[] gp->h_kprog;
  delete[] gp->h_mprog;
  delete gp;
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
  UP(L, nullptr, nn) UP(Linv, nullptr, nn) UP(alpha, nullptr, n) UP(mu, nullptr, n) UP(out, nullptr, 2)
#undef UP
  hipStream_t s = 0;
  hipLaunchKernelGGL(kmat_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, gp->kprog, klen, nf, (int64_t)n, gp->X,
                     (int64_t)n, gp->X, gp->sn2, gp->L);
  hipLaunchKernelGGL(mean_kernel, dim3((n + 255) / 256), dim3(256), 0, s, gp->mprog, mlen, (int64_t)n, gp->X, gp->mu);
  hipLaunchKernelGGL(gp_factor_kernel, dim3(1), dim3(FACT_TPB), 0, s, n, gp->L, gp->y, gp->mu, gp->alpha, gp->Linv, gp->out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  double res[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(res, gp->out, sizeof(res), hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    hilo_gp_destroy(gp);
    return fail(HILO_EHIP, "GP factorisation failed: %s", hipGetErrorString(e));
  }
  if (res[1] != 0.0) {
    hilo_gp_destroy(gp);
    return fail(HILO_EINVAL, "K + sn2 I is not positive definite (pivot %d); the reference adds no jitter "
                             "(inference.py:206)", (int)res[1]);
  }
  gp->lml = res[0];
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

extern "C" int hilo_gp_log_marginal_likelihood(hilo_gp* gp, double* lml_host) {
  HILO_REQUIRE(gp && lml_host, "hilo_gp_log_marginal_likelihood: NULL argument");
  *lml_host = gp->lml;
  return HILO_OK;
}

extern "C" int hilo_gp_predict(hilo_gp* gp, int64_t m, const double* Xq, int noise_free, double* mean, double* var,
                               void* stream) {
  HILO_REQUIRE(gp, "hilo_gp_predict: NULL handle");
  HILO_REQUIRE(m >= 0, "hilo_gp_predict: negative query count");
  if (m == 0) return HILO_OK;
  HILO_REQUIRE(Xq && mean, "hilo_gp_predict: NULL argument");
  HILO_HIP_CHECK(hipSetDevice(gp->device));
  int Q = 32;
  while (Q > 1 && ((size_t)gp->n * Q + PRED_TPB) * sizeof(double) > 128 * 1024) Q >>= 1;
  const size_t lds = ((size_t)gp->n * Q + PRED_TPB) * sizeof(double);
  if (lds > 160 * 1024) return fail(HILO_ENOTSUP, "n = %d training points exceed the LDS-resident predict tile", gp->n);
  if (lds > 64 * 1024)
    HILO_HIP_CHECK(hipFuncSetAttribute((const void*)gp_predict_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned grid = (unsigned)((m + Q - 1) / Q);
  hipLaunchKernelGGL(gp_predict_kernel, dim3(grid), dim3(PRED_TPB), lds, (hipStream_t)stream, gp->kprog, gp->klen,
                     gp->mprog, gp->mlen, gp->nf, gp->n, gp->X, gp->alpha, gp->Linv, noise_free ? 0.0 : gp->sn2, m, Xq, Q,
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
  hipLaunchKernelGGL(kmat_kernel, dim3((unsigned)((n2 + 255) / 256), (unsigned)n1), dim3(256), 0, (hipStream_t)stream, dprog,
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
