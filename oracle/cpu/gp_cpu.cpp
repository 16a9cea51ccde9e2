// CPU baseline of the GP prediction (mean + variance per query column): oracle/gp.py::Posterior.predict for the squared-exponential
// kernel with one length scale per feature (hilo_mpc/modules/machine_learning/gp/gp.py:699-718, kernel.py:538-555, :696;
// inference.py:197-217), C++17 + OpenMP over blocks of query columns, the columns of a block side by side so that the kernel
// vector, the mean and the triangular substitution vectorise across them.
// TEST INFRASTRUCTURE / BASELINE ONLY - see nmpc_cpu.cpp.  Validated against the numpy oracle in tests/test_cpu_baseline.py.
#include <omp.h>

#include <cmath>
#include <cstdint>
#include <vector>

extern "C" {

// HOST pointers.  X [nf][n] training inputs, alpha [n], R [n][n] upper factor (R' R = K + sn2 I, row-major), Minv [nf] = 1 / l_d^2,
// Xq [nf][m] query columns; outputs mu [m], var [m] (with the noise variance unless noise_free).  n_threads <= 0: all cores.
int hilo_cpu_gp_predict(int n, int nf, const double* X, const double* alpha, const double* R, double sf2, const double* Minv,
                        double bias, double sn2, int noise_free, int64_t m, const double* Xq, double* mu, double* var,
                        int n_threads) {
  if (!X || !alpha || !R || !Minv || !Xq || !mu || !var || n < 1 || nf < 1) return 1;
  if (n_threads <= 0) n_threads = omp_get_max_threads();
  constexpr int Q = 8;                              // query columns per block
  // the substitution walks the columns of R' = rows of R below the diagonal: keep R' packed by rows, 1 / diagonal aside
  std::vector<double> Lt((size_t)n * n), dinv(n);
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < i; ++j) Lt[(size_t)i * n + j] = R[(size_t)j * n + i];
    dinv[i] = 1.0 / R[(size_t)i * n + i];
  }
  const int64_t nblk = (m + Q - 1) / Q;
#pragma omp parallel num_threads(n_threads)
  {
    std::vector<double> v((size_t)n * Q);
#pragma omp for schedule(static)
    for (int64_t blk = 0; blk < nblk; ++blk) {
      const int64_t q0 = blk * Q;
      const int nq = (int)(m - q0 < Q ? m - q0 : Q);
      double xq[16][Q], acc[Q], s2[Q];
      for (int d = 0; d < nf && d < 16; ++d)
        for (int q = 0; q < Q; ++q) xq[d][q] = Xq[(size_t)d * m + q0 + (q < nq ? q : 0)];
      for (int q = 0; q < Q; ++q) { acc[q] = bias; s2[q] = 0.0; }
      for (int i = 0; i < n; ++i) {               // k* and the mean
        double r2[Q];
        for (int q = 0; q < Q; ++q) r2[q] = 0.0;
        for (int d = 0; d < nf; ++d) {
          const double xi = X[(size_t)d * n + i], Md = Minv[d];
#pragma omp simd
          for (int q = 0; q < Q; ++q) { const double t = xq[d][q] - xi; r2[q] += Md * t * t; }
        }
        const double a = alpha[i];
        for (int q = 0; q < Q; ++q) {
          const double k = sf2 * std::exp(-0.5 * r2[q]);
          v[(size_t)i * Q + q] = k;
          acc[q] += a * k;
        }
      }
      for (int i = 0; i < n; ++i) {               // v = R'^-1 k*, |v|^2
        double t[Q];
        for (int q = 0; q < Q; ++q) t[q] = v[(size_t)i * Q + q];
        const double* Li = &Lt[(size_t)i * n];
        for (int j = 0; j < i; ++j) {
          const double l = Li[j];
          const double* vj = &v[(size_t)j * Q];
#pragma omp simd
          for (int q = 0; q < Q; ++q) t[q] -= l * vj[q];
        }
        for (int q = 0; q < Q; ++q) {
          t[q] *= dinv[i];
          v[(size_t)i * Q + q] = t[q];
          s2[q] += t[q] * t[q];
        }
      }
      for (int q = 0; q < nq; ++q) {
        mu[q0 + q] = acc[q];
        var[q0 + q] = sf2 - s2[q] + (noise_free ? 0.0 : sn2);
      }
    }
  }
  return 0;
}

}  // extern "C"
