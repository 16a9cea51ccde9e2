// CPU baseline of the LMPC hot path (BASELINE configuration 1): the dense convex QP of `LMPC.optimize`
// (hilo_mpc/modules/controller/mpc.py:2307-2394: one `ca.conic` call per measured state) for a batch of instances,
// C++17 + OpenMP over the instances.
//
// TEST INFRASTRUCTURE / BASELINE ONLY: loaded by bench.py's `cpu_baseline` leg and by tests/ (through oracle/cpu/__init__.py),
// never by the product package.  Validated against oracle/lmpc.py::solve_qp in tests/test_cpu_baseline.py before it is timed.
//
// The iteration is oracle/lmpc.py::solve_qp statement by statement, without its active-set polish (which the product's kernels
// do not have either): Mehrotra's predictor-corrector on the free variables (fixed ones, lb == ub, substituted), the Newton system
// by the Schur complement of the equality constraints - M = H + Sigma + reg I = L L^T (one square root per entry when H is diagonal,
// as in the device kernel), X = L^-1 A^T, S = X^T X + reg I = Ls Ls^T -
// convergence on max(|rd| / (1 + |g|), |rp|, mu) <= tol, infeasibility by OOQP's rule (merit 1e4 times its smallest value so far).
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace {

constexpr double INF = std::numeric_limits<double>::infinity();

struct Work {
  int n, m;
  std::vector<double> Hf, Af, gf, bf, l, u, x, y, zl, zu, M, X, S, base, r1, dx, dy, dzl, dzu, t, rp, xfix, cl, cu, dq;
  std::vector<char> fixed, hl, hu;
  Work(int n_, int m_) : n(n_), m(m_), Hf(n_ * n_), Af(m_ * n_), gf(n_), bf(m_), l(n_), u(n_), x(n_), y(m_), zl(n_), zu(n_), M(n_ * n_),
                         X(n_ * m_), S(m_ * m_), base(n_), r1(n_), dx(n_), dy(m_), dzl(n_), dzu(n_), t(n_), rp(m_), xfix(n_), cl(n_), cu(n_), dq(n_),
                         fixed(n_), hl(n_), hu(n_) {}
};

bool cholesky(double* A, int n) {   // lower factor in place
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0)) return false;
    const double s = std::sqrt(d);
    A[j * n + j] = s;
    for (int i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (int k = 0; k < j; ++k) v -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = v / s;
    }
  }
  return true;
}
void fsub(const double* L, int n, double* b) {   // L z = b
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int j = 0; j < i; ++j) s -= L[i * n + j] * b[j];
    b[i] = s / L[i * n + i];
  }
}
void bsub(const double* L, int n, double* b) {   // L^T z = b
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < n; ++j) s -= L[j * n + i] * b[j];
    b[i] = s / L[i * n + i];
  }
}

// one instance; returns the status (1 solved, 3 infeasible, 5 iteration limit, -1 factorisation failed)
int solve_one(Work& k, const double* H, const double* g, const double* A, const double* b, const double* lb, const double* ub,
              double tol, int max_iter, double reg, double* xout, int* iters) {
  const int n = k.n, m = k.m;
  for (int i = 0; i < n; ++i) {
    k.fixed[i] = lb[i] == ub[i];
    k.xfix[i] = k.fixed[i] ? lb[i] : 0.0;
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) k.Hf[i * n + j] = (k.fixed[i] || k.fixed[j]) ? (i == j ? 1.0 : 0.0) : H[i * n + j];
  for (int r = 0; r < m; ++r)
    for (int j = 0; j < n; ++j) k.Af[r * n + j] = k.fixed[j] ? 0.0 : A[r * n + j];
  // H diagonal (the LMPC with diagonal weights): H + Sigma is diagonal too - its factor is a square root per entry, like in the
  // device kernel (csrc/hilo_qp.hip); dq holds the diagonal of the factor
  bool hdiag = true;
  for (int i = 0; i < n && hdiag; ++i)
    for (int j = 0; j < n; ++j)
      if (i != j && k.Hf[i * n + j] != 0.0) { hdiag = false; break; }
  double gmax = 0.0;
  for (int i = 0; i < n; ++i) {
    double s = g[i];
    for (int j = 0; j < n; ++j) s += H[i * n + j] * k.xfix[j];
    k.gf[i] = k.fixed[i] ? 0.0 : s;
    gmax = std::max(gmax, std::fabs(k.gf[i]));
  }
  for (int r = 0; r < m; ++r) {
    double s = b[r];
    for (int j = 0; j < n; ++j) s -= A[r * n + j] * k.xfix[j];
    k.bf[r] = s;
    k.y[r] = 0.0;
  }
  int nb = 0;
  for (int i = 0; i < n; ++i) {
    k.hl[i] = !k.fixed[i] && lb[i] > -INF;
    k.hu[i] = !k.fixed[i] && ub[i] < INF;
    k.l[i] = lb[i];
    k.u[i] = ub[i];
    double xi = 0.0;
    if (k.hl[i] && k.hu[i]) xi = 0.5 * (lb[i] + ub[i]);
    else if (k.hl[i]) xi = std::max(0.0, lb[i] + 1.0);
    else if (k.hu[i]) xi = std::min(0.0, ub[i] - 1.0);
    k.x[i] = xi;
    k.zl[i] = k.hl[i] ? 1.0 : 0.0;
    k.zu[i] = k.hu[i] ? 1.0 : 0.0;
    nb += (int)k.hl[i] + (int)k.hu[i];
  }
  nb = std::max(nb, 1);
  int status = 5, it = 0;
  double phi_min = INF;
  auto newton = [&](const double* r, double* dx, double* dy) {
    if (hdiag) {
      for (int i = 0; i < n; ++i) k.t[i] = r[i] / k.dq[i];
    } else {
      for (int i = 0; i < n; ++i) k.t[i] = r[i];
      fsub(k.M.data(), n, k.t.data());
    }
    for (int a = 0; a < m; ++a) {
      double s = k.rp[a];
      for (int i = 0; i < n; ++i) s += k.X[i * m + a] * k.t[i];
      dy[a] = s;
    }
    fsub(k.S.data(), m, dy);
    bsub(k.S.data(), m, dy);
    for (int i = 0; i < n; ++i) {
      double s = k.t[i];
      for (int a = 0; a < m; ++a) s -= k.X[i * m + a] * dy[a];
      dx[i] = hdiag ? s / k.dq[i] : s;
    }
    if (!hdiag) bsub(k.M.data(), n, dx);
  };
  auto steps = [&](double tau, double& ap, double& ad) {
    ap = ad = 1.0;
    for (int i = 0; i < n; ++i) {
      if (k.hl[i]) {
        const double s = k.x[i] - k.l[i];
        if (k.dx[i] < 0.0) ap = std::min(ap, -tau * s / k.dx[i]);
        if (k.dzl[i] < 0.0) ad = std::min(ad, -tau * k.zl[i] / k.dzl[i]);
      }
      if (k.hu[i]) {
        const double s = k.u[i] - k.x[i];
        if (k.dx[i] > 0.0) ap = std::min(ap, tau * s / k.dx[i]);
        if (k.dzu[i] < 0.0) ad = std::min(ad, -tau * k.zu[i] / k.dzu[i]);
      }
    }
  };
  for (it = 0; it < max_iter; ++it) {
    double rdmax = 0.0, rpmax = 0.0, mu = 0.0;
    for (int i = 0; i < n; ++i) {
      double s = k.gf[i];
      for (int j = 0; j < n; ++j) s += k.Hf[i * n + j] * k.x[j];
      for (int r = 0; r < m; ++r) s += k.Af[r * n + i] * k.y[r];
      if (k.fixed[i]) s = 0.0;
      k.base[i] = -s;
      rdmax = std::max(rdmax, std::fabs(s - k.zl[i] + k.zu[i]));
      if (k.hl[i]) mu += (k.x[i] - k.l[i]) * k.zl[i];
      if (k.hu[i]) mu += (k.u[i] - k.x[i]) * k.zu[i];
    }
    for (int r = 0; r < m; ++r) {
      double s = -k.bf[r];
      for (int j = 0; j < n; ++j) s += k.Af[r * n + j] * k.x[j];
      k.rp[r] = s;
      rpmax = std::max(rpmax, std::fabs(s));
    }
    mu /= nb;
    const double phi = std::max(std::max(rdmax / (1.0 + gmax), rpmax), mu);
    if (!std::isfinite(phi)) { status = 3; break; }
    if (phi <= tol) { status = 1; break; }
    phi_min = std::min(phi_min, phi);
    if (phi >= 1e4 * phi_min) { status = 3; break; }
    if (hdiag) {
      bool ok = true;
      for (int i = 0; i < n; ++i) {
        double d = k.Hf[i * n + i];
        if (!k.fixed[i]) {
          d += reg;
          if (k.hl[i]) d += k.zl[i] / (k.x[i] - k.l[i]);
          if (k.hu[i]) d += k.zu[i] / (k.u[i] - k.x[i]);
        }
        ok = ok && d > 0.0;
        k.dq[i] = std::sqrt(d);
      }
      if (!ok) { status = -1; break; }
      for (int i = 0; i < n; ++i) {
        const double di = 1.0 / k.dq[i];
        for (int a = 0; a < m; ++a) k.X[i * m + a] = k.Af[a * n + i] * di;
      }
    } else {
      for (int i = 0; i < n * n; ++i) k.M[i] = k.Hf[i];
      for (int i = 0; i < n; ++i) {
        if (k.fixed[i]) continue;
        double d = reg;
        if (k.hl[i]) d += k.zl[i] / (k.x[i] - k.l[i]);
        if (k.hu[i]) d += k.zu[i] / (k.u[i] - k.x[i]);
        k.M[i * n + i] += d;
      }
      if (!cholesky(k.M.data(), n)) { status = -1; break; }
      for (int a = 0; a < m; ++a) {
        for (int i = 0; i < n; ++i) k.t[i] = k.Af[a * n + i];
        fsub(k.M.data(), n, k.t.data());
        for (int i = 0; i < n; ++i) k.X[i * m + a] = k.t[i];
      }
    }
    for (int a = 0; a < m; ++a)
      for (int c = 0; c <= a; ++c) {
        double s = a == c ? reg : 0.0;
        for (int i = 0; i < n; ++i) s += k.X[i * m + a] * k.X[i * m + c];
        k.S[a * m + c] = k.S[c * m + a] = s;
      }
    if (m > 0 && !cholesky(k.S.data(), m)) { status = -1; break; }
    newton(k.base.data(), k.dx.data(), k.dy.data());                                    // predictor (sigma = 0)
    for (int i = 0; i < n; ++i) {
      k.dzl[i] = k.hl[i] ? -k.zl[i] - k.zl[i] / (k.x[i] - k.l[i]) * k.dx[i] : 0.0;
      k.dzu[i] = k.hu[i] ? -k.zu[i] + k.zu[i] / (k.u[i] - k.x[i]) * k.dx[i] : 0.0;
    }
    double ap, ad;
    steps(1.0, ap, ad);
    double mu_aff = 0.0;
    for (int i = 0; i < n; ++i) {
      if (k.hl[i]) mu_aff += (k.x[i] - k.l[i] + ap * k.dx[i]) * (k.zl[i] + ad * k.dzl[i]);
      if (k.hu[i]) mu_aff += (k.u[i] - k.x[i] - ap * k.dx[i]) * (k.zu[i] + ad * k.dzu[i]);
    }
    mu_aff /= nb;
    const double sg = mu > 0.0 ? mu_aff / mu : 0.0, sm = sg * sg * sg * mu;
    for (int i = 0; i < n; ++i) {
      double s = k.base[i];
      if (k.hl[i]) s += (sm - k.dx[i] * k.dzl[i]) / (k.x[i] - k.l[i]);
      if (k.hu[i]) s -= (sm + k.dx[i] * k.dzu[i]) / (k.u[i] - k.x[i]);
      k.r1[i] = s;
    }
    double* cl = k.cl.data();                                                           // second-order terms of the predictor
    double* cu = k.cu.data();
    for (int i = 0; i < n; ++i) { cl[i] = k.dx[i] * k.dzl[i]; cu[i] = -k.dx[i] * k.dzu[i]; }
    newton(k.r1.data(), k.dx.data(), k.dy.data());
    for (int i = 0; i < n; ++i) {
      k.dzl[i] = k.hl[i] ? (sm - cl[i]) / (k.x[i] - k.l[i]) - k.zl[i] - k.zl[i] / (k.x[i] - k.l[i]) * k.dx[i] : 0.0;
      k.dzu[i] = k.hu[i] ? (sm - cu[i]) / (k.u[i] - k.x[i]) - k.zu[i] + k.zu[i] / (k.u[i] - k.x[i]) * k.dx[i] : 0.0;
    }
    steps(std::max(0.995, 1.0 - mu), ap, ad);
    for (int i = 0; i < n; ++i) {
      k.x[i] += ap * k.dx[i];
      k.zl[i] += ad * k.dzl[i];
      k.zu[i] += ad * k.dzu[i];
    }
    for (int r = 0; r < m; ++r) k.y[r] += ad * k.dy[r];
  }
  for (int i = 0; i < n; ++i) xout[i] = k.fixed[i] ? k.xfix[i] : k.x[i];
  *iters = it;
  return status;
}

}  // namespace

extern "C" {

// HOST pointers.  H [n][n], g [n], A [m][n], b [m] shared by the batch; lbx / ubx [batch][n]; outputs x [batch][n], status, iters.
int hilo_cpu_qp_solve(int n, int m, const double* H, const double* g, const double* A, const double* b, int64_t batch,
                      const double* lbx, const double* ubx, double tol, int max_iter, double reg, double* x, int32_t* status,
                      int32_t* iters, int n_threads) {
  if (!H || !g || (m > 0 && (!A || !b)) || !lbx || !ubx || !x || !status || !iters || n < 1) return 1;
  if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel num_threads(n_threads)
  {
    Work k(n, m);
#pragma omp for schedule(dynamic, 8)
    for (int64_t q = 0; q < batch; ++q) {
      int it = 0;
      status[q] = solve_one(k, H, g, A, b, lbx + q * n, ubx + q * n, tol, max_iter, reg, x + q * n, &it);
      iters[q] = it;
    }
  }
  return 0;
}

}  // extern "C"
