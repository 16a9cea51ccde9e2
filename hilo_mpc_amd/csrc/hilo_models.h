// Device model zoo + explicit Runge-Kutta shooting map.
//
// The reference holds a model as CasADi expressions (hilo_mpc/modules/dynamic_model/dynamic_model.py) and
// discretises it with `Model.discretize('erk'|'rk4', order)` -> RungeKutta._explicit
// (hilo_mpc/util/modeling.py:1213-1281, tableaux :1008-1085, order->tableau :1239-1250).  Here a model is a
// functor whose `ode`/`meas` are templated on the scalar type so that the same statement serves values
// (double), first-order sensitivities (Dual<N>) and second-order directional derivatives (Jet2).
//
// ids match HILO_MODEL_* in include/hilo_hip.h.
#pragma once
#include "hilo_ad.h"

namespace hilo {

// LDS-qualified pointer types: keep the address space through non-inlined calls (ds_read/ds_write, not flat_*)
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(1))) double gbl_double;
typedef const __attribute__((address_space(1))) double gbl_cdouble;
typedef const __attribute__((address_space(3))) double lds_cdouble;

enum ModelId : int {
  MODEL_LTI = 0,          // x+ = A x + B u, y = C x; A,B,C packed in p (row-major), dims fixed per instantiation
  MODEL_TOY1D = 1,        // reference tests/test_KFs.py:548-556
  MODEL_BIOREACTOR3 = 2,  // reference tests/test_KFs.py:691-712
  MODEL_CHEMOSTAT4 = 3,   // hilo_mpc/library/models.py:163-198 closed with the rate laws of :143-148
  MODEL_PENDULUM4 = 4,    // reference tests/test_NMPC.py:12-43
  MODEL_ROBOT6 = 5,
  MODEL_CSTR3 = 6,
  MODEL_LINEAR2 = 7,      // reference tests/test_KFs.py:247-255
  MODEL_CHEMOSTAT4_GP = 8,  // chemostat4 with the growth rate `mu` substituted by a GP mean (dynamic_model.py:3040-3125)
};

// ------------------------------------------------------------------------------------------------
// Learned terms.  `Model.substitute_from(gp)` (dynamic_model.py:3040-3125) replaces a model
// parameter by `gp.predict(features)[0]`, the posterior mean  m(x*) + sum_i alpha_i k(X_i, x*)
// (inference.py:211-213).  A model whose right-hand side holds such a term receives an `ext` context:
//   gp    packed posterior (hilo_gp.hip::gp_pack_se2): [n, sf2, bias, M_0, M_1, (X_0i, X_1i, alpha_i) * n] for a
//         squared-exponential kernel (ARD or isotropic, M_d = 1/l_d^2) and a constant mean, two features
//   group lanes [gbase, gbase + gs) of the wave evaluate the model at the SAME point (they differ only in their
//         Taylor directions), so the n kernel evaluations are split among them and summed through LDS.
// ------------------------------------------------------------------------------------------------
struct NoExt {};
constexpr int GP2_HDR = 5;
constexpr int GP2_MAXN = 256;   // kernel terms of a learned two-feature mean that the solve kernel keeps in LDS (cooperative variant)
struct GpExt {
  const double* gp;   // packed posterior mean in device memory: [n, sf2, bias, M0, M1, (X0_i, X1_i, alpha_i) * n]
  lds_double* scr;    // 12 doubles per lane (partials | group totals); unused when gs == 1
  int gs, gl, gbase;
  bool idle;          // lane has no task: contributes nothing, still takes part in the exchange
  lds_cdouble* tab = nullptr;   // the same table staged in LDS by the solve kernel (or null: read `gp`)
  // value / gradient / Hessian of the mean at the Runge-Kutta stage points of ONE interval, [4][6] in LDS (or null): written by a
  // values-only evaluation (the line search's trial point), read instead of summing the kernel terms again when the derivative
  // phase runs at that very point (`use_cache`); `ctr` counts the model evaluations of the interval
  lds_double* cache = nullptr;
  int* ctr = nullptr;
  bool use_cache = false;
};

// the few type traits the device code needs, written out: under hiprtc there is no <type_traits>
template <bool V> struct bool_const { static constexpr bool value = V; };
template <class...> using void_tt = void;
template <bool C, class A, class B> struct cond { using type = A; };
template <class A, class B> struct cond<false, A, B> { using type = B; };
template <bool C, class A, class B> using cond_t = typename cond<C, A, B>::type;
template <class A, class B> struct same_type : bool_const<false> {};
template <class A> struct same_type<A, A> : bool_const<true> {};

// symbolic derivatives of a model's right-hand side (generated: hilo_models_sym.h for the zoo, hilo_mpc_amd/codegen.py for
// models written as expressions); absent -> the engine differentiates with Taylor sweeps
template <class M> struct ModelSym { static constexpr bool value = false, HAS_MEAS = false; };

template <class M, class = void> struct model_has_ext : bool_const<false> {};
template <class M> struct model_has_ext<M, void_tt<decltype(M::EXT)>> : bool_const<M::EXT> {};

// one kernel term: k = alpha exp(-1/2 (M0 d0^2 + M1 d1^2)) and its contributions to value / gradient / Hessian
template <int ORDER>
__device__ __forceinline__ void gp2_term(double x0, double x1, double al, double M0, double M1, double s, double i, double* acc) {
  const double d0 = s - x0, d1 = i - x1;
  const double t0 = M0 * d0, t1 = M1 * d1;
  const double k = al * ::exp(-0.5 * (d0 * t0 + d1 * t1));
  acc[0] += k;
  if constexpr (ORDER == 2) {
    acc[1] -= k * t0;
    acc[2] -= k * t1;
    acc[3] += k * (t0 * t0 - M0);
    acc[4] += k * t0 * t1;
    acc[5] += k * (t1 * t1 - M1);
  }
}
// this lane's share of the kernel sum: terms j0, j0 + dj, ... in that order (the order is part of the result); four terms'
// operands are requested before the first exponential so that their latency (LDS or L2) is paid once per four terms
template <int ORDER, class P>
__device__ __forceinline__ void gp2_sum(P rows, int n, double M0, double M1, double s, double i, int j0, int dj, double* acc) {
  int j = j0;
  for (; j + 3 * dj < n; j += 4 * dj) {
    double x0[4], x1[4], al[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      P r = rows + 3 * (j + q * dj);
      x0[q] = r[0]; x1[q] = r[1]; al[q] = r[2];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) gp2_term<ORDER>(x0[q], x1[q], al[q], M0, M1, s, i, acc);
  }
  for (; j < n; j += dj) {
    P r = rows + 3 * j;
    gp2_term<ORDER>(r[0], r[1], r[2], M0, M1, s, i, acc);
  }
}

// value (ORDER 0) or value / gradient / Hessian (ORDER 2: out = m, g0, g1, h00, h01, h11) of the GP mean at (s, i)
template <int ORDER>
__device__ __forceinline__ void gp2_taylor(const GpExt& e, double s, double i, double* out) {
  constexpr int NC = ORDER == 0 ? 1 : 6;
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  double sf2, bias;
  if (e.tab) {                  // wave-uniform
    lds_cdouble* g = e.tab;
    const int n = (int)g[0];
    sf2 = g[1]; bias = g[2];
    gp2_sum<ORDER>(g + GP2_HDR, n, g[3], g[4], s, i, e.idle ? n : e.gl, e.gs, acc);
  } else {
    const double* g = e.gp;
    const int n = (int)g[0];
    sf2 = g[1]; bias = g[2];
    gp2_sum<ORDER>(g + GP2_HDR, n, g[3], g[4], s, i, e.idle ? n : e.gl, e.gs, acc);
  }
  if (e.gs > 1 || e.idle) {  // wave-uniform by construction of the groups (all lanes of a cooperative pass get here)
    const int lane = threadIdx.x;
    lds_double* part = e.scr;
    lds_double* tot = e.scr + 6 * blockDim.x;
#pragma unroll
    for (int c = 0; c < NC; ++c) part[lane * 6 + c] = acc[c];
    __syncthreads();
    if (!e.idle) {
      for (int c = e.gl; c < NC; c += e.gs) {
        double t = 0.0;
        int q = 0;
        for (; q + 8 <= e.gs; q += 8) {   // eight partials requested before the first addition (same order of the sum)
          double v[8];
#pragma unroll
          for (int w = 0; w < 8; ++w) v[w] = part[(e.gbase + q + w) * 6 + c];
#pragma unroll
          for (int w = 0; w < 8; ++w) t += v[w];
        }
        for (; q < e.gs; ++q) t += part[(e.gbase + q) * 6 + c];
        tot[e.gbase * 6 + c] = t;
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = e.idle ? 0.0 : tot[e.gbase * 6 + c];
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) out[c] = sf2 * acc[c];
  out[0] += bias;
}
__device__ __forceinline__ double gp2_mean(const GpExt& e, double s, double i) {
  if (e.cache) {               // wave-uniform: keep the second-order data of this stage point for the derivative phase
    double o[6];
    gp2_taylor<2>(e, s, i, o);
    const int idx = (*e.ctr)++;
    if (idx < 4 && !e.idle && e.gl == 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) e.cache[idx * 6 + c] = o[c];
    }
    return o[0];
  }
  double o[1];
  gp2_taylor<0>(e, s, i, o);
  return o[0];
}
__device__ __forceinline__ Jet2 gp2_mean(const GpExt& e, const Jet2& s, const Jet2& i) {
  double o[6];
  if (e.use_cache) {           // wave-uniform
    const int idx = (*e.ctr)++;
#pragma unroll
    for (int c = 0; c < 6; ++c) o[c] = e.cache[(idx < 4 ? idx : 3) * 6 + c];
  } else {
    gp2_taylor<2>(e, s.v, i.v, o);
  }
  return Jet2(o[0], o[1] * s.a + o[2] * i.a,
              o[1] * s.b + o[2] * i.b + o[3] * s.a * s.a + 2.0 * o[4] * s.a * i.a + o[5] * i.a * i.a);
}

// Learned term of a run-time compiled model: posterior mean of a squared-exponential GP over any number of features, every
// lane sums all n kernel terms itself in the scalar type it is evaluated with (values, Dual, Jet2) - general, not fast; the
// two-feature zoo variant above shares the sum among the lanes of a stage.  `feat` holds the GP's features in ITS order
// (gp.features); pack = hilo_gp.hip::gp_pack_se.
template <class T>
__device__ __forceinline__ T gp_se_mean(const double* g, const T* feat) {
  const int n = (int)g[0], na = (int)g[1];
  const double sf2 = g[2], bias = g[3];
  const double* ad = g + 4;
  const double* Md = g + 4 + na;
  const double* r = g + 4 + 2 * na;
  T acc = T(0.0);
  for (int i = 0; i < n; ++i, r += na + 1) {
    T d2 = T(0.0);
    for (int q = 0; q < na; ++q) {
      const T df = feat[(int)ad[q]] - r[q];
      d2 = d2 + Md[q] * (df * df);
    }
    acc = acc + r[na] * exp(-0.5 * d2);
  }
  return bias + sf2 * acc;
}

// d mean / d feature j of the same posterior (the GP Jacobian of the stochastic NMPC's covariance propagation, mpc.py:2534-2539):
// sf2 sum_i alpha_i exp(-d2_i / 2) (-M_j (x_j - X_ij)); zero for a feature outside the kernel's active dimensions.
template <class T>
__device__ __forceinline__ T gp_se_dmean(const double* g, const T* feat, int j) {
  const int n = (int)g[0], na = (int)g[1];
  const double sf2 = g[2];
  const double* ad = g + 4;
  const double* Md = g + 4 + na;
  const double* r = g + 4 + 2 * na;
  int qj = -1;
  for (int q = 0; q < na; ++q)
    if ((int)ad[q] == j) qj = q;
  T acc = T(0.0);
  if (qj < 0) return acc;
  for (int i = 0; i < n; ++i, r += na + 1) {
    T d2 = T(0.0);
    for (int q = 0; q < na; ++q) {
      const T df = feat[(int)ad[q]] - r[q];
      d2 = d2 + Md[q] * (df * df);
    }
    acc = acc + (r[na] * Md[qj]) * ((r[qj] - feat[j]) * exp(-0.5 * d2));
  }
  return sf2 * acc;
}

// Posterior variance INCLUDING the noise variance (`gp.predict(x)[1]` with noise_free=False, gp.py:699-713; inference.py:
// 214-216: k** - v^T v, v = L^-1 k*): the pack carries sn2 and L^-1 (row-major n x n) behind the mean's data.  Every lane keeps
// k* (n <= GP_VAR_MAX values of T) in private memory: general, not fast (indexed at run time the array lives in scratch: 6 KB per
// lane for second-order Taylor numbers - sized for the 200 training points of BASELINE configuration 4's learned term).
constexpr int GP_VAR_MAX = 256;
template <class T>
__device__ __forceinline__ T gp_se_var(const double* g, const T* feat) {
  const int n = (int)g[0], na = (int)g[1];
  const double sf2 = g[2];
  const double* ad = g + 4;
  const double* Md = g + 4 + na;
  const double* r = g + 4 + 2 * na;
  const double* tail = r + (size_t)n * (na + 1);
  const double sn2 = tail[0];
  const double* Li = tail + 1;
  T ks[GP_VAR_MAX];
  for (int i = 0; i < n && i < GP_VAR_MAX; ++i, r += na + 1) {
    T d2 = T(0.0);
    for (int q = 0; q < na; ++q) {
      const T df = feat[(int)ad[q]] - r[q];
      d2 = d2 + Md[q] * (df * df);
    }
    ks[i] = sf2 * exp(-0.5 * d2);
  }
  T acc = T(0.0);
  for (int i = 0; i < n && i < GP_VAR_MAX; ++i, Li += n) {
    T v = T(0.0);
    for (int k = 0; k <= i; ++k) v = v + Li[k] * ks[k];
    acc = acc + v * v;
  }
  return (sf2 + sn2) - acc;
}

// ---- semi-explicit index-1 DAE  dx/dt = f(x, z, u, p),  0 = g(x, z, u, p)  of a run-time compiled model (codegen.py::
// dae_model_source: M::NZ, M::z_guess, M::ode_z, M::alg, M::alg_jz, M::meas_z) -----------------------------------------------
// Newton's method on g(x, z, u, p) = 0 for z IN THE SCALAR TYPE Z: started from the model's constant guess, iterated until the
// value of the residual is at round-off, then two more sweeps - in Taylor / dual arithmetic every sweep after the values have
// converged makes one more derivative order of the implicit function z = zeta(x, u, p) exact.
template <class M, class Z, class P>
__device__ __forceinline__ void dae_solve(const Z* x, const Z* u, const P* p, Z* z) {
  constexpr int NZ = M::NZ;
  if constexpr (!same_type<Z, double>::value && same_type<P, double>::value) {
    // derivative-carrying scalar types: the VALUES by the iteration below in plain doubles, then two sweeps z <- z - Jv^-1 g(x, z, u)
    // in the scalar type with the inverse of the value Jacobian (a double matrix): sweep k makes the order-k coefficients of the
    // implicit function exact - no division in Taylor / dual arithmetic, no iteration in it
    constexpr int NX = M::NX, NU = M::NU;
    double xv[NX], uv[NU > 0 ? NU : 1], zv[NZ], Jv[NZ * NZ], Ji[NZ * NZ];
#pragma unroll
    for (int i = 0; i < NX; ++i) xv[i] = valof(x[i]);
#pragma unroll
    for (int i = 0; i < NU; ++i) uv[i] = valof(u[i]);
    dae_solve<M, double, P>(xv, uv, p, zv);
    M::alg_jz(xv, zv, uv, p, Jv);
#pragma unroll
    for (int i = 0; i < NZ * NZ; ++i) Ji[i] = (i / NZ == i % NZ) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < NZ; ++c) {        // Gauss-Jordan with partial pivoting (static indices: conditional swaps)
#pragma unroll
      for (int q = c + 1; q < NZ; ++q) {
        if (fabs(Jv[q * NZ + c]) > fabs(Jv[c * NZ + c])) {
#pragma unroll
          for (int j = 0; j < NZ; ++j) {
            double t = Jv[c * NZ + j]; Jv[c * NZ + j] = Jv[q * NZ + j]; Jv[q * NZ + j] = t;
            t = Ji[c * NZ + j]; Ji[c * NZ + j] = Ji[q * NZ + j]; Ji[q * NZ + j] = t;
          }
        }
      }
      const double ip = rcp_fast(Jv[c * NZ + c]);
#pragma unroll
      for (int j = 0; j < NZ; ++j) { Jv[c * NZ + j] *= ip; Ji[c * NZ + j] *= ip; }
#pragma unroll
      for (int q = 0; q < NZ; ++q) {
        if (q != c) {
          const double f = Jv[q * NZ + c];
#pragma unroll
          for (int j = 0; j < NZ; ++j) { Jv[q * NZ + j] -= f * Jv[c * NZ + j]; Ji[q * NZ + j] -= f * Ji[c * NZ + j]; }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NZ; ++i) z[i] = Z(zv[i]);
#pragma unroll
    for (int sweep = 0; sweep < 2; ++sweep) {
      Z r[NZ];
      M::alg(x, z, u, p, r);
#pragma unroll
      for (int i = 0; i < NZ; ++i) {
        Z acc = z[i];
#pragma unroll
        for (int j = 0; j < NZ; ++j) acc = acc - Ji[i * NZ + j] * r[j];
        z[i] = acc;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NZ; ++i) z[i] = Z(M::z_guess(i));
  int done = 0;
  for (int it = 0; it < 40; ++it) {
    if (!__any((int)(done < 2))) break;   // wave-uniform exit (see hilo_colloc.h::solve): further sweeps of a finished lane are harmless
    Z r[NZ], J[NZ * NZ];
    M::alg(x, z, u, p, r);
    M::alg_jz(x, z, u, p, J);
    double rmax = 0.0, zmax = 0.0;
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      rmax = fmax(rmax, fabs(valof(r[i])));
      zmax = fmax(zmax, fabs(valof(z[i])));
    }
    // J dz = r by Gaussian elimination with partial pivoting on the values (static indices: conditional swaps)
#pragma unroll
    for (int c = 0; c < NZ; ++c) {
#pragma unroll
      for (int q = c + 1; q < NZ; ++q) {
        if (fabs(valof(J[q * NZ + c])) > fabs(valof(J[c * NZ + c]))) {
#pragma unroll
          for (int j = c; j < NZ; ++j) { const Z t = J[c * NZ + j]; J[c * NZ + j] = J[q * NZ + j]; J[q * NZ + j] = t; }
          const Z t = r[c]; r[c] = r[q]; r[q] = t;
        }
      }
      const Z ip = 1.0 / J[c * NZ + c];
#pragma unroll
      for (int q = c + 1; q < NZ; ++q) {
        const Z f = J[q * NZ + c] * ip;
#pragma unroll
        for (int j = c + 1; j < NZ; ++j) J[q * NZ + j] = J[q * NZ + j] - f * J[c * NZ + j];
        r[q] = r[q] - f * r[c];
      }
    }
#pragma unroll
    for (int c = NZ - 1; c >= 0; --c) {
      Z acc = r[c];
#pragma unroll
      for (int j = c + 1; j < NZ; ++j) acc = acc - J[c * NZ + j] * r[j];
      r[c] = acc / J[c * NZ + c];
    }
#pragma unroll
    for (int i = 0; i < NZ; ++i) z[i] = z[i] - r[i];
    if (rmax <= 1e-13 * (1.0 + zmax)) ++done;   // the values had converged before this sweep
  }
}

// the ODE the engine sees: dx/dt = f(x, zeta(x, u, p), u, p); likewise the measurement map
template <class M, class T, class U, class P>
__device__ __forceinline__ void dae_ode(const T* x, const U* u, const P* p, T* dx) {
  using Z = decltype(*x + *u);
  static_assert(same_type<Z, T>::value, "dae_ode: the state's scalar type must carry the input's");
  constexpr int NX = M::NX, NU = M::NU;
  Z uz[NU > 0 ? NU : 1], z[M::NZ];
#pragma unroll
  for (int i = 0; i < NU; ++i) uz[i] = Z(u[i]);
  if constexpr (M::ODE_USES_Z) dae_solve<M>(x, uz, p, z);   // (an algebraic state that only constraints / outputs name: not solved for here)
  else {
#pragma unroll
    for (int i = 0; i < M::NZ; ++i) z[i] = Z(0.0);
  }
  M::ode_z(x, z, uz, p, dx);
  (void)NX;
}
template <class M, class T, class U, class P>
__device__ __forceinline__ void dae_meas(const T* x, const U* u, const P* p, T* y) {
  using Z = decltype(*x + *u);
  static_assert(same_type<Z, T>::value, "dae_meas: the state's scalar type must carry the input's");
  constexpr int NU = M::NU;
  Z uz[NU > 0 ? NU : 1], z[M::NZ];
#pragma unroll
  for (int i = 0; i < NU; ++i) uz[i] = Z(u[i]);
  if constexpr (M::MEAS_USES_Z) dae_solve<M>(x, uz, p, z);
  else {
#pragma unroll
    for (int i = 0; i < M::NZ; ++i) z[i] = Z(0.0);
  }
  M::meas_z(x, z, uz, p, y);
}
template <class M, class = void> struct model_nz { static constexpr int value = 0; };
template <class M> struct model_nz<M, void_tt<decltype(M::NZ)>> { static constexpr int value = M::NZ; };

// ---- tests/test_KFs.py:247-255: dx1 = -k1 x1 + u, dx2 = k1 x1 - k2 x2, y = x2 -------------------------
struct Linear2 {
  static constexpr int NX = 2, NU = 1, NP = 2, NY = 1;
  static constexpr bool DISCRETE = false;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P* p, double, T* dx) {
    dx[0] = -1.0 * (p[0] * x[0]) + u[0];
    dx[1] = p[0] * x[0] - p[1] * x[1];
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) { y[0] = x[1]; }
};

// ---- tests/test_KFs.py:548-556: x+ = x/2 + 25 dt x/(1+x^2), y = x^2/20 (discrete) ---------------------
struct Toy1D {
  static constexpr int NX = 1, NU = 0, NP = 0, NY = 1;
  static constexpr bool DISCRETE = true;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U*, const P*, double dt, T* xn) {
    xn[0] = x[0] / 2.0 + 25.0 * dt * x[0] / (1.0 + x[0] * x[0]);
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) { y[0] = x[0] * x[0] / 20.0; }
};

// ---- tests/test_KFs.py:691-712: bioreactor T, cB, cS; p = [alpha, T_amb, mu_0, mu_1, K, Y]; u = D ----
struct Bioreactor3 {
  static constexpr int NX = 3, NU = 1, NP = 6, NY = 2;
  static constexpr bool DISCRETE = false;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P* p, double, T* dx) {
    const T r = (p[2] + p[3] * x[0]) * x[2] * x[1] / (p[4] + x[2]);
    dx[0] = p[0] * (p[1] - x[0]);
    dx[1] = r - u[0] * x[1];
    dx[2] = -1.0 * (r / p[5]) - u[0] * x[2];
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) { y[0] = x[0]; y[1] = x[1]; }
};

// ---- CSTR-sized benchmark model: states X,S,P,I; inputs DS,DI; p = [Sf,If,ISF,IRF] ----
struct Chemostat4 {
  static constexpr int NX = 4, NU = 2, NP = 4, NY = 2;
  static constexpr bool DISCRETE = false;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P* p, double, T* dx) {
    const T X = x[0], S = x[1], Pr = x[2], I = x[3];
    const T phi = 0.407 * S / (0.108 + S + S * S / 14814.0);
    const T mu = phi * (p[2] + 0.22 * p[3] / (0.22 + I));
    const T Rs = 2.0 * mu;
    const T Rfp = phi * (0.0005 + I) / (0.022 + I);
    const T D = T(u[0]) + u[1];
    dx[0] = mu * X - D * X;
    dx[1] = -1.0 * (Rs * X) - D * S + u[0] * p[0];
    dx[2] = Rfp * X - D * Pr;
    dx[3] = -1.0 * (D * I) + u[1] * p[1];
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) { y[0] = x[0]; y[1] = x[2]; }
};

// ---- Chemostat4 with the growth rate of the biomass balance supplied by a GP over the features (S, I) -----
// (`nmpc_hybrid_bio.ipynb`: `model.substitute_from(gp)` with gp.labels = ['mu']; in the 'simple' model mu only enters
// dX/dt, hilo_mpc/library/models.py:163-198; Rs and Rfp keep the closed-form rate laws of Chemostat4)
struct Chemostat4Gp {
  static constexpr int NX = 4, NU = 2, NP = 4, NY = 2;
  static constexpr bool DISCRETE = false;
  static constexpr bool EXT = true;
  template <class T, class U, class P>
  __device__ static void ode(const T* x, const U* u, const P* p, double, T* dx, const GpExt& ext) {
    const T X = x[0], S = x[1], Pr = x[2], I = x[3];
    const T phi = 0.407 * S / (0.108 + S + S * S / 14814.0);
    const T Rs = 2.0 * (phi * (p[2] + 0.22 * p[3] / (0.22 + I)));
    const T Rfp = phi * (0.0005 + I) / (0.022 + I);
    const T mu = gp2_mean(ext, S, I);
    const T D = T(u[0]) + u[1];
    dx[0] = mu * X - D * X;
    dx[1] = -1.0 * (Rs * X) - D * S + u[0] * p[0];
    dx[2] = Rfp * X - D * Pr;
    dx[3] = -1.0 * (D * I) + u[1] * p[1];
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) { y[0] = x[0]; y[1] = x[2]; }
};

// ---- tests/test_NMPC.py:12-43: cart-pendulum x,v,theta,omega; input F; all states measured ---------------
struct Pendulum4 {
  static constexpr int NX = 4, NU = 1, NP = 0, NY = 4;
  static constexpr bool DISCRETE = false;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P*, double, T* dx) {
    const double M = 5.0, m = 1.0, l = 1.0, g = 9.81;
    const T s = sin(x[2]), c = cos(x[2]);
    const T dv = 1.0 / (M + m - m * c) * (m * g * s - m * l * s * x[3] * x[3] + u[0]);
    dx[0] = x[1];
    dx[1] = dv;
    dx[2] = x[3];
    dx[3] = 1.0 / l * (dv * c + g * s);
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) {
    y[0] = x[0]; y[1] = x[1]; y[2] = x[2]; y[3] = x[3];
  }
};

// ---- planar mobile robot with heading: px,vx,py,vy,psi,omega; inputs a (along the heading), alpha ----
struct Robot6 {
  static constexpr int NX = 6, NU = 2, NP = 0, NY = 2;
  static constexpr bool DISCRETE = false;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P*, double, T* dx) {
    dx[0] = x[1];
    dx[1] = u[0] * cos(x[4]);
    dx[2] = x[3];
    dx[3] = u[0] * sin(x[4]);
    dx[4] = x[5];
    dx[5] = T(u[1]);
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) { y[0] = x[0]; y[1] = x[2]; }
};

// ---- docs/docsource/examples/CSTR_Example.ipynb cell 6 (`true_plant_model`, constants of cell 4): reversible exothermic
// reaction A <-> B in a cooled CSTR; states C_A, C_B, T; input Q; measurement r (the reaction rate).  The statements
// follow the notebook's expression structure term by term (the same tree the run-time compiled model is emitted from).
struct Cstr3 {
  static constexpr int NX = 3, NU = 1, NP = 0, NY = 1;
  static constexpr bool DISCRETE = false;
  template <class T>
  HD static T rate(const T* x) {
    return 5000.0 * exp(-10000.0 / (1.987 * x[2])) * x[0] - 1000000.0 * exp(-15000.0 / (1.987 * x[2])) * x[1];
  }
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P*, double, T* dx) {
    const T r = rate(x);
    dx[0] = 0.016666666666666666 * (1.0 - x[0]) - r;
    dx[1] = -0.016666666666666666 * x[1] + r;
    dx[2] = (-1.0 * (-5000.0 * r)) / 1000.0 + 0.016666666666666666 * (400.0 - x[2]) + u[0] / 100000.0;
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P*, double, T* y) { y[0] = rate(x); }
};

// ---- LTI with compile-time dims; p = [A (NX*NX) | B (NX*NU) | C (NY*NX)] row-major ------------------------
template <int NX_, int NU_, int NY_>
struct Lti {
  static constexpr int NX = NX_, NU = NU_, NP = NX_ * NX_ + NX_ * NU_ + NY_ * NX_, NY = NY_;
  static constexpr bool DISCRETE = true;
  template <class T, class U, class P>
  HD static void ode(const T* x, const U* u, const P* p, double, T* xn) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      T s = T(0.0);
#pragma unroll
      for (int j = 0; j < NX; ++j) s = s + p[i * NX + j] * x[j];
#pragma unroll
      for (int j = 0; j < NU; ++j) s = s + p[NX * NX + i * NU + j] * u[j];
      xn[i] = s;
    }
  }
  template <class T, class U, class P>
  HD static void meas(const T* x, const U*, const P* p, double, T* y) {
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      T s = T(0.0);
#pragma unroll
      for (int j = 0; j < NX; ++j) s = s + p[NX * NX + NX * NU + i * NX + j] * x[j];
      y[i] = s;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Explicit Runge-Kutta map, modeling.py:1213-1281.  order 1..4 -> forward Euler / midpoint / Kutta / classic.
// Written as four guarded, fully unrolled stages so `order` can be a run-time (wave-uniform) value.
// ------------------------------------------------------------------------------------------------
// Tableau entries as selects on the (wave-uniform) order so that they stay in scalar registers.
template <int I, int J> HD double erk_a(int order) {
  if constexpr (I == 1 && J == 0) return order >= 2 ? 0.5 : 0.0;
  else if constexpr (I == 2 && J == 0) return order == 3 ? -1.0 : 0.0;
  else if constexpr (I == 2 && J == 1) return order == 3 ? 2.0 : (order == 4 ? 0.5 : 0.0);
  else if constexpr (I == 3 && J == 2) return order == 4 ? 1.0 : 0.0;
  else return 0.0;
}
template <int I> HD double erk_b(int order) {
  if constexpr (I == 0) return order == 1 ? 1.0 : (order == 2 ? 0.0 : 1.0 / 6);
  else if constexpr (I == 1) return order == 2 ? 1.0 : (order == 3 ? 2.0 / 3 : (order == 4 ? 1.0 / 3 : 0.0));
  else if constexpr (I == 2) return order == 3 ? 1.0 / 6 : (order == 4 ? 1.0 / 3 : 0.0);
  else return order == 4 ? 1.0 / 6 : 0.0;
}

template <class M, int I, class T, class U, class P, class E>
HD void erk_stage(int order, const T* x, const U* u, const P* p, double h, T (*k)[M::NX], const E& ext) {
  constexpr int NX = M::NX;
  if (I < order) {
    T xi[NX];
#pragma unroll
    for (int s = 0; s < NX; ++s) {
      T acc = x[s];
      if constexpr (I >= 1) acc = acc + (h * erk_a<I, 0>(order)) * k[0][s];
      if constexpr (I >= 2) acc = acc + (h * erk_a<I, 1>(order)) * k[1][s];
      if constexpr (I >= 3) acc = acc + (h * erk_a<I, 2>(order)) * k[2][s];
      xi[s] = acc;
    }
    if constexpr (model_has_ext<M>::value) M::ode(xi, u, p, h, k[I], ext);
    else M::ode(xi, u, p, h, k[I]);
  } else {
#pragma unroll
    for (int s = 0; s < NX; ++s) k[I][s] = T(0.0);
  }
}

template <class M, class T, class U, class P, class E>
HD void erk_step(int order, const T* x, const U* u, const P* p, double h, T* xn, const E& ext) {
  constexpr int NX = M::NX;
  T k[4][NX];
  erk_stage<M, 0>(order, x, u, p, h, k, ext);
  erk_stage<M, 1>(order, x, u, p, h, k, ext);
  erk_stage<M, 2>(order, x, u, p, h, k, ext);
  erk_stage<M, 3>(order, x, u, p, h, k, ext);
#pragma unroll
  for (int s = 0; s < NX; ++s)
    xn[s] = x[s] + (h * erk_b<0>(order)) * k[0][s] + (h * erk_b<1>(order)) * k[1][s] +
            (h * erk_b<2>(order)) * k[2][s] + (h * erk_b<3>(order)) * k[3][s];
}

// The classic tableau with ONE slope alive: the weighted sum is accumulated stage by stage - the same terms in the same order as
// erk_step's final sum - and the stage points leave out the tableau's zero entries (`x + (h 0) k0 + ...`: exact zeros that strict
// fp64 arithmetic may not drop and that keep every earlier slope alive).  Same results as erk_step<M>(4, ...), a third of its live
// values: with forward-mode types (Dual<N>, Jet2) the four slopes are most of a kernel's registers.
template <class M, class T, class U, class P, class E>
HD void rk4_classic(const T* x, const U* u, const P* p, double h, T* xn, const E& ext) {
  constexpr int NX = M::NX;
  T k[NX], xi[NX], acc[NX];
  auto f = [&](const T* at) {
    if constexpr (model_has_ext<M>::value) M::ode(at, u, p, h, k, ext);
    else M::ode(at, u, p, h, k);
  };
  f(x);
#pragma unroll
  for (int s = 0; s < NX; ++s) { acc[s] = x[s] + (h * (1.0 / 6)) * k[s]; xi[s] = x[s] + (h * 0.5) * k[s]; }
  f(xi);
#pragma unroll
  for (int s = 0; s < NX; ++s) { acc[s] = acc[s] + (h * (1.0 / 3)) * k[s]; xi[s] = x[s] + (h * 0.5) * k[s]; }
  f(xi);
#pragma unroll
  for (int s = 0; s < NX; ++s) { acc[s] = acc[s] + (h * (1.0 / 3)) * k[s]; xi[s] = x[s] + (h * 1.0) * k[s]; }
  f(xi);
#pragma unroll
  for (int s = 0; s < NX; ++s) xn[s] = acc[s] + (h * (1.0 / 6)) * k[s];
}

// one sampling interval of the shooting map: discrete models are evaluated directly (mpc.py:1381-1389,:1665),
// continuous ones through ERK of the requested order with `nsub` equal sub-steps
template <class M, class T, class U, class P, class E = NoExt>
HD void model_step(int order, int nsub, const T* x, const U* u, const P* p, double dt, T* xn, const E& ext = E()) {
  if constexpr (M::DISCRETE) {
    if constexpr (model_has_ext<M>::value) M::ode(x, u, p, dt, xn, ext);
    else M::ode(x, u, p, dt, xn);
  } else {
    constexpr int NX = M::NX;
    if (order == 4 && nsub == 1) {  // the common recipe `model.discretize('rk4')`
#ifdef HILO_RK4_GENERIC               // developer builds: the generic map with the tableau folded at compile time
      erk_step<M>(4, x, u, p, dt, xn, ext);
#else
      rk4_classic<M>(x, u, p, dt, xn, ext);
#endif
      return;
    }
    const double h = dt / nsub;
    T xc[NX];
#pragma unroll
    for (int s = 0; s < NX; ++s) xc[s] = x[s];
    for (int it = 0; it < nsub; ++it) {
      T xt[NX];
      erk_step<M>(order, xc, u, p, h, xt, ext);
#pragma unroll
      for (int s = 0; s < NX; ++s) xc[s] = xt[s];
    }
#pragma unroll
    for (int s = 0; s < NX; ++s) xn[s] = xc[s];
  }
}

}  // namespace hilo

#include "hilo_models_sym.h"
