// Forward-mode automatic differentiation scalars used by every model-evaluating kernel.
//
// The reference obtains all derivatives from CasADi's symbolic AD. On the GPU a model is a
// C++ functor templated on its scalar type, and derivatives come from instantiating it with
//   Dual<N>   value + N first-order tangents   (EKF Jacobians F, H; shooting sensitivities A_k, B_k)
//   Jet2      value + first + second derivative along ONE direction (univariate Taylor, order 2), from which
//             the Lagrangian-Hessian blocks are recovered by polarisation: H_ij = (q(e_i+e_j) - q(e_i) - q(e_j))/2
// Everything lives in registers: all loops have compile-time bounds and are fully unrolled.
#pragma once
// Under hiprtc (run-time compilation of user models, hilo_jit.hip) the HIP device API and the math functions are
// predeclared and no system header is reachable: every #include of a system header is guarded.
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <math.h>
#else
#ifndef INFINITY
#define INFINITY (__builtin_huge_val())
#endif
#endif

#define HD __host__ __device__ __forceinline__

namespace hilo {

// ------------------------------------------------------------------------------------------------
// Dual<N>: f, df/ds_1..N
// ------------------------------------------------------------------------------------------------
template <int N>
struct Dual {
  double v;
  double d[N];
  HD Dual() {}
  HD Dual(double c) : v(c) {
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = 0.0;
  }
};

template <int N> HD Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
template <int N> HD Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
template <int N> HD Dual<N> operator-(const Dual<N>& a) {
  Dual<N> r; r.v = -a.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
  return r;
}
template <int N> HD Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
HD double rcp_fast(double x);   // (below: v_rcp_f64 + two Newton steps, <= 1 ulp; a dependent chain of 5 instructions where the IEEE
                                // sequence has 12 - the filters' team kernels wait for twelve divisions in a row per step)
template <int N> HD Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; const double ib = rcp_fast(b.v); r.v = a.v * ib;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
template <int N> HD Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> HD Dual<N> operator+(double b, const Dual<N>& a) { Dual<N> r = a; r.v += b; return r; }
template <int N> HD Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> HD Dual<N> operator-(double b, const Dual<N>& a) { Dual<N> r = -a; r.v += b; return r; }
template <int N> HD Dual<N> operator*(const Dual<N>& a, double b) {
  Dual<N> r; r.v = a.v * b;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b;
  return r;
}
template <int N> HD Dual<N> operator*(double b, const Dual<N>& a) { return a * b; }
template <int N> HD Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> HD Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }

template <int N> HD Dual<N> chain(const Dual<N>& a, double f, double df) {
  Dual<N> r; r.v = f;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = df * a.d[i];
  return r;
}
// sine and cosine of one argument from ONE argument reduction (a model that names sin(psi) and cos(psi) asks for both twice in
// derivative arithmetic: four reductions of ~100 instructions each per evaluation with the separate calls)
HD void sin_cos(double x, double& s, double& c) {
#if defined(__HIP_DEVICE_COMPILE__)
  ::sincos(x, &s, &c);
#else
  s = ::sin(x);
  c = ::cos(x);
#endif
}
template <int N> HD Dual<N> sin(const Dual<N>& a) { double s, c; sin_cos(a.v, s, c); return chain(a, s, c); }
template <int N> HD Dual<N> cos(const Dual<N>& a) { double s, c; sin_cos(a.v, s, c); return chain(a, c, -s); }
template <int N> HD Dual<N> exp(const Dual<N>& a) { const double e = ::exp(a.v); return chain(a, e, e); }
template <int N> HD Dual<N> log(const Dual<N>& a) { return chain(a, ::log(a.v), 1.0 / a.v); }
template <int N> HD Dual<N> sqrt(const Dual<N>& a) { const double s = ::sqrt(a.v); return chain(a, s, 0.5 / s); }
template <int N> HD Dual<N> sq(const Dual<N>& a) { return chain(a, a.v * a.v, 2.0 * a.v); }
// the rest of the reference's function table (hilo_mpc/util/parsing.py:36-58); non-smooth ones as CasADi differentiates them:
// d|a| = sign(a) da, d sign(a) = 0
HD double sgn_of(double v) { return v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0); }
template <int N> HD Dual<N> log10(const Dual<N>& a) { return chain(a, ::log10(a.v), 0.4342944819032518 / a.v); }
template <int N> HD Dual<N> fabs(const Dual<N>& a) { return chain(a, ::fabs(a.v), sgn_of(a.v)); }
template <int N> HD Dual<N> sign(const Dual<N>& a) { return chain(a, sgn_of(a.v), 0.0); }
template <int N> HD Dual<N> asin(const Dual<N>& a) { return chain(a, ::asin(a.v), 1.0 / ::sqrt(1.0 - a.v * a.v)); }
template <int N> HD Dual<N> acos(const Dual<N>& a) { return chain(a, ::acos(a.v), -1.0 / ::sqrt(1.0 - a.v * a.v)); }
template <int N> HD Dual<N> atan(const Dual<N>& a) { return chain(a, ::atan(a.v), 1.0 / (1.0 + a.v * a.v)); }
template <int N> HD Dual<N> asinh(const Dual<N>& a) { return chain(a, ::asinh(a.v), 1.0 / ::sqrt(a.v * a.v + 1.0)); }
template <int N> HD Dual<N> acosh(const Dual<N>& a) { return chain(a, ::acosh(a.v), 1.0 / ::sqrt(a.v * a.v - 1.0)); }
template <int N> HD Dual<N> atanh(const Dual<N>& a) { return chain(a, ::atanh(a.v), 1.0 / (1.0 - a.v * a.v)); }
template <int N> HD Dual<N> atan2(const Dual<N>& y, const Dual<N>& x) {
  Dual<N> r; r.v = ::atan2(y.v, x.v);
  const double ir = 1.0 / (x.v * x.v + y.v * y.v);
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * ir;
  return r;
}
template <int N> HD Dual<N> atan2(const Dual<N>& y, double x) { return atan2(y, Dual<N>(x)); }
template <int N> HD Dual<N> atan2(double y, const Dual<N>& x) { return atan2(Dual<N>(y), x); }

// ------------------------------------------------------------------------------------------------
// Jet2: univariate Taylor coefficients along one direction: f(t) = v + a t + (b/2) t^2 (a = f', b = f'')
// ------------------------------------------------------------------------------------------------
// reciprocal for the Taylor arithmetic: v_rcp_f64 + two Newton steps (5 instructions, <= 1 ulp) instead of the 12-instruction
// IEEE division sequence; a zero divisor gives NaN, which the line search rejects like inf
HD double rcp_fast(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}

// A double whose division is x * rcp_fast(y) (1-2 ulp) instead of the IEEE sequence (a ~150-cycle dependent chain): the scalar
// type the engine's derivative phase evaluates model right-hand sides with - the same rounding the Taylor numbers below use
// for their value component.
struct FastD {
  double v;
  HD FastD() {}
  HD FastD(double c) : v(c) {}
};
HD FastD operator+(FastD a, FastD b) { return FastD(a.v + b.v); }
HD FastD operator-(FastD a, FastD b) { return FastD(a.v - b.v); }
HD FastD operator-(FastD a) { return FastD(-a.v); }
HD FastD operator*(FastD a, FastD b) { return FastD(a.v * b.v); }
HD FastD operator/(FastD a, FastD b) { return FastD(a.v * rcp_fast(b.v)); }
HD FastD operator+(FastD a, double c) { return FastD(a.v + c); }
HD FastD operator+(double c, FastD a) { return FastD(c + a.v); }
HD FastD operator-(FastD a, double c) { return FastD(a.v - c); }
HD FastD operator-(double c, FastD a) { return FastD(c - a.v); }
HD FastD operator*(FastD a, double c) { return FastD(a.v * c); }
HD FastD operator*(double c, FastD a) { return FastD(c * a.v); }
HD FastD operator/(FastD a, double c) { return FastD(a.v * (1.0 / c)); }
HD FastD operator/(double c, FastD a) { return FastD(c * rcp_fast(a.v)); }
HD FastD sin(FastD a) { return FastD(::sin(a.v)); }
HD FastD cos(FastD a) { return FastD(::cos(a.v)); }
HD FastD exp(FastD a) { return FastD(::exp(a.v)); }
HD FastD log(FastD a) { return FastD(::log(a.v)); }
HD FastD sqrt(FastD a) { return FastD(::sqrt(a.v)); }
HD FastD sq(FastD a) { return FastD(a.v * a.v); }
HD FastD log10(FastD a) { return FastD(::log10(a.v)); }
HD FastD fabs(FastD a) { return FastD(::fabs(a.v)); }
HD FastD sign(FastD a) { return FastD(sgn_of(a.v)); }
HD FastD asin(FastD a) { return FastD(::asin(a.v)); }
HD FastD acos(FastD a) { return FastD(::acos(a.v)); }
HD FastD atan(FastD a) { return FastD(::atan(a.v)); }
HD FastD asinh(FastD a) { return FastD(::asinh(a.v)); }
HD FastD acosh(FastD a) { return FastD(::acosh(a.v)); }
HD FastD atanh(FastD a) { return FastD(::atanh(a.v)); }
HD FastD atan2(FastD y, FastD x) { return FastD(::atan2(y.v, x.v)); }
HD FastD atan2(FastD y, double x) { return FastD(::atan2(y.v, x)); }
HD FastD atan2(double y, FastD x) { return FastD(::atan2(y, x.v)); }

struct Jet2 {
  double v, a, b;
  HD Jet2() {}
  HD Jet2(double c) : v(c), a(0.0), b(0.0) {}
  HD Jet2(double v_, double a_, double b_) : v(v_), a(a_), b(b_) {}
};
HD Jet2 operator+(const Jet2& x, const Jet2& y) { return Jet2(x.v + y.v, x.a + y.a, x.b + y.b); }
HD Jet2 operator-(const Jet2& x, const Jet2& y) { return Jet2(x.v - y.v, x.a - y.a, x.b - y.b); }
HD Jet2 operator-(const Jet2& x) { return Jet2(-x.v, -x.a, -x.b); }
HD Jet2 operator*(const Jet2& x, const Jet2& y) {
  return Jet2(x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + 2.0 * x.a * y.a + x.v * y.b);
}
HD Jet2 operator/(const Jet2& x, const Jet2& y) {
  const double iy = rcp_fast(y.v);
  const double v = x.v * iy;
  const double a = (x.a - v * y.a) * iy;
  const double b = (x.b - 2.0 * a * y.a - v * y.b) * iy;
  return Jet2(v, a, b);
}
HD Jet2 operator+(const Jet2& x, double c) { return Jet2(x.v + c, x.a, x.b); }
HD Jet2 operator+(double c, const Jet2& x) { return Jet2(x.v + c, x.a, x.b); }
HD Jet2 operator-(const Jet2& x, double c) { return Jet2(x.v - c, x.a, x.b); }
HD Jet2 operator-(double c, const Jet2& x) { return Jet2(c - x.v, -x.a, -x.b); }
HD Jet2 operator*(const Jet2& x, double c) { return Jet2(x.v * c, x.a * c, x.b * c); }
HD Jet2 operator*(double c, const Jet2& x) { return Jet2(x.v * c, x.a * c, x.b * c); }
HD Jet2 operator/(const Jet2& x, double c) { return x * (1.0 / c); }
HD Jet2 operator/(double c, const Jet2& y) { return Jet2(c) / y; }
// g(f): g' f', g'' f'^2 + g' f''
HD Jet2 chain(const Jet2& x, double g, double g1, double g2) {
  return Jet2(g, g1 * x.a, g2 * x.a * x.a + g1 * x.b);
}
HD Jet2 sin(const Jet2& x) { double s, c; sin_cos(x.v, s, c); return chain(x, s, c, -s); }
HD Jet2 cos(const Jet2& x) { double s, c; sin_cos(x.v, s, c); return chain(x, c, -s, -c); }
HD Jet2 exp(const Jet2& x) { const double e = ::exp(x.v); return chain(x, e, e, e); }
HD Jet2 log(const Jet2& x) { const double i = 1.0 / x.v; return chain(x, ::log(x.v), i, -i * i); }
HD Jet2 sqrt(const Jet2& x) { const double s = ::sqrt(x.v); return chain(x, s, 0.5 / s, -0.25 / (s * x.v)); }
HD Jet2 sq(const Jet2& x) { return chain(x, x.v * x.v, 2.0 * x.v, 2.0); }
HD Jet2 log10(const Jet2& x) { const double i = 1.0 / x.v, c = 0.4342944819032518; return chain(x, ::log10(x.v), c * i, -c * i * i); }
HD Jet2 fabs(const Jet2& x) { return chain(x, ::fabs(x.v), sgn_of(x.v), 0.0); }
HD Jet2 sign(const Jet2& x) { return chain(x, sgn_of(x.v), 0.0, 0.0); }
HD Jet2 asin(const Jet2& x) { const double r = 1.0 / ::sqrt(1.0 - x.v * x.v); return chain(x, ::asin(x.v), r, x.v * r * r * r); }
HD Jet2 acos(const Jet2& x) { const double r = 1.0 / ::sqrt(1.0 - x.v * x.v); return chain(x, ::acos(x.v), -r, -x.v * r * r * r); }
HD Jet2 atan(const Jet2& x) { const double r = 1.0 / (1.0 + x.v * x.v); return chain(x, ::atan(x.v), r, -2.0 * x.v * r * r); }
HD Jet2 asinh(const Jet2& x) { const double r = 1.0 / ::sqrt(x.v * x.v + 1.0); return chain(x, ::asinh(x.v), r, -x.v * r * r * r); }
HD Jet2 acosh(const Jet2& x) { const double r = 1.0 / ::sqrt(x.v * x.v - 1.0); return chain(x, ::acosh(x.v), r, -x.v * r * r * r); }
HD Jet2 atanh(const Jet2& x) { const double r = 1.0 / (1.0 - x.v * x.v); return chain(x, ::atanh(x.v), r, 2.0 * x.v * r * r); }
// f = atan2(y, x):  f' = n / r with n = x y' - y x', r = x^2 + y^2;  f'' = (n' r - n r') / r^2, n' = x y'' - y x'', r' = 2 (x x' + y y')
HD Jet2 atan2(const Jet2& y, const Jet2& x) {
  const double r = x.v * x.v + y.v * y.v, ir = 1.0 / r, n = x.v * y.a - y.v * x.a;
  const double n1 = x.v * y.b - y.v * x.b, r1 = 2.0 * (x.v * x.a + y.v * y.a);
  return Jet2(::atan2(y.v, x.v), n * ir, (n1 * r - n * r1) * ir * ir);
}
HD Jet2 atan2(const Jet2& y, double x) { return atan2(y, Jet2(x)); }
HD Jet2 atan2(double y, const Jet2& x) { return atan2(Jet2(y), x); }

// value part of any of the scalar types (decisions inside generic code: pivoting, convergence tests)
HD double valof(double v) { return v; }
HD double valof(const FastD& v) { return v.v; }
HD double valof(const Jet2& v) { return v.v; }
template <int N> HD double valof(const Dual<N>& v) { return v.v; }

// plain doubles take part in the same generic code (explicit overloads: inside this namespace the AD
// overloads would otherwise hide ::sin etc. and a double would convert silently to Jet2)
HD double sin(double x) { return ::sin(x); }
HD double cos(double x) { return ::cos(x); }
HD double exp(double x) { return ::exp(x); }
HD double log(double x) { return ::log(x); }
HD double sqrt(double x) { return ::sqrt(x); }
HD double sq(double x) { return x * x; }
HD double log10(double x) { return ::log10(x); }
HD double fabs(double x) { return ::fabs(x); }
HD double sign(double x) { return sgn_of(x); }
HD double asin(double x) { return ::asin(x); }
HD double acos(double x) { return ::acos(x); }
HD double atan(double x) { return ::atan(x); }
HD double asinh(double x) { return ::asinh(x); }
HD double acosh(double x) { return ::acosh(x); }
HD double atanh(double x) { return ::atanh(x); }
HD double atan2(double y, double x) { return ::atan2(y, x); }
HD double value(double x) { return x; }
template <int N> HD double value(const Dual<N>& x) { return x.v; }
HD double value(const Jet2& x) { return x.v; }

}  // namespace hilo
