// Shared host-side plumbing for libhilo_hip.so (error reporting, launch helpers) and device tile staging.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/hilo_hip.h"
#else
// run-time compiled device code (hilo_jit.hip): only the status codes of the public header are needed
#define HILO_STATUS_SOLVED 1
#define HILO_STATUS_ACCEPTABLE 2
#define HILO_STATUS_INFEASIBLE 3
#define HILO_STATUS_RESTORATION_FAILED 4
#define HILO_STATUS_MAXITER 5
#define HILO_STATUS_OTHER (-1)
typedef long long int64_t;
typedef int int32_t;
#endif

namespace hilo {
#ifndef __HIPCC_RTC__

// hilo_gp.hip: device copy of a trained GP's posterior mean in the layout of hilo_models.h::GpExt; the GP must have a
// squared-exponential kernel over exactly two features and a constant (or zero) mean, HILO_ENOTSUP otherwise
int gp_pack_se2(const hilo_gp* gp, double** d_pack);
// the same for any number of active features, in the layout of hilo_models.h::gp_se_mean (run-time compiled models)
int gp_pack_se(const hilo_gp* gp, double** d_pack);

// thread-local last-error text (hilo_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...);

#define HILO_HIP_CHECK(expr)                                                                  \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      return ::hilo::fail(HILO_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),  \
                          __FILE__, __LINE__);                                                \
  } while (0)

#define HILO_REQUIRE(cond, ...)                                  \
  do {                                                           \
    if (!(cond)) return ::hilo::fail(HILO_EINVAL, __VA_ARGS__);  \
  } while (0)

#endif  // !__HIPCC_RTC__

// ------------------------------------------------------------------------------------------------
// Coalesced AoS <-> per-lane staging through LDS.
//
// The ABI keeps one instance's packed tile contiguous ([B][ROW] doubles, ROW = nx*(nx+1) ...), which is what a
// host caller holds; a thread-per-instance kernel reading it directly would touch 64 different cache lines per
// load instruction.  Instead the workgroup copies its TPB*ROW contiguous doubles with unit-stride lanes into
// LDS (row pitch ROW+1 when ROW is even -> odd pitch, conflict-free ds_read_b64 across a 32-lane group), and each
// lane then picks up its own row.
// ------------------------------------------------------------------------------------------------
template <int ROW> struct TilePitch { static constexpr int value = (ROW % 2 == 0) ? ROW + 1 : ROW; };

template <int ROW, int TPB>
__device__ __forceinline__ void tile_load(const double* __restrict__ g, int64_t first, int count,
                                          double* __restrict__ lds, double* __restrict__ reg) {
  constexpr int PITCH = TilePitch<ROW>::value;
  const double* src = g + first * ROW;
  const int total = count * ROW;
  for (int e = threadIdx.x; e < total; e += TPB) {
    const int i = e / ROW, c = e - i * ROW;
    lds[i * PITCH + c] = src[e];
  }
  __syncthreads();
  if ((int)threadIdx.x < count) {
#pragma unroll
    for (int c = 0; c < ROW; ++c) reg[c] = lds[threadIdx.x * PITCH + c];
  }
  __syncthreads();
}

template <int ROW, int TPB>
__device__ __forceinline__ void tile_store(double* __restrict__ g, int64_t first, int count,
                                           double* __restrict__ lds, const double* __restrict__ reg) {
  constexpr int PITCH = TilePitch<ROW>::value;
  if ((int)threadIdx.x < count) {
#pragma unroll
    for (int c = 0; c < ROW; ++c) lds[threadIdx.x * PITCH + c] = reg[c];
  }
  __syncthreads();
  double* dst = g + first * ROW;
  const int total = count * ROW;
  for (int e = threadIdx.x; e < total; e += TPB) {
    const int i = e / ROW, c = e - i * ROW;
    dst[e] = lds[i * PITCH + c];
  }
  __syncthreads();
}

// per-instance small vector with optional sharing (stride 0)
template <int LEN>
__device__ __forceinline__ void vec_load(const double* __restrict__ g, int64_t inst, int64_t stride,
                                         double* __restrict__ reg) {
  const double* src = g + inst * stride;
#pragma unroll
  for (int c = 0; c < LEN; ++c) reg[c] = src[c];
}

// bounds of variable i of instance b: the per-instance (or, with pinned values, shared: bs = 0) rows lbx / ubx, except that the first
// `npin` variables of hilo_qp_solve_pinned are fixed at xpin[b][i] (the measured state x_0, mpc.py:2361-2362, without the host
// writing it into two bound rows first)
__device__ __forceinline__ void qp_bounds(const double* __restrict__ lbx, const double* __restrict__ ubx, int64_t bs,
                                          const double* __restrict__ xpin, int npin, int64_t ps, int64_t b, int i, double& lo,
                                          double& up) {
  if (xpin != nullptr && i < npin) {
    lo = up = xpin[b * ps + i];
  } else {
    lo = lbx[b * bs + i];
    up = ubx[b * bs + i];
  }
}

}  // namespace hilo
