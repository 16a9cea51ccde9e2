// microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 (per SIMD) with NACC independent accumulator chains
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs, int iters) {
  double* d; (void)hipMalloc(&d, sizeof(double) * wgs * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NACC><<<wgs, 256>>>(d, 10, 1.0, 1.0);
  (void)hipEventRecord(e0);
  k<NACC><<<wgs, 256>>>(d, iters, 1.0, 1.0);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)wgs * 4 * iters * NACC * 2048.0;
  printf("NACC %d wgs %d: %.3f ms  %.1f TFLOP/s  cycles/mfma/SIMD @2.4GHz (1 wave/SIMD if wgs=256): %.1f\n", NACC, wgs, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / ((double)iters * NACC * (wgs / 256.0)));
  (void)hipFree(d);
}
int main() {
  run<1>(256, 20000); run<2>(256, 20000); run<4>(256, 10000); run<4>(512, 10000); run<1>(512, 20000); run<4>(2048, 4000);
  return 0;
}
