// Internal interface between hilo_mhe.hip (C ABI, handle) and hilo_mhe_est.hip (parameter-estimating policy).
#pragma once
#include "hilo_ocp.h"

namespace hilo {

struct MheEstArgs {
  const OcpConst* dev;
  int64_t batch;
  const double* x0e;      // [B][mx+np]: values of the PINNED parameter slots (state slots unused)
  const double* par;      // [B][mx+np]: x_arrival | p_arrival
  const double* sd;       // [B][N+1][nu+ny]
  int64_t sd_stride;
  const double* v0e;      // engine-layout start rows
  int64_t v0e_stride;
  double *ve, *f_opt, *lame;
  int32_t *status, *iters;
  double* kkt;
  size_t lds_bytes;
  hipStream_t stream;
};
struct MheEstVariant {
  int model_id;
  size_t (*lds_bytes)(int N);
  int (*launch)(const MheEstArgs& a);
  void (*offsets)(int* o);   // O_WX, O_WP, O_WY, O_WW, O_SU
};
const MheEstVariant* mhe_est_find(int model_id);
// x0e[b] = [0 (mx) | p_b], par[b] = [x_arrival_b | p_b]
int mhe_est_pack(int64_t batch, int mx, int np, const double* p, int64_t p_stride, const double* xa, double* x0e, double* par,
                 hipStream_t s);
// has_w = 0: an estimator without state noise (rows [p | x | ...] / [xa])
int mhe_est_convert_in(int64_t batch, int N, int mx, int np, const double* v, int64_t v_stride, double* ve, hipStream_t s, int has_w = 1);
int mhe_est_convert_out(const OcpConst* pc, int64_t batch, int N, int mx, int np, const double* ve, const double* lame,
                        double* v, double* lam_g, double* x_opt, hipStream_t s);

}  // namespace hilo
