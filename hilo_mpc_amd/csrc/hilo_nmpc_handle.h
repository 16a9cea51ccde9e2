// The NMPC handle behind the C ABI (shared by the precompiled path, hilo_nmpc.hip, and the run-time compiled one,
// hilo_nmpc_user.hip).
#pragma once
#include "hilo_jit.h"
#include "hilo_nmpc_gen.h"

struct hilo_nmpc {
  using OcpConst = hilo::OcpConst; using GenVariant = hilo::GenVariant; using TrackBigVariant = hilo::TrackBigVariant;
  using TvVariant = hilo::TvVariant; using CollVariant = hilo::CollVariant;
  int device, model_id, nx, nu, np, N, n_v, n_g;
  OcpConst host;
  OcpConst* dev;
  unsigned base_free_mask;   // x_0 components that are variables by construction (theta_0, shared slack)
  double* par_buf;   // [par_batch][np + nu] device: model parameters | u_old
  int64_t par_batch;
  long long* prof;   // optional phase-cycle counters (hilo_nmpc_profile)
  double* v_guess;   // [n_v] device
  double* ext_pack;  // packed learned term of the model (GpExt) or NULL
  const GenVariant* gen;  // general variant (path following / stage constraints), NULL for the tracking policy
  int nu_out;        // width of the returned first input (model inputs, without the virtual path input)
  double* ws;        // iterate workspace of BIG variants [ws_batch][ws_bytes]
  int64_t ws_batch;
  const TrackBigVariant* big; // long-horizon variant of the tracking policy (iterate in the workspace) or NULL
  const TvVariant* tv;       // per-stage-data variant (trajectory references, time-varying parameters) or NULL
  const CollVariant* coll;   // collocation variant of the tracking policy or NULL
  double *vc, *lamc;         // the engine's compact [x | u] solution and defect multipliers (collocation output pass)
  int64_t vc_batch;
  int n_vc;                  // (N+1) nx + N nu
  int n_gc;                  // length of the engine's compact multiplier row of a run-time compiled collocation problem (0: N nxv)
  double* v_warm;    // [warm_batch][n_v] device: previous solution (mpc.py:725-726)
  int64_t warm_batch;
  int warm_valid;
  size_t lds_bytes;
  // run-time compiled problem (hilo_jit.hip): kernels of the loaded module, policy, engine dimensions
  hilo::JitKernels jit;
  int jit_policy;            // -1: precompiled
  int nxe, nue, nxv, ntail;  // engine state / input width, reference x width per stage, shared tail (slacks) in v
  int Nc;
  int tv_width;              // doubles per stage of the per-stage data table (0: none)
  size_t jit_ws_bytes;       // per-instance iterate workspace of a run-time compiled problem (0: iterate in LDS)
  int jit_coll_d;            // non-zero: an output pass follows the solve - the collocation degree of a run-time compiled problem, or
                             // 100 + order for algebraic states under an explicit Runge-Kutta transcription (hilo_nmpc_user.h::erk_dae_output)
  double* user_gp_pack[4];   // packed learned terms of a run-time compiled model (gp_pack_se) or NULL
  double* aux_g;             // caller's buffers for the constraint values / bound multipliers of the next solves, or NULL
  double* aux_lam_x;
  double* plant_out;         // caller's buffer for the fused plant step (hilo_nmpc_set_plant_out) or NULL
  double* gather;            // caller's gather table (hilo_nmpc_set_gather) or NULL
  int gather_stride;
  const double *var_lb, *var_ub;   // per-call lbx / ubx rows [B][n_v] of the next solves (hilo_nmpc_set_var_bounds) or NULL
};


namespace hilo {
// creation of a problem on the general run-time compiled policy (csrc/hilo_nmpc_user.h)
int nmpc_user_create(const hilo_nmpc_desc* d, int device, hilo_nmpc** out);
// packs desc.user_gp[] and writes the pointers into the loaded module's table; no-op without learned terms
int nmpc_bind_user_gps(hilo_nmpc* h, const hilo_nmpc_desc* d);
}
