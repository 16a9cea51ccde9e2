// Batched Kalman / extended / unscented Kalman filter: C ABI and launches (device code: csrc/hilo_kf_kernel.h).
//
// Reference semantics: hilo_mpc/modules/estimator/kf.py
//   predict  :71-133   (UKF :505-554)      update :135-186 (UKF :556-604)      step = update(predict) :258-265
#include "hilo_jit.h"
#include "hilo_kf_kernel.h"

namespace hilo {

template <class M, bool UKF, int MODE>
int launch(const KfParams& kp, int64_t batch, const double* in, const double* y, const double* up,
           int64_t up_stride, const double* Q, int64_t q_stride, const double* R, int64_t r_stride, double* out,
           double* y_pred, hipStream_t s) {
  if (batch == 0) return HILO_OK;
  int64_t ipw = (batch + 1023) / 1024;   // 256 compute units x 4 SIMDs
  ipw = ipw < 16 ? 16 : (ipw > KF_TPB ? KF_TPB : ipw);
  const unsigned grid = (unsigned)((batch + ipw - 1) / ipw);
  hipLaunchKernelGGL((kf_kernel<M, UKF, MODE>), dim3(grid), dim3(KF_TPB), 0, s, kp, batch, in, y, up, up_stride, Q,
                     q_stride, R, r_stride, out, y_pred, (int)ipw);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

template <class M>
int dispatch_kind(int mode, const KfParams& kp, int64_t batch, const double* in, const double* y,
                  const double* up, int64_t us, const double* Q, int64_t qs, const double* R, int64_t rs,
                  double* out, double* yp, hipStream_t s) {
  const bool ukf = kp.kind == HILO_KF_UKF;
  if (ukf) {
    if (mode == 0) return launch<M, true, 0>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
    if (mode == 1) return launch<M, true, 1>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
    return launch<M, true, 2>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
  }
  if (mode == 0) return launch<M, false, 0>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
  if (mode == 1) return launch<M, false, 1>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
  return launch<M, false, 2>(kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
}

// One instance on a team of lanes (csrc/hilo_kf_kernel.h::kf_team_body) while one instance per lane would leave lanes idle: up to
// two waves per SIMD of teams (256 compute units x 4 SIMDs x 64 lanes x 2).  The EKF of a continuous model integrates the
// augmented ODE on one lane.  HILO_KF_TEAM=0 (developer / test knob) keeps every batch on the one-lane kernels.
static int team_lanes(int kind, int nx) {
  int t = 1;
  const int need = kind == HILO_KF_UKF ? 2 * nx + 1 : nx;
  while (t < need) t *= 2;
  return t;
}
static bool use_team(const KfParams& kp, int nx, bool discrete, int64_t batch) {
  static const int knob = [] { const char* e = getenv("HILO_KF_TEAM"); return e ? atoi(e) : 1; }();
  const int t = team_lanes(kp.kind, nx);
  if (!knob || t > KF_TPB) return false;
  if (kp.kind != HILO_KF_UKF && kp.continuous && !discrete) return false;
  return batch * t <= 2 * 1024 * (int64_t)KF_TPB;
}

template <class M>
int launch_multi(const KfParams& kp, int64_t batch, int steps, const double* in, const double* y, const double* up, int64_t us,
                 int64_t ustep, const double* Q, int64_t qs, const double* R, int64_t rs, double* out, int64_t ostep, double* yp,
                 hipStream_t s) {
  if (use_team(kp, M::NX, M::DISCRETE, batch)) {
    if (kp.kind == HILO_KF_UKF) {
      const unsigned grid = (unsigned)((batch + KfTeam<M, true>::TEAMS - 1) / KfTeam<M, true>::TEAMS);
      hipLaunchKernelGGL((kf_team_kernel<M, true>), dim3(grid), dim3(KF_TPB), 0, s, kp, batch, steps, in, y, up, us, ustep, Q, qs, R, rs,
                         out, ostep, yp);
    } else {
      const unsigned grid = (unsigned)((batch + KfTeam<M, false>::TEAMS - 1) / KfTeam<M, false>::TEAMS);
      hipLaunchKernelGGL((kf_team_kernel<M, false>), dim3(grid), dim3(KF_TPB), 0, s, kp, batch, steps, in, y, up, us, ustep, Q, qs, R, rs,
                         out, ostep, yp);
    }
    HILO_HIP_CHECK(hipGetLastError());
    return HILO_OK;
  }
  int64_t ipw = (batch + 1023) / 1024;       // same instance split as the single step
  ipw = ipw < 16 ? 16 : (ipw > KF_TPB ? KF_TPB : ipw);
  const unsigned grid = (unsigned)((batch + ipw - 1) / ipw);
  // the common recipe - `discretize('rk4')`, one sub-step, Q and R shared by the batch - on the variant that fits two waves per SIMD
  // (csrc/hilo_models.h::rk4_classic without the run-time order dispatch; the same arithmetic, fewer registers)
  const bool rk4 = kp.n_sub == 1 && (kp.kind == HILO_KF_UKF ? (kp.erk_order == 4 || kp.continuous) : (kp.erk_order == 4 && !kp.continuous));
  static const bool lean_knob = [] { const char* e = getenv("HILO_KF_LEAN"); return e ? atoi(e) != 0 : true; }();
  if constexpr (!M::DISCRETE) if (rk4 && qs == 0 && rs == 0 && lean_knob) {
    if (kp.kind == HILO_KF_UKF && batch >= 2 * 1024 * (int64_t)KF_TPB)      // two waves per SIMD to fill: the 252-register variant
      hipLaunchKernelGGL((kf_multi_kernel<M, true, 2>), dim3(grid), dim3(KF_TPB), 0, s, kp, batch, steps, in, y, up, us, ustep, Q, qs,
                         R, rs, out, ostep, yp, (int)ipw);
    else if (kp.kind == HILO_KF_UKF)
      hipLaunchKernelGGL((kf_multi_kernel<M, true, 1>), dim3(grid), dim3(KF_TPB), 0, s, kp, batch, steps, in, y, up, us, ustep, Q, qs,
                         R, rs, out, ostep, yp, (int)ipw);
    else
      hipLaunchKernelGGL((ekf_multi_lean_kernel<M>), dim3(grid), dim3(KF_TPB), 0, s, kp, batch, steps, in, y, up, us, ustep, Q, qs,
                         R, rs, out, ostep, yp, (int)ipw);
    HILO_HIP_CHECK(hipGetLastError());
    return HILO_OK;
  }
  if (kp.kind == HILO_KF_UKF)
    hipLaunchKernelGGL((kf_multi_kernel<M, true>), dim3(grid), dim3(KF_TPB), 0, s, kp, batch, steps, in, y, up, us, ustep, Q, qs, R,
                       rs, out, ostep, yp, (int)ipw);
  else
    hipLaunchKernelGGL((kf_multi_kernel<M, false>), dim3(grid), dim3(KF_TPB), 0, s, kp, batch, steps, in, y, up, us, ustep, Q, qs, R,
                       rs, out, ostep, yp, (int)ipw);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

}  // namespace hilo

using namespace hilo;

struct hilo_kf {
  hilo_kf_desc desc;
  int device;
  int nx, nu, np, ny;
  bool discrete;
  KfParams kp;
  hilo::JitKfKernels jit;   // model given as source (HILO_MODEL_USER): kernels compiled at create
  double* user_gp_pack[4] = {nullptr, nullptr, nullptr, nullptr};   // packed learned terms of that model (gp_pack_se)
};

#define HILO_KF_MODELS(X)                 \
  X(HILO_MODEL_TOY1D, Toy1D)              \
  X(HILO_MODEL_BIOREACTOR3, Bioreactor3)  \
  X(HILO_MODEL_CHEMOSTAT4, Chemostat4)    \
  X(HILO_MODEL_PENDULUM4, Pendulum4)      \
  X(HILO_MODEL_LINEAR2, Linear2)

static int kf_model_dims(const hilo_kf_desc* d, int* nx, int* nu, int* np, int* ny, int* discrete) {
  switch (d->model_id) {
#define X(ID, T) case ID: *nx = T::NX; *nu = T::NU; *np = T::NP; *ny = T::NY; *discrete = T::DISCRETE; return HILO_OK;
    HILO_KF_MODELS(X)
#undef X
    case HILO_MODEL_LTI:
      if (d->lti_nx == 2 && d->lti_nu == 1 && d->lti_ny == 1) { *nx = 2; *nu = 1; *np = 4 + 2 + 2; *ny = 1; *discrete = 1; return HILO_OK; }
      if (d->lti_nx == 2 && d->lti_nu == 1 && d->lti_ny == 2) { *nx = 2; *nu = 1; *np = 4 + 2 + 4; *ny = 2; *discrete = 1; return HILO_OK; }
      if (d->lti_nx == 4 && d->lti_nu == 2 && d->lti_ny == 2) { *nx = 4; *nu = 2; *np = 16 + 8 + 8; *ny = 2; *discrete = 1; return HILO_OK; }
      return fail(HILO_ENOTSUP, "LTI filter built for (nx,nu,ny) in {(2,1,1),(2,1,2),(4,2,2)}, got (%d,%d,%d)",
                  d->lti_nx, d->lti_nu, d->lti_ny);
    default:
      return fail(HILO_EINVAL, "unknown model id %d", d->model_id);
  }
}

extern "C" int hilo_model_dims(int model_id, int* nx, int* nu, int* np, int* ny, int* discrete) {
  hilo_kf_desc d = {};
  d.model_id = model_id;
  if (model_id == HILO_MODEL_LTI) return fail(HILO_EINVAL, "LTI dimensions are caller-defined");
  if (model_id == HILO_MODEL_CHEMOSTAT4_GP) model_id = d.model_id = HILO_MODEL_CHEMOSTAT4;  // same signature
  if (model_id == HILO_MODEL_CSTR3) {  // controller-only model (no filter instantiation)
    if (nx) *nx = Cstr3::NX;
    if (nu) *nu = Cstr3::NU;
    if (np) *np = Cstr3::NP;
    if (ny) *ny = Cstr3::NY;
    if (discrete) *discrete = Cstr3::DISCRETE;
    return HILO_OK;
  }
  if (model_id == HILO_MODEL_ROBOT6) {  // controller-only model (no filter instantiation)
    if (nx) *nx = Robot6::NX;
    if (nu) *nu = Robot6::NU;
    if (np) *np = Robot6::NP;
    if (ny) *ny = Robot6::NY;
    if (discrete) *discrete = Robot6::DISCRETE;
    return HILO_OK;
  }
  int a, b, c, e, f;
  int rc = kf_model_dims(&d, &a, &b, &c, &e, &f);
  if (rc) return rc;
  if (nx) *nx = a;
  if (nu) *nu = b;
  if (np) *np = c;
  if (ny) *ny = e;
  if (discrete) *discrete = f;
  return HILO_OK;
}

extern "C" int hilo_kf_create(const hilo_kf_desc* desc, int device, hilo_kf** out) {
  HILO_REQUIRE(desc && out, "hilo_kf_create: NULL argument");
  HILO_REQUIRE(desc->kind >= HILO_KF_KF && desc->kind <= HILO_KF_UKF, "hilo_kf_create: unknown kind %d", desc->kind);
  HILO_REQUIRE(desc->dt > 0.0, "hilo_kf_create: dt must be positive");
  int nx, nu, np, ny, disc;
  int rc;
  JitKfKernels jit;
  if (desc->model_id == 100 /* HILO_MODEL_USER */) {
    HILO_REQUIRE(desc->user_source && desc->user_source[0], "hilo_kf_create: HILO_MODEL_USER needs desc.user_source");
    HILO_REQUIRE(desc->n_user_gp >= 0 && desc->n_user_gp <= 4, "hilo_kf_create: at most 4 learned terms (got %d)", desc->n_user_gp);
    rc = jit_kf_kernels(desc->user_source, device, &jit, false, desc->n_user_gp > 0);
    if (rc) return rc;
    if (getenv("HILO_JIT_COMPILE_ONLY")) return HILO_COMPILED_ONLY;   // cache warmed, no handle
    nx = jit.dims[0]; nu = jit.dims[1]; np = jit.dims[2]; ny = jit.dims[3]; disc = jit.dims[4];
    HILO_REQUIRE(ny >= 1, "hilo_kf_create: the model has no measurement equations");
  } else {
    rc = kf_model_dims(desc, &nx, &nu, &np, &ny, &disc);
    if (rc) return rc;
  }
  HILO_REQUIRE(disc || (desc->erk_order >= 1 && desc->erk_order <= 4) || desc->continuous,
               "hilo_kf_create: erk_order must be 1..4 (modeling.py:1239-1250), got %d", desc->erk_order);
  if (desc->kind == HILO_KF_UKF) {
    // kf.py:475-484
    HILO_REQUIRE(desc->alpha > 0.0 && desc->alpha <= 1.0,
                 "The parameter alpha needs to lie in the interval (0, 1]. Supplied alpha is %g.", desc->alpha);
    HILO_REQUIRE(desc->kappa >= 0.0,
                 "The parameter kappa needs to be greater or equal to 0. Supplied kappa is %g.", desc->kappa);
  }
  int ndev = 0;
  HILO_HIP_CHECK(hipGetDeviceCount(&ndev));
  HILO_REQUIRE(device >= 0 && device < ndev, "hilo_kf_create: device %d out of range (%d visible)", device, ndev);
  hilo_kf* kf = new hilo_kf();
  kf->desc = *desc;
  kf->desc.user_source = nullptr;   // not kept: the kernels are
  for (auto& g : kf->desc.user_gp) g = nullptr;   // nor the learned terms' handles: their posteriors are packed below
  kf->jit = jit;
  kf->device = device;
  kf->nx = nx; kf->nu = nu; kf->np = np; kf->ny = ny;
  kf->discrete = disc != 0;
  KfParams& kp = kf->kp;
  kp.kind = desc->kind;
  kp.continuous = desc->continuous;
  kp.erk_order = desc->erk_order >= 1 ? desc->erk_order : 4;
  kp.n_sub = desc->n_sub >= 1 ? desc->n_sub : 1;
  kp.dt = desc->dt;
  // kf.py:493-500
  const double lam = desc->alpha * desc->alpha * (nx + desc->kappa) - nx;
  kp.gamma = ::sqrt(nx + lam);
  kp.wm0 = lam / (nx + lam);
  kp.wc0 = lam / (nx + lam) + 1 - desc->alpha * desc->alpha + desc->beta;
  kp.wi = 1 / (2 * (nx + lam));
  if (desc->model_id == 100 && desc->n_user_gp > 0) {   // learned terms: packed posteriors into the module's table
    const double* table[4] = {nullptr, nullptr, nullptr, nullptr};
    rc = kf->jit.gp_table ? HILO_OK : fail(HILO_EHIP, "hilo_kf_create: the compiled module exports no learned-term table");
    for (int k = 0; !rc && k < desc->n_user_gp; ++k) {
      if (!desc->user_gp[k]) rc = fail(HILO_EINVAL, "hilo_kf_create: user_gp[%d] is NULL", k);
      else rc = gp_pack_se(desc->user_gp[k], &kf->user_gp_pack[k]);
      table[k] = kf->user_gp_pack[k];
    }
    if (!rc && hipMemcpy((void*)kf->jit.gp_table, table, sizeof(table), hipMemcpyHostToDevice) != hipSuccess)
      rc = fail(HILO_EHIP, "hilo_kf_create: writing the learned-term table failed");
    if (rc) { hilo_kf_destroy(kf); return rc; }
  }
  *out = kf;
  return HILO_OK;
}

extern "C" void hilo_kf_destroy(hilo_kf* kf) {
  if (!kf) return;
  for (double* g : kf->user_gp_pack)
    if (g) (void)hipFree(g);
  jit_kf_unload(&kf->jit);   // (a shared module stays loaded: `owned` is only set for a private instance)
  delete kf;
}

#ifdef HILO_KF_PROF   // developer builds (tools/dbg/kf_phase.py): the cycle stamps of the team kernels
extern "C" int hilo_debug_kf_prof(unsigned long long* out) {
  HILO_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(hilo::hilo_kf_prof), sizeof(unsigned long long) * 16));
  return HILO_OK;
}
#endif

// compile the filter kernels of a model source into the cache without loading them (no GPU needed)
extern "C" int hilo_jit_precompile_kf(const char* user_source) {
  HILO_REQUIRE(user_source && user_source[0], "hilo_jit_precompile_kf: empty source");
  JitKfKernels k;
  return jit_kf_kernels(user_source, 0, &k, true);
}

extern "C" int hilo_kf_dims(const hilo_kf* kf, int* nx, int* nu, int* np, int* ny, int* pred_width) {
  HILO_REQUIRE(kf, "hilo_kf_dims: NULL handle");
  if (nx) *nx = kf->nx;
  if (nu) *nu = kf->nu;
  if (np) *np = kf->np;
  if (ny) *ny = kf->ny;
  if (pred_width) *pred_width = kf->kp.kind == HILO_KF_UKF ? 1 + kf->nx + 2 * kf->nx + 1 : kf->nx + 1;
  return HILO_OK;
}

extern "C" int hilo_kf_steps(hilo_kf* kf, int64_t batch, int steps, const double* xP, const double* y, const double* up,
                             int64_t up_stride, int64_t up_step_stride, const double* Q, int64_t q_stride, const double* R,
                             int64_t r_stride, double* xP_out, int keep_all, double* y_pred, void* stream);

static int kf_run(hilo_kf* kf, int mode, int64_t batch, const double* in, const double* y, const double* up,
                  int64_t us, const double* Q, int64_t qs, const double* R, int64_t rs, double* out, double* yp,
                  void* stream) {
  HILO_REQUIRE(kf, "hilo_kf: NULL handle");
  HILO_REQUIRE(batch >= 0, "hilo_kf: negative batch");
  if (batch == 0) return HILO_OK;
  HILO_REQUIRE(in && out, "hilo_kf: NULL tile pointer");
  HILO_REQUIRE(kf->nu + kf->np == 0 || up, "hilo_kf: the model has %d inputs/parameters but `up` is NULL", kf->nu + kf->np);
  HILO_REQUIRE(us == 0 || us >= kf->nu + kf->np, "hilo_kf: up_stride %lld < nu+np", (long long)us);
  if (mode != 1) HILO_REQUIRE(Q && (qs == 0 || qs >= kf->nx * kf->nx), "hilo_kf: bad Q / q_stride");
  if (mode != 0) HILO_REQUIRE(R && y && yp && (rs == 0 || rs >= kf->ny * kf->ny), "hilo_kf: bad R / y / y_pred");
  HILO_HIP_CHECK(hipSetDevice(kf->device));
  hipStream_t s = (hipStream_t)stream;
  const KfParams& kp = kf->kp;
  if (mode == 2 && use_team(kp, kf->nx, kf->discrete, batch))   // a fused step of a small batch: one step of the team kernel
    return hilo_kf_steps(kf, batch, 1, in, y, up, us, 0, Q, qs, R, rs, out, 0, yp, stream);
  if (kf->desc.model_id == 100 /* HILO_MODEL_USER */) {
    hipFunction_t f = kf->jit.f[kp.kind == HILO_KF_UKF ? 1 : 0][mode];
    HILO_REQUIRE(f, "hilo_kf: the run-time compiled filter kernels are not loaded");
    int64_t ipw = (batch + 1023) / 1024;
    ipw = ipw < 16 ? 16 : (ipw > KF_TPB ? KF_TPB : ipw);
    int ipw_i = (int)ipw;
    const unsigned grid = (unsigned)((batch + ipw - 1) / ipw);
    KfParams kpv = kp;
    void* args[] = {&kpv, &batch, &in, &y, &up, &us, &Q, &qs, &R, &rs, &out, &yp, &ipw_i};
    HILO_HIP_CHECK(hipModuleLaunchKernel(f, grid, 1, 1, KF_TPB, 1, 1, 0, s, args, nullptr));
    return HILO_OK;
  }
  switch (kf->desc.model_id) {
#define X(ID, T) case ID: return dispatch_kind<T>(mode, kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
    HILO_KF_MODELS(X)
#undef X
    case HILO_MODEL_LTI:
      if (kf->nx == 2 && kf->ny == 1) return dispatch_kind<Lti<2, 1, 1>>(mode, kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
      if (kf->nx == 2 && kf->ny == 2) return dispatch_kind<Lti<2, 1, 2>>(mode, kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
      return dispatch_kind<Lti<4, 2, 2>>(mode, kp, batch, in, y, up, us, Q, qs, R, rs, out, yp, s);
  }
  return fail(HILO_EINVAL, "unknown model id %d", kf->desc.model_id);
}

extern "C" int hilo_kf_predict(hilo_kf* kf, int64_t batch, const double* xP, const double* up, int64_t up_stride,
                               const double* Q, int64_t q_stride, double* pred, void* stream) {
  return kf_run(kf, 0, batch, xP, nullptr, up, up_stride, Q, q_stride, nullptr, 0, pred, nullptr, stream);
}
extern "C" int hilo_kf_update(hilo_kf* kf, int64_t batch, const double* pred, const double* y, const double* up,
                              int64_t up_stride, const double* R, int64_t r_stride, double* xP_out, double* y_pred,
                              void* stream) {
  return kf_run(kf, 1, batch, pred, y, up, up_stride, nullptr, 0, R, r_stride, xP_out, y_pred, stream);
}
extern "C" int hilo_kf_step(hilo_kf* kf, int64_t batch, const double* xP, const double* y, const double* up,
                            int64_t up_stride, const double* Q, int64_t q_stride, const double* R, int64_t r_stride,
                            double* xP_out, double* y_pred, void* stream) {
  return kf_run(kf, 2, batch, xP, y, up, up_stride, Q, q_stride, R, r_stride, xP_out, y_pred, stream);
}

// `steps` fused estimate() steps in one launch: the reference's `self._function.mapaccum(steps)` (kf.py:296-306)
static int kf_steps_impl(hilo_kf* kf, int64_t batch, int steps, const double* xP, const double* y, const double* up,
                         int64_t up_stride, int64_t up_step_stride, const double* pp, int64_t pp_stride, const double* Q,
                         int64_t q_stride, const double* R, int64_t r_stride, double* xP_out, int keep_all, double* y_pred,
                         void* stream);

extern "C" int hilo_kf_steps(hilo_kf* kf, int64_t batch, int steps, const double* xP, const double* y, const double* up,
                             int64_t up_stride, int64_t up_step_stride, const double* Q, int64_t q_stride, const double* R,
                             int64_t r_stride, double* xP_out, int keep_all, double* y_pred, void* stream) {
  HILO_REQUIRE(kf, "hilo_kf_steps: NULL handle");
  HILO_REQUIRE(up_stride == 0 || up_stride >= kf->nu + kf->np, "hilo_kf_steps: up_stride %lld < nu+np", (long long)up_stride);
  return kf_steps_impl(kf, batch, steps, xP, y, up, up_stride, up_step_stride, nullptr, 0, Q, q_stride, R, r_stride, xP_out, keep_all,
                       y_pred, stream);
}

// The same with the inputs and the parameters in their OWN arrays - what the reference hands its function: u and p are separate
// arguments there (kf.py:130), and packing [u; p] rows on the device first is a launch of its own per call.
extern "C" int hilo_kf_steps_split(hilo_kf* kf, int64_t batch, int steps, const double* xP, const double* y, const double* u,
                                   int64_t u_stride, int64_t u_step_stride, const double* p, int64_t p_stride, const double* Q,
                                   int64_t q_stride, const double* R, int64_t r_stride, double* xP_out, int keep_all, double* y_pred,
                                   void* stream) {
  HILO_REQUIRE(kf, "hilo_kf_steps_split: NULL handle");
  HILO_REQUIRE(kf->nu == 0 || u, "hilo_kf_steps_split: the model has %d inputs but `u` is NULL", kf->nu);
  HILO_REQUIRE(kf->np == 0 || p, "hilo_kf_steps_split: the model has %d parameters but `p` is NULL", kf->np);
  HILO_REQUIRE(u_stride == 0 || u_stride >= kf->nu, "hilo_kf_steps_split: u_stride %lld < nu", (long long)u_stride);
  HILO_REQUIRE(p_stride == 0 || p_stride >= kf->np, "hilo_kf_steps_split: p_stride %lld < np", (long long)p_stride);
  if (kf->np == 0)      // one of the two is empty: the other array IS the packed rows
    return kf_steps_impl(kf, batch, steps, xP, y, u, u_stride, u_step_stride, nullptr, 0, Q, q_stride, R, r_stride, xP_out, keep_all,
                         y_pred, stream);
  if (kf->nu == 0)
    return kf_steps_impl(kf, batch, steps, xP, y, p, p_stride, 0, nullptr, 0, Q, q_stride, R, r_stride, xP_out, keep_all, y_pred, stream);
  return kf_steps_impl(kf, batch, steps, xP, y, u, u_stride, u_step_stride, p, p_stride, Q, q_stride, R, r_stride, xP_out, keep_all,
                       y_pred, stream);
}

static int kf_steps_impl(hilo_kf* kf, int64_t batch, int steps, const double* xP, const double* y, const double* up,
                         int64_t up_stride, int64_t up_step_stride, const double* pp, int64_t pp_stride, const double* Q,
                         int64_t q_stride, const double* R, int64_t r_stride, double* xP_out, int keep_all, double* y_pred,
                         void* stream) {
  HILO_REQUIRE(kf, "hilo_kf_steps: NULL handle");
  HILO_REQUIRE(batch >= 0 && steps >= 1, "hilo_kf_steps: need batch >= 0 and steps >= 1");
  if (batch == 0) return HILO_OK;
  HILO_REQUIRE(xP && xP_out && y && y_pred && Q && R, "hilo_kf_steps: NULL argument");
  HILO_REQUIRE(kf->nu + kf->np == 0 || up, "hilo_kf_steps: the model has %d inputs/parameters but `up` is NULL", kf->nu + kf->np);
  HILO_REQUIRE(q_stride == 0 || q_stride >= kf->nx * kf->nx, "hilo_kf_steps: bad q_stride");
  HILO_REQUIRE(r_stride == 0 || r_stride >= kf->ny * kf->ny, "hilo_kf_steps: bad r_stride");
  HILO_HIP_CHECK(hipSetDevice(kf->device));
  hipStream_t s = (hipStream_t)stream;
  KfParams kp = kf->kp;
  kp.pp = pp;
  kp.pp_stride = (long long)pp_stride;
  static const double zero = 0.0;
  if (!up) up = &zero;
  const int64_t ostep = keep_all ? batch * (int64_t)kf->nx * (kf->nx + 1) : 0;
  if (kf->desc.model_id == 100 /* HILO_MODEL_USER */) {
    if (use_team(kp, kf->nx, kf->discrete, batch)) {
      hipFunction_t f = kf->jit.team[kp.kind == HILO_KF_UKF ? 1 : 0];
      HILO_REQUIRE(f, "hilo_kf_steps: the run-time compiled filter kernels are not loaded");
      const int teams = KF_TPB / team_lanes(kp.kind, kf->nx);
      const unsigned grid = (unsigned)((batch + teams - 1) / teams);
      KfParams kpv = kp;
      int64_t ostep_v = ostep;
      void* args[] = {&kpv, &batch, &steps, &xP, &y, &up, &up_stride, &up_step_stride, &Q, &q_stride, &R, &r_stride, &xP_out, &ostep_v,
                      &y_pred};
      HILO_HIP_CHECK(hipModuleLaunchKernel(f, grid, 1, 1, KF_TPB, 1, 1, 0, s, args, nullptr));
      return HILO_OK;
    }
    hipFunction_t f = kf->jit.multi[kp.kind == HILO_KF_UKF ? 1 : 0];
    HILO_REQUIRE(f, "hilo_kf_steps: the run-time compiled filter kernels are not loaded");
    int64_t ipw = (batch + 1023) / 1024;
    ipw = ipw < 16 ? 16 : (ipw > KF_TPB ? KF_TPB : ipw);
    int ipw_i = (int)ipw;
    const unsigned grid = (unsigned)((batch + ipw - 1) / ipw);
    KfParams kpv = kp;
    int64_t ostep_v = ostep;
    void* args[] = {&kpv, &batch, &steps, &xP, &y, &up, &up_stride, &up_step_stride, &Q, &q_stride, &R, &r_stride, &xP_out, &ostep_v,
                    &y_pred, &ipw_i};
    HILO_HIP_CHECK(hipModuleLaunchKernel(f, grid, 1, 1, KF_TPB, 1, 1, 0, s, args, nullptr));
    return HILO_OK;
  }
  switch (kf->desc.model_id) {
#define X(ID, T) case ID: return launch_multi<T>(kp, batch, steps, xP, y, up, up_stride, up_step_stride, Q, q_stride, R, r_stride, xP_out, ostep, y_pred, s);
    HILO_KF_MODELS(X)
#undef X
    case HILO_MODEL_LTI:
      if (kf->nx == 2 && kf->ny == 1) return launch_multi<Lti<2, 1, 1>>(kp, batch, steps, xP, y, up, up_stride, up_step_stride, Q, q_stride, R, r_stride, xP_out, ostep, y_pred, s);
      if (kf->nx == 2 && kf->ny == 2) return launch_multi<Lti<2, 1, 2>>(kp, batch, steps, xP, y, up, up_stride, up_step_stride, Q, q_stride, R, r_stride, xP_out, ostep, y_pred, s);
      return launch_multi<Lti<4, 2, 2>>(kp, batch, steps, xP, y, up, up_stride, up_step_stride, Q, q_stride, R, r_stride, xP_out, ostep, y_pred, s);
  }
  return fail(HILO_EINVAL, "unknown model id %d", kf->desc.model_id);
}

// ---- particle filter (pf.py): the propagate / measure / weigh function of the reference's setup(), and the resampling gather ----
namespace hilo {
template <class M>
static int pf_launch(const KfParams& kp, int64_t batch, int n, const double* X, const double* y, const double* up, int64_t us,
                     const double* w, const double* v, const double* R, int64_t rs, double* Xp, double* Y, double* q, hipStream_t s) {
  hipLaunchKernelGGL((pf_kernel<M>), dim3((unsigned)batch), dim3(PF_TPB), 0, s, kp, n, X, y, up, us, w, v, R, rs, Xp, Y, q);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}
}  // namespace hilo

extern "C" int hilo_pf_function(hilo_kf* kf, int64_t batch, int n_samples, const double* X, const double* y, const double* up,
                                int64_t up_stride, const double* w, const double* v, const double* R, int64_t r_stride,
                                double* X_prop, double* Y, double* q, void* stream) {
  HILO_REQUIRE(kf, "hilo_pf_function: NULL handle");
  HILO_REQUIRE(batch >= 0 && n_samples >= 1, "hilo_pf_function: need batch >= 0 and n_samples >= 1");
  if (batch == 0) return HILO_OK;
  HILO_REQUIRE(X && y && w && v && R && X_prop && Y && q, "hilo_pf_function: NULL argument");
  HILO_REQUIRE(kf->nu + kf->np == 0 || up, "hilo_pf_function: the model has %d inputs/parameters but `up` is NULL", kf->nu + kf->np);
  HILO_REQUIRE(up_stride == 0 || up_stride >= kf->nu + kf->np, "hilo_pf_function: up_stride %lld < nu+np", (long long)up_stride);
  const int nye = kf->ny > 0 ? kf->ny : kf->nx;
  HILO_REQUIRE(r_stride == 0 || r_stride >= nye * nye, "hilo_pf_function: bad r_stride");
  HILO_HIP_CHECK(hipSetDevice(kf->device));
  hipStream_t s = (hipStream_t)stream;
  const KfParams& kp = kf->kp;
  static const double zero = 0.0;
  if (!up) up = &zero;   // never read (nu + np == 0); keeps the pointer arithmetic of the kernel defined
  if (kf->desc.model_id == 100 /* HILO_MODEL_USER */) {
    HILO_REQUIRE(kf->jit.pf, "hilo_pf_function: the run-time compiled particle kernel is not loaded");
    KfParams kpv = kp;
    void* args[] = {&kpv, &n_samples, &X, &y, &up, &up_stride, &w, &v, &R, &r_stride, &X_prop, &Y, &q};
    HILO_HIP_CHECK(hipModuleLaunchKernel(kf->jit.pf, (unsigned)batch, 1, 1, PF_TPB, 1, 1, 0, s, args, nullptr));
    return HILO_OK;
  }
  switch (kf->desc.model_id) {
#define X_(ID, T) case ID: return pf_launch<T>(kp, batch, n_samples, X, y, up, up_stride, w, v, R, r_stride, X_prop, Y, q, s);
    HILO_KF_MODELS(X_)
#undef X_
    case HILO_MODEL_LTI:
      if (kf->nx == 2 && kf->ny == 1) return pf_launch<Lti<2, 1, 1>>(kp, batch, n_samples, X, y, up, up_stride, w, v, R, r_stride, X_prop, Y, q, s);
      if (kf->nx == 2 && kf->ny == 2) return pf_launch<Lti<2, 1, 2>>(kp, batch, n_samples, X, y, up, up_stride, w, v, R, r_stride, X_prop, Y, q, s);
      return pf_launch<Lti<4, 2, 2>>(kp, batch, n_samples, X, y, up, up_stride, w, v, R, r_stride, X_prop, Y, q, s);
  }
  return fail(HILO_EINVAL, "unknown model id %d", kf->desc.model_id);
}

extern "C" int hilo_pf_resample(hilo_kf* kf, int64_t batch, int n_samples, const double* X_prop, const double* Y, const double* q,
                                const double* uniforms, double* X, double* Y_out, int32_t* index, void* stream) {
  HILO_REQUIRE(kf, "hilo_pf_resample: NULL handle");
  HILO_REQUIRE(batch >= 0 && n_samples >= 1 && n_samples <= 8192, "hilo_pf_resample: need batch >= 0 and 1 <= n_samples <= 8192");
  if (batch == 0) return HILO_OK;
  HILO_REQUIRE(X_prop && Y && q && uniforms && X && Y_out && index, "hilo_pf_resample: NULL argument");
  HILO_HIP_CHECK(hipSetDevice(kf->device));
  const int nye = kf->ny > 0 ? kf->ny : kf->nx;
  const size_t lds = sizeof(double) * ((size_t)n_samples + PF_TPB);
  if (lds > 64 * 1024)
    HILO_HIP_CHECK(hipFuncSetAttribute((const void*)pf_resample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(pf_resample_kernel, dim3((unsigned)batch), dim3(PF_TPB), lds, (hipStream_t)stream, n_samples, kf->nx, nye, X_prop,
                     Y, q, uniforms, X, Y_out, (int*)index);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

extern "C" int hilo_pf_stats(hilo_kf* kf, int64_t batch, int n_samples, double* X, const double* Y, const double* add,
                             double* x_mean, double* y_mean, double* P, double* x_min, double* x_max, void* stream) {
  HILO_REQUIRE(kf, "hilo_pf_stats: NULL handle");
  HILO_REQUIRE(batch >= 0 && n_samples >= 2, "hilo_pf_stats: need batch >= 0 and n_samples >= 2");
  if (batch == 0) return HILO_OK;
  HILO_REQUIRE(X && Y && x_mean && y_mean && P && x_min && x_max, "hilo_pf_stats: NULL argument");
  HILO_REQUIRE(kf->nx <= 16, "hilo_pf_stats: built for up to 16 states");
  HILO_HIP_CHECK(hipSetDevice(kf->device));
  const int nye = kf->ny > 0 ? kf->ny : kf->nx;
  hipLaunchKernelGGL(pf_stats_kernel, dim3((unsigned)batch), dim3(PF_TPB), 0, (hipStream_t)stream, n_samples, kf->nx, nye, X, Y, add,
                     x_mean, y_mean, P, x_min, x_max);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}
