// Run-time compilation of NMPC problems for models that are not in the compiled zoo: host-side interface.
//
// The reference builds its solver from a symbolic model at `setup()` (`ca.nlpsol` on the CasADi graph,
// hilo_mpc/modules/controller/mpc.py:1778-1787).  The counterpart here: the host hands the model (and the problem's
// free-form functions) as HIP source of functors shaped like csrc/hilo_models.h; this module wraps it into a translation
// unit that instantiates one of the engine's policies for it, compiles it with hiprtc for gfx950 against the engine
// headers that ship next to the library, caches the code object on disk, and loads the kernels.
#pragma once
#include <string>

#include "hilo_common.h"

namespace hilo {

struct OcpConst;
struct OcpExtra;

enum JitPolicy : int {
  JIT_TRACK = 0,   // NmpcTrack<UserModel, BIG>          (hilo_nmpc_track.h) - identical code path to the zoo models
  JIT_GEN = 1,     // NmpcGen<UserModel, NTH, NE, NC, BIG> (hilo_nmpc_gen.h): compiles, not used by the host code
  JIT_USER = 2,    // NmpcUser<UserModel, UserFun, UserCfg> (hilo_nmpc_user.h) - the general policy
  JIT_MHE = 3,     // MheNoise<UserModel, SYM, COLL_D>     (hilo_mhe_policy.h) - moving-horizon estimator with state noise
};

struct JitRequest {
  std::string user_source;   // defines `UserModel` (struct or alias of a zoo functor) and, if has_fun, `UserFun`
  int policy = JIT_TRACK;
  int nth = 0, ne = 0, nc = 0, coll_d = 0, N = 1;
  int nq = 0;                    // JIT_USER: accumulator states of custom constraint rows (hilo_nmpc_user.h)
  bool hold = false, cont = false, tv = false, big = false, has_fun = false;
  bool sym = true;               // JIT_TRACK: symbolic model derivatives when the source provides them (one RK step per interval)
  bool mhe_gen = false;          // JIT_MHE: the general estimator policy MheGen (parameters as states, optional state noise)
  bool mhe_noise = true;         // JIT_MHE with mhe_gen: state noise variables present
  unsigned wz_mask[24] = {0};    // JIT_USER with has_wz_mask: row masks of the non-zero stage weights (bit j of row i: Wz[i][j] != 0) -
  bool has_wz_mask = false;      // compile-time constants of the unit, the quadratic form of the Lagrange term unrolls over them
  bool private_module = false;   // load a module of its own (its learned-term table belongs to ONE handle); the code object
                                 // still comes from the cache
};

struct JitKernels {
  hipFunction_t solve = nullptr, plant = nullptr, coll_out = nullptr;
  hipModule_t owned = nullptr;         // set for a private module: jit_unload() when the handle goes
  const double** gp_table = nullptr;   // device address of the module's hilo_user_gp[4] (learned terms of the user model)
  int dims[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // nx, nu, np, ny, discrete, lds bytes of the solve kernel, engine NX, engine NU
};

// compile (or fetch from the in-memory / on-disk cache) and load on `device`; on failure hilo_last_error() holds the compiler log
int jit_nmpc_kernels(const JitRequest& r, int device, JitKernels* out);

void jit_unload(JitKernels* k);

// Kalman / extended / unscented filter kernels of a model given as source (`UserModel`, same shape as for the controllers):
// f[UKF][MODE] with MODE 0 = predict, 1 = update, 2 = fused step (csrc/hilo_kf_kernel.h::kf_body); launch arguments are those
// of kf_kernel.  dims = nx, nu, np, ny, discrete.
struct JitKfKernels {
  hipFunction_t f[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  hipFunction_t pf = nullptr;   // particle-filter function of the same model (hilo_kf_kernel.h::pf_body)
  hipFunction_t multi[2] = {nullptr, nullptr};   // several fused steps per launch (kf_multi_body): [UKF]
  hipFunction_t team[2] = {nullptr, nullptr};    // the same on a team of lanes per instance (kf_team_body): small batches
  int dims[5] = {0, 0, 0, 0, 0};
  const double** gp_table = nullptr;   // device address of the module's hilo_user_gp[4] (learned terms of the user model)
  hipModule_t owned = nullptr;         // a module instance of the filter's own (private_module): unloaded with the handle
};
// private_module: load an instance of the module for this caller alone (its learned-term table is written per filter)
int jit_kf_kernels(const std::string& user_source, int device, JitKfKernels* out, bool compile_only = false,
                   bool private_module = false);
void jit_kf_unload(JitKfKernels* k);

// launch helpers (hipModuleLaunchKernel)
int jit_launch_solve(hipFunction_t f, const OcpConst* dev, int64_t batch, const double* x0, const double* par, int64_t par_stride,
                     const double* sdata, int64_t sd_stride, const double* v0, int64_t v0_stride, double* v_opt, double* f_opt,
                     double* lam_g, double* first, int32_t* status, int32_t* iters, double* kkt, long long* prof, double* ws,
                     hipStream_t s, OcpExtra ex);
int jit_launch_plant(hipFunction_t f, const OcpConst* dev, int64_t batch, const double* x, const double* u, const double* par,
                     int64_t par_stride, double* xn, hipStream_t s);
int jit_launch_coll_out(hipFunction_t f, const OcpConst* dev, int64_t batch, int N, const double* vc, const double* lamc,
                        const double* par, int64_t par_stride, const double* sdata, int64_t sd_stride, double* v, double* lam_g,
                        hipStream_t s);

}  // namespace hilo
