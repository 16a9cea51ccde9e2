// Run-time compilation of user problems with hiprtc (see hilo_jit.h).
//
// Pipeline:  translation unit = configuration macros + #include of the engine policy header + the user's functor source +
// three extern "C" kernels (solve, plant step, collocation output)  ->  hiprtcCompileProgram for gfx950 with the engine
// headers on the include path (csrc/ next to the library)  ->  code object cached as <hash>.hsaco  ->  hipModuleLoadData.
// The hash covers the translation unit, the compile options and the text of every engine header, so an edited engine never
// meets a stale code object.  The horizon is a compile-time constant of the unit: its LDS block is a static array (a
// module kernel cannot opt into more than 64 KB of dynamic LDS through hipFuncSetAttribute).
#include <dirent.h>
#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <map>
#include <mutex>
#include <vector>

#include "hilo_jit.h"
#include "hilo_ocp.h"

namespace hilo {

static std::string lib_dir() {
  Dl_info info;
  if (dladdr((const void*)&lib_dir, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    const size_t s = p.rfind('/');
    return s == std::string::npos ? std::string(".") : p.substr(0, s);
  }
  return ".";
}

static bool read_file(const std::string& path, std::string& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[1 << 16];
  size_t n;
  out.clear();
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
  fclose(f);
  return true;
}

static uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

// fingerprint of the engine headers (sorted by name)
static int headers_fingerprint(const std::string& csrc, uint64_t* fp) {
  std::vector<std::string> names;
  DIR* d = opendir(csrc.c_str());
  if (!d) return fail(HILO_EINVAL, "run-time compilation: engine headers not found in %s (they ship next to libhilo_hip.so)", csrc.c_str());
  while (dirent* e = readdir(d)) {
    const std::string n(e->d_name);
    if (n.size() > 2 && n.compare(n.size() - 2, 2, ".h") == 0) names.push_back(n);
  }
  closedir(d);
  for (size_t i = 0; i < names.size(); ++i)
    for (size_t j = i + 1; j < names.size(); ++j)
      if (names[j] < names[i]) std::swap(names[i], names[j]);
  uint64_t h = 1469598103934665603ull;
  for (const auto& n : names) {
    std::string text;
    if (!read_file(csrc + "/" + n, text)) return fail(HILO_EINVAL, "run-time compilation: cannot read %s/%s", csrc.c_str(), n.c_str());
    h = fnv1a(n, h);
    h = fnv1a(text, h);
  }
  *fp = h;
  return HILO_OK;
}

static std::string translation_unit(const JitRequest& r) {
  char cfg[1024];
  snprintf(cfg, sizeof(cfg),
           "#define HILO_OCP_TPB 64\n"
           "#define HILO_USER_POLICY %d\n"
           "#define HILO_USER_NTH %d\n#define HILO_USER_NE %d\n#define HILO_USER_NC %d\n#define HILO_USER_COLL_D %d\n"
           "#define HILO_USER_N %d\n#define HILO_USER_HOLD %d\n#define HILO_USER_CONT %d\n#define HILO_USER_TV %d\n"
           "#define HILO_USER_BIG %d\n#define HILO_USER_HAS_FUN %d\n#define HILO_USER_SYM %d\n#define HILO_USER_NQ %d\n",
           r.policy, r.nth, r.ne, r.nc, r.coll_d, r.N, (int)r.hold, (int)r.cont, (int)r.tv, (int)r.big, (int)r.has_fun, (int)r.sym, r.nq);
  std::string s(cfg);
  {
    std::string m = "#define HILO_USER_HAS_WZM ";
    m += r.has_wz_mask ? "1\n" : "0\n";
    m += "#define HILO_USER_WZM {";
    for (int i = 0; i < 24; ++i) m += (i ? ", " : "") + std::to_string(r.has_wz_mask ? r.wz_mask[i] : 0u) + "u";
    s += m + "}\n";
  }
  s += "#include \"hilo_nmpc_gen.h\"\n#include \"hilo_nmpc_track.h\"\n#include \"hilo_nmpc_user.h\"\n";
  // device pointers to the packed learned terms (gp_pack_se) the user source refers to as hilo_user_gp[k]; written by the host
  // after the module is loaded (a module with learned terms is private to its handle: JitRequest::private_module).
  s += "extern \"C\" { __device__ const double* hilo_user_gp[4]; }\n";   // unmangled: looked up with hipModuleGetGlobal
  s += "namespace hilo {\n";
  s += r.user_source;
  s += R"(
struct UserCfg {
  static constexpr int NTH = HILO_USER_NTH, NE = HILO_USER_NE, NC = HILO_USER_NC, COLL_D = HILO_USER_COLL_D, N = HILO_USER_N,
                       NQ = HILO_USER_NQ;
  static constexpr bool HOLD = HILO_USER_HOLD, CONT = HILO_USER_CONT, TV = HILO_USER_TV, BIG = HILO_USER_BIG;
  static constexpr bool HAS_WZM = HILO_USER_HAS_WZM;
  static constexpr unsigned WZM[24] = HILO_USER_WZM;
};
#if !HILO_USER_HAS_FUN
using UserFun = NoUserFun;
#endif
#if HILO_USER_POLICY == 0
using PB = NmpcTrack<UserModel, UserCfg::BIG, HILO_USER_SYM>;
#elif HILO_USER_POLICY == 1
using PB = NmpcGen<UserModel, UserCfg::NTH, UserCfg::NE, UserCfg::NC, UserCfg::BIG>;
#else
using PB = NmpcUser<UserModel, UserFun, UserCfg>;
#endif
using EngineT = Ocp<PB>;
constexpr size_t USER_LDS = EngineT::lds_doubles(UserCfg::N);
static_assert(USER_LDS * 8 <= 160 * 1024, "the iterate of this problem does not fit the 160 KB of LDS");

#ifndef HILO_USER_WAVES
#define HILO_USER_WAVES 1
#endif
extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HILO_USER_WAVES, HILO_USER_WAVES)))
void hilo_user_solve(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ x0, const double* __restrict__ par,
                     int64_t par_stride, const double* __restrict__ sdata, int64_t sd_stride, const double* __restrict__ v0,
                     int64_t v0_stride, double* __restrict__ v_opt, double* __restrict__ f_opt, double* __restrict__ lam_g,
                     double* __restrict__ first, int32_t* __restrict__ status, int32_t* __restrict__ iters, double* __restrict__ kkt,
                     long long* __restrict__ prof, double* __restrict__ ws, const OcpExtra ex) {
  __shared__ double lds[USER_LDS];
  ocp_solve_body<PB, 64>((lds_double*)lds, pcg, batch, x0, par, par_stride, sdata, sd_stride, v0, v0_stride, 0, 0, v_opt, f_opt,
                         lam_g, first, 0, status, iters, kkt, prof, ws, ex);
}

extern "C" __global__ void hilo_user_plant(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ x,
                                           const double* __restrict__ u, const double* __restrict__ par, int64_t par_stride,
                                           double* __restrict__ xn) {
  user_plant_step<UserModel>(pcg, batch, x, u, par, par_stride, xn);
}

extern "C" __global__ void hilo_user_coll_out(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ vc,
                                              const double* __restrict__ lamc, const double* __restrict__ par, int64_t par_stride,
                                              const double* __restrict__ sdata, int64_t sd_stride, double* __restrict__ v,
                                              double* __restrict__ lam_g) {
#if HILO_USER_POLICY == 2 && HILO_USER_COLL_D > 0
  PB::coll_output(pcg, batch, vc, lamc, par, par_stride, sdata, sd_stride, v, lam_g);
#elif HILO_USER_POLICY == 2
  PB::erk_dae_output(pcg, batch, vc, lamc, par, par_stride, sdata, sd_stride, v, lam_g);   // (algebraic states under explicit Runge-Kutta)
#endif
}

// dimensions of what was compiled, read back by the host as a consistency check
extern "C" __global__ void hilo_user_info(int* out) {
  out[0] = UserModel::NX; out[1] = UserModel::NU; out[2] = UserModel::NP; out[3] = UserModel::NY;
  out[4] = UserModel::DISCRETE ? 1 : 0; out[5] = (int)(USER_LDS * 8); out[6] = PB::NX; out[7] = PB::NU;
}
}  // namespace hilo
)";
  return s;
}

// The moving-horizon estimator's policy for a model given as source (csrc/hilo_mhe_policy.h): same kernel names as the
// controllers' unit, so that loading and launching are shared.  v carries the parameter prefix of the reference's layout
// (mhe.py:614-623): rows of v0 and v_opt are [p (NP) | x | w]; `first` receives x_N un-scaled (mhe.py:381-384).
static std::string translation_unit_mhe(const JitRequest& r) {
  char cfg[512];
  snprintf(cfg, sizeof(cfg), "#define HILO_OCP_TPB 64\n#define HILO_USER_N %d\n#define HILO_USER_COLL_D %d\n#define HILO_USER_SYM %d\n"
                             "#define HILO_USER_MHE_GEN %d\n#define HILO_USER_MHE_NOISE %d\n#define HILO_USER_HAS_FUN %d\n#define HILO_USER_NC %d\n",
           r.N, r.coll_d, (int)r.sym, (int)r.mhe_gen, (int)r.mhe_noise, (int)r.has_fun, r.nc);
  std::string s(cfg);
  s += "#include \"hilo_mhe_policy.h\"\n";
  s += "extern \"C\" { __device__ const double* hilo_user_gp[4]; }\n";
  s += "namespace hilo {\n";
  s += r.user_source;
  s += R"(
#if HILO_USER_MHE_GEN
#if !HILO_USER_HAS_FUN
using UserFun = NoMheFun;
#endif
using PB = MheGen<UserModel, HILO_USER_COLL_D, HILO_USER_MHE_NOISE, UserFun, HILO_USER_NC>;
constexpr int V_PREFIX = 0;             // engine-layout rows [xa | w] (the host converts, hilo_mhe.hip)
#else
using PB = MheNoise<UserModel, HILO_USER_SYM, HILO_USER_COLL_D>;
constexpr int V_PREFIX = UserModel::NP;
#endif
using EngineT = Ocp<PB>;
constexpr size_t USER_LDS = EngineT::lds_doubles(HILO_USER_N);
static_assert(USER_LDS * 8 <= 160 * 1024, "the iterate of this estimation window does not fit the 160 KB of LDS");

extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void hilo_user_solve(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ x0, const double* __restrict__ par,
                     int64_t par_stride, const double* __restrict__ sdata, int64_t sd_stride, const double* __restrict__ v0,
                     int64_t v0_stride, double* __restrict__ v_opt, double* __restrict__ f_opt, double* __restrict__ lam_g,
                     double* __restrict__ first, int32_t* __restrict__ status, int32_t* __restrict__ iters, double* __restrict__ kkt,
                     long long* __restrict__ prof, double* __restrict__ ws, const OcpExtra ex) {
  __shared__ double lds[USER_LDS];
  ocp_solve_body<PB, 64>((lds_double*)lds, pcg, batch, x0, par, par_stride, sdata, sd_stride, v0, v0_stride, V_PREFIX,
                         V_PREFIX, v_opt, f_opt, lam_g, first, 1, status, iters, kkt, prof, ws, ex);
}

extern "C" __global__ void hilo_user_plant(const OcpConst* __restrict__, int64_t, const double* __restrict__, const double* __restrict__,
                                           const double* __restrict__, int64_t, double* __restrict__) {}

extern "C" __global__ void hilo_user_coll_out(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ vc,
                                              const double* __restrict__ lamc, const double* __restrict__ par, int64_t par_stride,
                                              const double* __restrict__ sdata, int64_t sd_stride, double* __restrict__ v,
                                              double* __restrict__ lam_g) {
#if HILO_USER_MHE_GEN
  // general estimator: engine layout in, reference layout out; `par` carries the x_opt output buffer (hilo_mhe.hip)
  mhe_gen_output<UserModel, HILO_USER_COLL_D, HILO_USER_MHE_NOISE, UserFun, HILO_USER_NC>(pcg, batch, vc, lamc, sdata, sd_stride, v, lam_g,
                                                                                          (double*)par);
#elif HILO_USER_COLL_D > 0
  mhe_coll_output<UserModel, HILO_USER_COLL_D>(pcg, batch, vc, lamc, par, par_stride, sdata, sd_stride, v, lam_g);
#endif
}

extern "C" __global__ void hilo_user_info(int* out) {
  out[0] = UserModel::NX; out[1] = UserModel::NU; out[2] = UserModel::NP; out[3] = UserModel::NY;
  out[4] = UserModel::DISCRETE ? 1 : 0; out[5] = (int)(USER_LDS * 8); out[6] = PB::NX; out[7] = PB::NU;
}
}  // namespace hilo
)";
  return s;
}

struct LoadedModule {
  hipModule_t mod;
  JitKernels k;
};
static std::mutex g_mu;
static std::map<std::string, LoadedModule> g_loaded;   // key: hash + device

static int compile(const std::string& tu, const std::vector<std::string>& opts, std::vector<char>& code) {
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, tu.c_str(), "hilo_user.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    return fail(HILO_EHIP, "hiprtcCreateProgram failed");
  std::vector<std::string> all(opts);
  if (const char* e = getenv("HILO_JIT_EXTRA_OPTS")) {   // developer knob: further compiler options, space separated (not part of the cache key)
    std::string w;
    for (const char* c = e;; ++c) {
      if (*c == ' ' || *c == '\0') {
        if (!w.empty()) all.push_back(w);
        w.clear();
        if (*c == '\0') break;
      } else w += *c;
    }
  }
  std::vector<const char*> o;
  for (const auto& s : all) o.push_back(s.c_str());
  const hiprtcResult r = hiprtcCompileProgram(prog, (int)o.size(), o.data());
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    // the head of the log names the first error: keep what fits the error buffer
    const int rc = fail(HILO_EINVAL, "run-time compilation of the user problem failed (%s): %.380s", hiprtcGetErrorString(r), log.c_str());
    hiprtcDestroyProgram(&prog);
    return rc;
  }
  size_t n = 0;
  hiprtcGetCodeSize(prog, &n);
  code.resize(n);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  return HILO_OK;
}

// key of a translation unit (hash of the unit, the options, the engine headers, the compiler version)
static int unit_key(const std::string& tu, std::vector<std::string>* opts, std::string* key) {
  const std::string dir = lib_dir(), csrc = dir + "/csrc";
  uint64_t hfp = 0;
  int rc = headers_fingerprint(csrc, &hfp);
  if (rc) return rc;
  // (the library build adds -mllvm -amdgpu-mfma-vgpr-form=1 for the engine's units, _build.py; the compiler inside hiprtc does not
  // know that option and ends the PROCESS on it - run-time compiled problems keep the accumulator-register form of the products)
  *opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + csrc};
  uint64_t h = fnv1a(tu, hfp);
  for (const auto& o : *opts) h = fnv1a(o.substr(0, 2) == "-I" ? std::string("-I") : o, h);
  int rtc_major = 0, rtc_minor = 0;
  hiprtcVersion(&rtc_major, &rtc_minor);
  char k[64];
  snprintf(k, sizeof(k), "%016llx_%d_%d", (unsigned long long)h, rtc_major, rtc_minor);
  *key = k;
  return HILO_OK;
}

// code object of a translation unit: from the cache (HILO_JIT_CACHE or <library dir>/jit_cache), else compiled and cached
static int unit_code(const std::string& tu, const std::vector<std::string>& opts, const std::string& key, std::vector<char>& code,
                     std::string* cpath_out) {
  const char* env = getenv("HILO_JIT_CACHE");
  const std::string cdir = env && *env ? std::string(env) : lib_dir() + "/jit_cache";
  const std::string cpath = cdir + "/" + key + ".hsaco";
  *cpath_out = cpath;
  std::string cached;
  if (read_file(cpath, cached) && !cached.empty()) {
    code.assign(cached.begin(), cached.end());
    return HILO_OK;
  }
  int rc = compile(tu, opts, code);
  if (rc) return rc;
  mkdir(cdir.c_str(), 0755);
  const std::string tmp = cpath + "." + std::to_string((long)getpid()) + ".tmp";
  if (FILE* f = fopen(tmp.c_str(), "wb")) {   // best effort: a read-only tree only costs the recompilation
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), cpath.c_str()) != 0) unlink(tmp.c_str());
  }
  return HILO_OK;
}

int jit_nmpc_kernels(const JitRequest& r, int device, JitKernels* out) {
  const std::string tu = r.policy == JIT_MHE ? translation_unit_mhe(r) : translation_unit(r);
  std::vector<std::string> opts;
  std::string key, cpath;
  int rc = unit_key(tu, &opts, &key);
  if (rc) return rc;
  const std::string mkey = key + "@" + std::to_string(device);
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_loaded.find(mkey);
  if (!r.private_module && it != g_loaded.end()) { *out = it->second.k; return HILO_OK; }
  std::vector<char> code;
  rc = unit_code(tu, opts, key, code, &cpath);
  if (rc) return rc;
  if (getenv("HILO_JIT_COMPILE_ONLY")) {   // pre-warming the cache on a machine without a GPU (__graft_entry__.build)
    *out = JitKernels();
    return HILO_OK;
  }
  HILO_HIP_CHECK(hipSetDevice(device));
  LoadedModule m;
  hipError_t e = hipModuleLoadData(&m.mod, code.data());
  if (e != hipSuccess) {
    unlink(cpath.c_str());   // a damaged cache entry must not stick
    return fail(HILO_EHIP, "hipModuleLoadData of the run-time compiled problem failed: %s", hipGetErrorString(e));
  }
  hipFunction_t info = nullptr;
  HILO_HIP_CHECK(hipModuleGetFunction(&m.k.solve, m.mod, "hilo_user_solve"));
  HILO_HIP_CHECK(hipModuleGetFunction(&m.k.plant, m.mod, "hilo_user_plant"));
  HILO_HIP_CHECK(hipModuleGetFunction(&m.k.coll_out, m.mod, "hilo_user_coll_out"));
  HILO_HIP_CHECK(hipModuleGetFunction(&info, m.mod, "hilo_user_info"));
  {
    hipDeviceptr_t gptr = nullptr;
    size_t gbytes = 0;
    HILO_HIP_CHECK(hipModuleGetGlobal(&gptr, &gbytes, m.mod, "hilo_user_gp"));
    m.k.gp_table = (const double**)gptr;
  }
  int* dinfo = nullptr;
  HILO_HIP_CHECK(hipMalloc((void**)&dinfo, sizeof(int) * 8));
  void* args[] = {&dinfo};
  HILO_HIP_CHECK(hipModuleLaunchKernel(info, 1, 1, 1, 1, 1, 1, 0, nullptr, args, nullptr));
  HILO_HIP_CHECK(hipMemcpy(m.k.dims, dinfo, sizeof(int) * 8, hipMemcpyDeviceToHost));
  HILO_HIP_CHECK(hipFree(dinfo));
  if (r.private_module) m.k.owned = m.mod;
  else g_loaded[mkey] = m;
  *out = m.k;
  return HILO_OK;
}

// ---- filters of models written as expressions: kf_body<UserModel, UKF, MODE> behind six extern "C" kernels ----------------
static std::map<std::string, JitKfKernels> g_kf_loaded;

int jit_kf_kernels(const std::string& user_source, int device, JitKfKernels* out, bool compile_only, bool private_module) {
  std::string tu = "#include \"hilo_kf_kernel.h\"\nextern \"C\" { __device__ const double* hilo_user_gp[4]; }\nnamespace hilo {\n";
  tu += user_source;
  tu += R"(
}  // namespace hilo
using namespace hilo;
#define HILO_KF_ENTRY(name, UKF, MODE)                                                                                          \
  extern "C" __global__ __launch_bounds__(KF_TPB) void name(KfParams kp, int64_t batch, const double* __restrict__ in_tile,     \
                                                            const double* __restrict__ y, const double* __restrict__ up,        \
                                                            int64_t up_stride, const double* __restrict__ Q, int64_t q_stride,  \
                                                            const double* __restrict__ R, int64_t r_stride,                     \
                                                            double* __restrict__ out_tile, double* __restrict__ y_pred, int ipw) { \
    kf_body<UserModel, UKF, MODE>(kp, batch, in_tile, y, up, up_stride, Q, q_stride, R, r_stride, out_tile, y_pred, ipw);        \
  }
HILO_KF_ENTRY(hilo_user_kf_e0, false, 0)
HILO_KF_ENTRY(hilo_user_kf_e1, false, 1)
HILO_KF_ENTRY(hilo_user_kf_e2, false, 2)
HILO_KF_ENTRY(hilo_user_kf_u0, true, 0)
HILO_KF_ENTRY(hilo_user_kf_u1, true, 1)
HILO_KF_ENTRY(hilo_user_kf_u2, true, 2)
#define HILO_KF_MULTI(name, UKF)                                                                                                 \
  extern "C" __global__ __launch_bounds__(KF_TPB) void name(KfParams kp, int64_t batch, int steps,                              \
                                                            const double* __restrict__ in_tile, const double* __restrict__ y,   \
                                                            const double* __restrict__ up, int64_t up_stride, int64_t up_step,  \
                                                            const double* __restrict__ Q, int64_t q_stride,                     \
                                                            const double* __restrict__ R, int64_t r_stride,                     \
                                                            double* __restrict__ out_tile, int64_t out_step,                    \
                                                            double* __restrict__ y_pred, int ipw) {                             \
    kf_multi_body<UserModel, UKF>(kp, batch, steps, in_tile, y, up, up_stride, up_step, Q, q_stride, R, r_stride, out_tile,      \
                                  out_step, y_pred, ipw);                                                                        \
  }
HILO_KF_MULTI(hilo_user_kf_em, false)
HILO_KF_MULTI(hilo_user_kf_um, true)
#define HILO_KF_TEAM(name, UKF)                                                                                                  \
  extern "C" __global__ __launch_bounds__(KF_TPB) void name(KfParams kp, int64_t batch, int steps,                              \
                                                            const double* __restrict__ in_tile, const double* __restrict__ y,   \
                                                            const double* __restrict__ up, int64_t up_stride, int64_t up_step,  \
                                                            const double* __restrict__ Q, int64_t q_stride,                     \
                                                            const double* __restrict__ R, int64_t r_stride,                     \
                                                            double* __restrict__ out_tile, int64_t out_step,                    \
                                                            double* __restrict__ y_pred) {                                      \
    if constexpr (KfTeam<UserModel, UKF>::OK)                                                                                    \
      kf_team_body<UserModel, UKF>(kp, batch, steps, in_tile, y, up, up_stride, up_step, Q, q_stride, R, r_stride, out_tile,     \
                                   out_step, y_pred);                                                                            \
  }
HILO_KF_TEAM(hilo_user_kf_et, false)
HILO_KF_TEAM(hilo_user_kf_ut, true)
extern "C" __global__ __launch_bounds__(PF_TPB) void hilo_user_pf(KfParams kp, int n, const double* __restrict__ X,
                                                                  const double* __restrict__ y, const double* __restrict__ up,
                                                                  int64_t up_stride, const double* __restrict__ w,
                                                                  const double* __restrict__ v, const double* __restrict__ R,
                                                                  int64_t r_stride, double* __restrict__ Xp, double* __restrict__ Y,
                                                                  double* __restrict__ q) {
  constexpr int NX = UserModel::NX, NYE = UserModel::NY > 0 ? UserModel::NY : UserModel::NX;
  const int64_t b = blockIdx.x;
  pf_body<UserModel>(kp, n, X + b * n * NX, y + b * NYE, up + b * up_stride, w + b * n * NX, v + b * n * NYE, R + b * r_stride,
                     Xp + b * n * NX, Y + b * n * NYE, q + b * n);
}
extern "C" __global__ void hilo_user_kf_info(int* o) {
  o[0] = UserModel::NX; o[1] = UserModel::NU; o[2] = UserModel::NP; o[3] = UserModel::NY; o[4] = UserModel::DISCRETE ? 1 : 0;
}
)";
  std::vector<std::string> opts;
  std::string key, cpath;
  int rc = unit_key(tu, &opts, &key);
  if (rc) return rc;
  const std::string mkey = key + "@" + std::to_string(device);
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_kf_loaded.find(mkey);
  if (!private_module && it != g_kf_loaded.end()) { *out = it->second; return HILO_OK; }
  std::vector<char> code;
  rc = unit_code(tu, opts, key, code, &cpath);
  if (rc) return rc;
  if (compile_only || getenv("HILO_JIT_COMPILE_ONLY")) { *out = JitKfKernels(); return HILO_OK; }
  HILO_HIP_CHECK(hipSetDevice(device));
  hipModule_t mod;
  hipError_t e = hipModuleLoadData(&mod, code.data());
  if (e != hipSuccess) {
    unlink(cpath.c_str());
    return fail(HILO_EHIP, "hipModuleLoadData of the run-time compiled filter failed: %s", hipGetErrorString(e));
  }
  JitKfKernels k;
  const char* names[2][3] = {{"hilo_user_kf_e0", "hilo_user_kf_e1", "hilo_user_kf_e2"}, {"hilo_user_kf_u0", "hilo_user_kf_u1", "hilo_user_kf_u2"}};
  for (int u = 0; u < 2; ++u)
    for (int m = 0; m < 3; ++m) HILO_HIP_CHECK(hipModuleGetFunction(&k.f[u][m], mod, names[u][m]));
  HILO_HIP_CHECK(hipModuleGetFunction(&k.pf, mod, "hilo_user_pf"));
  HILO_HIP_CHECK(hipModuleGetFunction(&k.multi[0], mod, "hilo_user_kf_em"));
  HILO_HIP_CHECK(hipModuleGetFunction(&k.multi[1], mod, "hilo_user_kf_um"));
  HILO_HIP_CHECK(hipModuleGetFunction(&k.team[0], mod, "hilo_user_kf_et"));
  HILO_HIP_CHECK(hipModuleGetFunction(&k.team[1], mod, "hilo_user_kf_ut"));
  hipFunction_t info = nullptr;
  HILO_HIP_CHECK(hipModuleGetFunction(&info, mod, "hilo_user_kf_info"));
  int* dinfo = nullptr;
  HILO_HIP_CHECK(hipMalloc((void**)&dinfo, sizeof(int) * 8));
  void* args[] = {&dinfo};
  HILO_HIP_CHECK(hipModuleLaunchKernel(info, 1, 1, 1, 1, 1, 1, 0, nullptr, args, nullptr));
  HILO_HIP_CHECK(hipMemcpy(k.dims, dinfo, sizeof(int) * 5, hipMemcpyDeviceToHost));
  HILO_HIP_CHECK(hipFree(dinfo));
  {
    hipDeviceptr_t gptr = nullptr;
    size_t gbytes = 0;
    HILO_HIP_CHECK(hipModuleGetGlobal(&gptr, &gbytes, mod, "hilo_user_gp"));
    k.gp_table = (const double**)gptr;
  }
  if (private_module) k.owned = mod;
  else g_kf_loaded[mkey] = k;
  *out = k;
  return HILO_OK;
}

void jit_kf_unload(JitKfKernels* k) {
  if (k && k->owned) (void)hipModuleUnload(k->owned);
  if (k) *k = JitKfKernels();
}

void jit_unload(JitKernels* k) {
  if (k && k->owned) (void)hipModuleUnload(k->owned);
  if (k) *k = JitKernels();
}

int jit_launch_solve(hipFunction_t f, const OcpConst* dev, int64_t batch, const double* x0, const double* par, int64_t par_stride,
                     const double* sdata, int64_t sd_stride, const double* v0, int64_t v0_stride, double* v_opt, double* f_opt,
                     double* lam_g, double* first, int32_t* status, int32_t* iters, double* kkt, long long* prof, double* ws,
                     hipStream_t s, OcpExtra ex) {
  void* args[] = {&dev, &batch, &x0, &par, &par_stride, &sdata, &sd_stride, &v0, &v0_stride, &v_opt, &f_opt, &lam_g, &first,
                  &status, &iters, &kkt, &prof, &ws, &ex};
  HILO_HIP_CHECK(hipModuleLaunchKernel(f, (unsigned)batch, 1, 1, 64, 1, 1, 0, s, args, nullptr));
  return HILO_OK;
}

int jit_launch_plant(hipFunction_t f, const OcpConst* dev, int64_t batch, const double* x, const double* u, const double* par,
                     int64_t par_stride, double* xn, hipStream_t s) {
  void* args[] = {&dev, &batch, &x, &u, &par, &par_stride, &xn};
  HILO_HIP_CHECK(hipModuleLaunchKernel(f, (unsigned)((batch + 255) / 256), 1, 1, 256, 1, 1, 0, s, args, nullptr));
  return HILO_OK;
}

int jit_launch_coll_out(hipFunction_t f, const OcpConst* dev, int64_t batch, int N, const double* vc, const double* lamc,
                        const double* par, int64_t par_stride, const double* sdata, int64_t sd_stride, double* v, double* lam_g,
                        hipStream_t s) {
  void* args[] = {&dev, &batch, &vc, &lamc, &par, &par_stride, &sdata, &sd_stride, &v, &lam_g};
  const int64_t tot = batch * N;
  HILO_HIP_CHECK(hipModuleLaunchKernel(f, (unsigned)((tot + 63) / 64), 1, 1, 64, 1, 1, 0, s, args, nullptr));
  return HILO_OK;
}

}  // namespace hilo
