// Generic batched interior-point engine for stage-structured optimal-control NLPs (NMPC, MHE share it).
//
//   min  sum_{k<N} l_k(x_k, u_k) + V(x_N)   s.t.  x_{k+1} = F_k(x_k, u_k),  lb <= (x_k, u_k) <= ub,  [x_0 fixed]
//
// A problem class is a policy `PB` (dimensions + templated `dyn`, `stage_cost`, `term_cost`); every derivative the
// solver needs comes from evaluating the policy on second-order univariate Taylor numbers (Jet2) along nz(nz+1)/2
// directions per stage: value, J.v, grad l.v and v^T (hess l - lam^T hess F) v in one sweep, Hessian blocks by
// polarisation.  The reference gets the same quantities from CasADi's symbolic AD inside IPOPT.
//
// Algorithm (restated from Waechter & Biegler, Math. Program. 106 (2006), the method behind `ca.nlpsol(...,'ipopt')`,
// hilo_mpc/modules/controller/mpc.py:1780, hilo_mpc/modules/estimator/mhe.py:784): monotone barrier parameter (eq. 7),
// fraction to the boundary (8), filter line search (Alg. A; alpha_min eq. 23) with a simplified feasibility
// restoration, inertia correction (Alg. IC) where "inertia" = positivity of the Riccati pivots, scaled optimality
// error (5,6).  The KKT system is solved by a Riccati recursion over the horizon.
//
// Mapping: one workgroup (one wave of 64 lanes) per problem instance, the whole iterate in LDS; see DESIGN.md 5.1.
#pragma once
#include "hilo_colloc.h"
#include "hilo_common.h"
#include "hilo_models.h"

namespace hilo {

constexpr int OCP_MAXNX = 12, OCP_MAXNU = 8, OCP_MAXNZ = OCP_MAXNX + OCP_MAXNU;
constexpr int OCP_FILTER = 16;
constexpr int OCP_MAXNC = 16;  // nonlinear inequality rows per stage (stage rows + terminal rows)
#ifndef HILO_OCP_TPB
#define HILO_OCP_TPB 64
#endif
#ifndef HILO_OCP_MINW
#define HILO_OCP_MINW 1
#endif
// the four large phases (derivatives, Riccati, values, restoration) are real function calls: one copy of each in the
// binary (they have 2-3 call sites), shorter live ranges; -DHILO_OCP_INLINE_PHASES inlines them (tuning experiments)
#ifdef HILO_OCP_INLINE_PHASES
#define OCP_PHASE __attribute__((always_inline))
#else
#define OCP_PHASE __attribute__((noinline))
#endif
constexpr int OCP_TPB = HILO_OCP_TPB;  // threads per instance (one or more waves)
constexpr int OCP_NCOST = 2 * OCP_MAXNZ * OCP_MAXNZ + 4 * OCP_MAXNZ + 64;

struct OcpConst {
  int N, order, nsub, max_iter, acceptable_iter, flags;
  int Nc;              // control horizon (mpc.py:1629-1630: beyond it the last input is held); read by policies with NH > 0
  int tail_off;        // offset of the shared tail (slacks) in the rows the engine READS (v0, lbx, ubx) when it is not right behind
                       // the inputs: with collocation the reference puts the slacks behind the collocation blocks (mpc.py:1529); 0 = default
  double dt;
  double lbz[OCP_MAXNZ], ubz[OCP_MAXNZ];  // relaxed bounds of a stage's (x,u) slots (scaled); +-inf if none
  double x0lb[OCP_MAXNX], x0ub[OCP_MAXNX];  // flags bit 1: own box of x_0 (optimize(fix_x0=False, x0_lb=, x0_ub=), mpc.py:803-807)
  double sz[OCP_MAXNZ];                   // scaling of (x,u)
  // interior-point constants (IPOPT defaults)
  double tol, acceptable_tol, mu_init, kappa_eps, kappa_mu, theta_mu, tau_min, bound_push, bound_frac, s_max,
      kappa_sigma, gamma_theta, gamma_phi, delta_ls, s_theta, s_phi, eta_phi, theta_min_fact, theta_max_fact,
      delta_w_min, delta_w_0, delta_w_max, kappa_w_minus, kappa_w_plus, kappa_w_plus_bar;
  const double* ext;  // learned-term data of the model in device memory (hilo_models.h GpExt) or NULL
  // general problems (path following, stage constraints; hilo_nmpc_gen.h)
  unsigned x0_free_mask;   // bit i: slot i of x_0 is a variable although x_0 is pinned (path variable, shared slack)
  unsigned k0_only_mask;   // bit i: the box of slot i only applies at stage 0 (shared slack carried as a constant state)
  int nc, nc_term;         // inequality rows of every stage; further rows that only exist at the last stage N-1 (terminal
                           // constraint on the integrated end state, mpc.py:1693-1700); nc + nc_term <= PB::NC
  double dlb[OCP_MAXNC], dub[OCP_MAXNC];  // their (relaxed) bounds; +-inf if none
  int n_con_ref, n_tcon_ref;              // rows per stage / terminal rows in the reference's g ...
  short row_ref[OCP_MAXNC], trow_ref[OCP_MAXNC];   // ... and where each active stage / terminal row sits there
  CollData coll;                          // collocation basis when the shooting map is the implicit one (hilo_colloc.h)
  // Taylor path: bit p of pair_mask = the pair direction e_i + e_j with p = dir_of(i, j, NZ) - NZ is swept (the Hessian entry
  // (i, j) of the interval's Lagrangian can be non-zero); cleared bits: entry taken as zero.  All ones by default.
  unsigned long long pair_mask[4];
  double bound_relax;                     // IPOPT's bound_relax_factor, for bounds that arrive per call (OcpExtra::lbx / ubx)
  // policy-defined cost data (weights, references, expression programs).  LAST member: an instance copies only the
  // PB::NCOST doubles its policy uses into LDS (40 KB per instance is the budget for four instances per CU)
  double cost[OCP_NCOST];
};

inline void ocp_default_options(OcpConst& c) {
  c.max_iter = 3000; c.acceptable_iter = 15;
  c.tol = 1e-8; c.acceptable_tol = 1e-6; c.mu_init = 0.1;
  c.kappa_eps = 10.0; c.kappa_mu = 0.2; c.theta_mu = 1.5; c.tau_min = 0.99;
  c.bound_push = 1e-2; c.bound_frac = 1e-2; c.s_max = 100.0; c.kappa_sigma = 1e10;
  c.gamma_theta = 1e-5; c.gamma_phi = 1e-8; c.delta_ls = 1.0; c.s_theta = 1.1; c.s_phi = 2.3; c.eta_phi = 1e-8;
  c.theta_min_fact = 1e-4; c.theta_max_fact = 1e4;
  c.delta_w_min = 1e-20; c.delta_w_0 = 1e-4; c.delta_w_max = 1e40; c.kappa_w_minus = 1.0 / 3; c.kappa_w_plus = 8.0;
  c.kappa_w_plus_bar = 100.0;
  for (auto& w : c.pair_mask) w = ~0ull;
}

// ---- block-wide reductions (result broadcast to every lane) ------------------------------------------------
struct OpSum { __device__ static double id() { return 0.0; } __device__ static double f(double a, double b) { return a + b; } };
// NaN-propagating maximum: fmax() silently drops NaN, which would let an error measure pass its tolerance with a NaN
// multiplier or residual somewhere in the iterate
__device__ __forceinline__ double nmax(double a, double b) { return (a != a || a > b) ? a : b; }
struct OpMax { __device__ static double id() { return -INFINITY; } __device__ static double f(double a, double b) { return nmax(a, b); } };
struct OpMin { __device__ static double id() { return INFINITY; } __device__ static double f(double a, double b) { return fmin(a, b); } };
struct OpMax2 { __device__ static double id() { return -INFINITY; } __device__ static double f(double a, double b) { return fmax(a, b); } };   // plain maximum

// wave-wide reduction without LDS traffic: DPP lane permutes inside each row of 16 (xor 1, xor 2, half-mirror, mirror), then
// the four row totals through v_readlane (scalar registers -> broadcast for free)
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  // (every lane of these permutations has a source lane: no "old" value to preserve, so none is initialised)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double read_lane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// Element loops: lane t handles e = t, t + T, ...  Written with a wave-uniform trip count and a predicated body instead of
// `for (e = t; e < n; e += T)`: the exit of a lane-strided loop is a join block that starts with the EXEC restore, and the
// ROCm 7.2 register allocator sometimes places the register saves of a following call in front of that restore, where they
// execute for no lane (see uni() below and tools/check_exec_prologue.py).  With a scalar loop branch the block after the
// loop is entered with every lane enabled.
#define OCP_FOR(e, n) \
  for (int e##_base = 0; e##_base < (n); e##_base += OCP_TPB) \
    if (const int e = e##_base + (int)threadIdx.x; e < (n))

// Wave-uniform values (one wave per instance: every solver decision is the same in all 64 lanes).  The compiler cannot
// see that for values that come back from a called phase or through memory, treats them as divergent and then lowers the
// solver's control flow with EXEC masks and keeps its scalars in vector registers.  v_readfirstlane pins them to scalar
// registers: scalar branches, and no vector live ranges across the divergent element loops (a register-allocator copy
// placed before the EXEC restore of such a loop's exit block lost the filter size of a constrained variant - see DESIGN.md).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
__device__ __forceinline__ double uni(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <class T>
__device__ __forceinline__ __attribute__((address_space(3))) T* uni(__attribute__((address_space(3))) T* p) {
  return (__attribute__((address_space(3))) T*)(size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)p);
}
template <class T>
__device__ __forceinline__ T* uni(T* p) {
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
// 1/sqrt(x) for x > 0: v_rsq_f64 seed (about 2^-26) and two Newton steps - the library rsqrt() is a ~100-cycle dependent
// chain that sits on the critical path of every Riccati stage
#ifndef HILO_RSQ_NEWTON
#define HILO_RSQ_NEWTON 2
#endif
__device__ __forceinline__ double rsq_fast(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * fma(-hx * y, y, 1.5);
  if (HILO_RSQ_NEWTON >= 2) y = y * fma(-hx * y, y, 1.5);
  return y;
}
typedef double v4d __attribute__((ext_vector_type(4)));   // accumulator of v_mfma_f64_16x16x4
struct FTheta { double f, theta, x; };  // objective and constraint violation of a point (x: sum of the caller's `extra` over the lanes)

template <class Op>
__device__ __forceinline__ double wave_reduce(double v) {
  v = Op::f(v, dpp_mov<0xB1>(v));   // quad_perm [1,0,3,2]
  v = Op::f(v, dpp_mov<0x4E>(v));   // quad_perm [2,3,0,1]
  v = Op::f(v, dpp_mov<0x141>(v));  // row_half_mirror
  v = Op::f(v, dpp_mov<0x140>(v));  // row_mirror
  return uni(Op::f(Op::f(read_lane(v, 0), read_lane(v, 16)), Op::f(read_lane(v, 32), read_lane(v, 48))));
}

// Several wave-wide reductions at once, level by level: the permutes of all values, then their operations - one reduction
// alone is a chain of dependent DPP moves and f64 operations (two wait states and the operation's latency per link, nothing else
// to issue with one wave per SIMD); n chains in lock step fill each other's gaps.  OPS: R_SUM / R_MAX (NaN-propagating) /
// R_MIN / R_MAX2 (plain) per value; results are wave-uniform (scalar registers).
enum { R_SUM = 0, R_MAX = 1, R_MIN = 2, R_MAX2 = 3 };
__device__ __forceinline__ double red_apply(int op, double a, double b) {
  switch (op) {
    case R_SUM: return a + b;
    case R_MAX: return nmax(a, b);
    case R_MIN: return fmin(a, b);
    default: return fmax(a, b);
  }
}
template <int... OPS>
struct WaveReduceN {
  static constexpr int n = sizeof...(OPS);
  template <int CTRL>
  __device__ __forceinline__ static void level(double* v) {
    constexpr int ops[n] = {OPS...};
    double t[n];
#pragma unroll
    for (int j = 0; j < n; ++j) t[j] = dpp_mov<CTRL>(v[j]);
#pragma unroll
    for (int j = 0; j < n; ++j) v[j] = red_apply(ops[j], v[j], t[j]);
  }
  __device__ __forceinline__ static void run(double* v) {
    static_assert(OCP_TPB == 64, "one wave per instance");
    constexpr int ops[n] = {OPS...};
    level<0xB1>(v);
    level<0x4E>(v);
    level<0x141>(v);
    level<0x140>(v);
    double a[n], b[n];
#pragma unroll
    for (int j = 0; j < n; ++j) {
      a[j] = red_apply(ops[j], read_lane(v[j], 0), read_lane(v[j], 16));
      b[j] = red_apply(ops[j], read_lane(v[j], 32), read_lane(v[j], 48));
    }
#pragma unroll
    for (int j = 0; j < n; ++j) v[j] = uni(red_apply(ops[j], a[j], b[j]));
  }
};

template <class Op>
__device__ __forceinline__ double block_reduce(double v, __attribute__((address_space(3))) double* scratch) {
  v = wave_reduce<Op>(v);
  if constexpr (OCP_TPB == 64) return v;
  const int nw = blockDim.x >> 6;
  if (nw == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = Op::id();
  for (int w = 0; w < nw; ++w) r = Op::f(r, scratch[w]);
  return r;
}

// optional members of a policy, with their defaults
//   NH     number of inputs carried as extra states so that they can be HELD beyond the control horizon (0: Nc == N)
//   FUSED  the policy evaluates the shooting map and the Lagrange term of an interval together (`dyn_cost`): the
//          continuous objective integrates the cost through the collocation states / Runge-Kutta stages of the map
// structural non-zeros of a model's generated derivatives (ModelSym<M>::XMASK / JMASK / HMASK, hilo_mpc_amd/symdiff.py); all ones
// for a model without them
template <class MS, class = void> struct sym_masks {
  static constexpr unsigned long long X = ~0ull, J = ~0ull, H = ~0ull;
};
template <class MS> struct sym_masks<MS, void_tt<decltype(MS::HAS_MASKS)>> {
  static constexpr unsigned long long X = MS::HAS_MASKS ? MS::XMASK : ~0ull, J = MS::HAS_MASKS ? MS::JMASK : ~0ull,
                                      H = MS::HAS_MASKS ? MS::HMASK : ~0ull;
};
template <class PB, class = void> struct pb_plant { static constexpr bool value = false; };
template <class PB> struct pb_plant<PB, void_tt<decltype(PB::PLANT)>> { static constexpr bool value = PB::PLANT; };
template <class PB, class = void> struct pb_symtab { static constexpr bool value = false; };
template <class PB> struct pb_symtab<PB, void_tt<decltype(PB::SYMTAB)>> { static constexpr bool value = PB::SYMTAB; };
template <class PB, class = void> struct pb_nh { static constexpr int value = 0; };
template <class PB> struct pb_nh<PB, void_tt<decltype(PB::NH)>> { static constexpr int value = PB::NH; };
template <class PB, class = void> struct pb_sym { static constexpr bool value = false; };
template <class PB> struct pb_sym<PB, void_tt<decltype(PB::SYM)>> { static constexpr bool value = PB::SYM; };
template <class PB, class = void> struct pb_sym_mhe { static constexpr bool value = false; };
template <class PB> struct pb_sym_mhe<PB, void_tt<decltype(PB::SYM_MHE)>> { static constexpr bool value = PB::SYM_MHE; };
// VEC_N: the horizon as a compile-time constant (run-time compiled policies) - with it a workspace-mode kernel can decide to keep
// the horizon-long VECTORS of the iterate in LDS (see Ocp::VEC_LDS); 0 = unknown
template <class PB, class = void> struct pb_vec_n { static constexpr int value = 0; };
template <class PB> struct pb_vec_n<PB, void_tt<decltype(PB::VEC_N)>> { static constexpr int value = PB::VEC_N; };
//   FUSED_CON  the inequality rows are evaluated together with the shooting map (`dyn_cost_con`): under collocation the reference
//          imposes the stage constraints at the collocation points too (mpc.py:1338-1356) - rows of the map's interior points
template <class PB, class = void> struct pb_fused_con { static constexpr bool value = false; };
template <class PB> struct pb_fused_con<PB, void_tt<decltype(PB::FUSED_CON)>> { static constexpr bool value = PB::FUSED_CON; };
//   PREP   doubles per interval of data the policy prepares ONCE per derivative evaluation (`prepare`) and every Taylor task of the
//          interval reads (`dyn_cost_prep`): an implicit shooting map's converged stage values and the factors of its Newton matrix
//          (collocation: the same for all directions of an interval - without it every direction repeats the Newton solve)
template <class PB, class = void> struct pb_prep { static constexpr int value = 0; };
template <class PB> struct pb_prep<PB, void_tt<decltype(PB::PREP)>> { static constexpr int value = PB::PREP; };
//   LAM_FIX  the policy adjusts the reported multipliers of the LAST shooting defect (`lam_fix`): rows the reference imposes on the
//          node variable x_N are rows on the integrated end state here (custom constraint functions, hilo_nmpc_user.h)
// ... or the lanes of a wave solve the collocation systems together: PB::coll_pass, states per interval in the workspace (PB::XCW)
template <class PB, class = void> struct pb_vec_max { static constexpr int value = 3; };
template <class PB> struct pb_vec_max<PB, void_tt<decltype(PB::VEC_MAX)>> { static constexpr int value = PB::VEC_MAX; };
template <class PB, class = void> struct pb_cstage { static constexpr int value = 0; };
template <class PB> struct pb_cstage<PB, void_tt<decltype(PB::CSTAGE)>> { static constexpr int value = PB::CSTAGE; };
template <class PB, class = void> struct pb_xcw { static constexpr int value = 0; };
template <class PB> struct pb_xcw<PB, void_tt<decltype(PB::XCW)>> { static constexpr int value = PB::XCW; };
template <class PB, class = void> struct pb_lam_fix { static constexpr bool value = false; };
template <class PB> struct pb_lam_fix<PB, void_tt<decltype(PB::LAM_FIX)>> { static constexpr bool value = PB::LAM_FIX; };
template <class PB, class = void> struct pb_fused { static constexpr bool value = false; };
template <class PB> struct pb_fused<PB, void_tt<decltype(PB::FUSED)>> { static constexpr bool value = PB::FUSED; };

// LDS / workspace footprint as plain functions of the dimensions (the host sizes run-time compiled problems with them)
__host__ __device__ constexpr size_t ocp_iter_doubles(int NX, int NU, int NC, int N) {
  // 9 slot vectors (Z Zt D zL zU grad sig lbA ubA), rbN; 4 defect-sized vectors (lam lamn c ct) + c0; the padded stage
  // blocks AB [N][NX][NZ+1] (column NZ: -c), W [N][NZ][NZ+1] (column NZ: right-hand side), Qd; the merged recursion outputs
  // P|p [N+1][NX][NX+1], K|kff [N][NU][NX+1], Acl|bcl [N][NX][NX+1]; inequality rows
  const size_t NZ = NX + NU, NDIR = NZ * (NZ + 1) / 2, S = (size_t)(N + 1) * NZ;
  return 9 * S + NX + 5 * (size_t)N * NX + (size_t)N * NX * (NZ + 1) + (size_t)N * NZ * (NZ + 1) + (size_t)(N + 1) * NDIR +
         (size_t)(N + 1) * NX * (NX + 1) + (size_t)N * NU * (NX + 1) + (size_t)N * NX * (NX + 1) + (size_t)N * NC * (14 + NZ) +
         (size_t)N * 4 * NX;   // an upper bound over the policies (Ocp<PB>::iter_doubles is the exact figure): direction table AND stage points
}
__host__ __device__ constexpr size_t ocp_fixed_doubles(int NX, int NU, int NCONST, int NPAR, int NSD, int NEXT, int N, int NPREP = 0) {
  const size_t NZ = NX + NU;
  return NCONST + NZ * NZ + NZ + (N + 1) + 2 * 16 + 16 + NPAR + (size_t)(N + 1) * (NSD > 0 ? NSD : 0) + 1 + NEXT +
         (NZ * (NZ + 1) / 2 + 2) / 2 + NPREP;
}

// Optional plumbing of a solve launch that saves separate kernels around it (all members may stay zero):
//   par2 / npar1  the per-instance parameter row comes from TWO arrays: entries [0, npar1) from `par` (row stride par_stride),
//                 the remaining NPAR - npar1 from `par2` (dense rows; NULL = zeros) - model parameters and the previous input
//                 of an NMPC step, without a packing kernel in front of the solve
//   v_copy        second copy of the solution rows (the handle's warm-start buffer, mpc.py:725-726) - no device copy after it
//   gather        row b of a [batch][gather_stride] fp64 table receives [first output (u_0) | status | iterations]: the send
//                 buffer of the per-step result gather (hilo_mpc_amd/dist.py)
//   lam_x / g     the other two vectors of the reference's solver result (mpc.py:722-723 keeps `sol` whole): bound multipliers in
//                 the layout of v (CasADi's sign: z_U - z_L; 0 for a pinned x_0) and the constraint values in the layout of lam_g
struct OcpExtra {
  const double* par2;
  int npar1;
  int gather_stride;
  double* v_copy;
  double* gather;
  double* lam_x;
  double* g;
  // lbx / ubx of the reference's solver call (`self._solver(x0=, lbx=, ubx=, ...)`, mpc.py:722): per instance and per call, rows in
  // the layout of v (scaled variables, row stride bx_stride); NULL = the bounds of the problem description.  IPOPT's bound
  // relaxation is applied here (pc.bound_relax).  The pinned x_0 slots keep following x0 (mpc.py:797-802 writes x0 into both).
  const double* lbx;
  const double* ubx;
  long long bx_stride;
  // closed loops whose plant IS the controller's model (benchmarks, simulations): row b receives x+ = Phi(x0_b, u_0, p_b), the map
  // of plant_step_kernel on the input just computed - the plant step fused into the solve (policies with PB::PLANT); may alias x0
  double* x_next;
};

#ifdef HILO_OCP_DPROF
__device__ long long g_dprof[32];
#define DTICK(i) { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long tn_ = clock64(); g_dprof[i] += tn_ - dt_; dt_ = tn_; } }
#define DTICK0 long long dt_ = clock64();
#define DTICKR dt_ = clock64();
#else
#define DTICK(i)
#define DTICK0
#define DTICKR
#endif

enum OcpPhase { PH_DERIV = 0, PH_ERR, PH_RICCATI, PH_STEP, PH_LS, PH_UPDATE, PH_NRIC, PH_NLS, PH_COUNT };

template <class PB>
struct Ocp {
  static constexpr int NX = PB::NX, NU = PB::NU, NZ = NX + NU, NDIR = NZ * (NZ + 1) / 2, NXDIR = NX * (NX + 1) / 2;
  static constexpr int NPAR = PB::NPAR > 0 ? PB::NPAR : 1, NSD = PB::NSD;
  static constexpr bool FIX_X0 = PB::FIX_X0;
  static constexpr int NH = pb_nh<PB>::value;          // held inputs (states NX-NH..NX-1), control horizon pc.Nc
  static constexpr bool FUSED = pb_fused<PB>::value;   // dyn_cost(): shooting map + Lagrange term in one evaluation
  static constexpr bool FUSED_CON = pb_fused_con<PB>::value;   // dyn_cost_con(): ... and the inequality rows
  static constexpr int PREP = pb_prep<PB>::value;              // per-interval data prepared once per derivative evaluation
  static constexpr int XCW = pb_xcw<PB>::value;                // > 0: cooperative collocation (PB::coll_pass, PB::dyn_cost_xc)
#ifndef HILO_OCP_PREPB
#define HILO_OCP_PREPB 3
#endif
  static constexpr int PREPB = XCW > 0 ? HILO_OCP_PREPB : 1;   // per-interval blocks staged in LDS at a time
  // the staging area `prepl`: the blocks of the direction passes, or (never at the same time) the cooperative pass's states + factors
  static constexpr int PREPL = PREP * PREPB > pb_cstage<PB>::value ? PREP * PREPB : pb_cstage<PB>::value;
  // model derivatives as generated straight-line code (ModelSym<PB::Model>, csrc/hilo_models_sym.h / codegen): second-order
  // adjoint through the Runge-Kutta stages instead of Taylor sweeps per direction pair (eval_derivs_sym)
  static constexpr bool SYM = pb_sym<PB>::value;
  static constexpr bool SYM_MHE = pb_sym_mhe<PB>::value;   // the same for the moving-horizon estimator's policy (eval_derivs_sym_mhe)
  // model with a learned term: lanes that evaluate the dynamics at the same point share its kernel sum (GpExt)
  static constexpr bool COOP = PB::COOP;
  // cooperative models: exchange scratch of the kernel sum (12 doubles per lane) + the learned term's table (hilo_models.h GpExt)
  static constexpr int NEXT_SCR = COOP ? 12 * OCP_TPB : 0;
  static constexpr int NEXT = COOP ? NEXT_SCR + GP2_HDR + 3 * GP2_MAXN : 0;
  // nonlinear inequality rows per stage (compile-time capacity; pc.nc of them are active): IPOPT's slack form
  //   d(x_k,u_k) - s_k = 0,  dlb <= s_k <= dub,  multipliers nu (equality), vL/vU (slack bounds)
  static constexpr int NC = PB::NC;
  static_assert(NC <= OCP_MAXNC, "too many constraint rows");
  // reference decision-vector layout [x (NXV per stage) | u | shared tail (NX - NXV)]; x_0 measurement width; inputs returned
  static constexpr int NXV = PB::NXV, NX0 = PB::NX0, NU0 = PB::NU0;
  static_assert(!COOP || (OCP_TPB == 64 && NDIR <= 64), "cooperative models need one wave per instance");
  // stage products of the Riccati recursion on the f64 matrix cores (one wave per instance, a stage's z and the
  // right-hand-side column fit the 16 columns of v_mfma_f64_16x16x4)
  static constexpr bool MFMA_STAGE = OCP_TPB == 64 && NZ + 1 <= 16 && NX <= 16;
  static constexpr int NCONST = (int)((__builtin_offsetof(OcpConst, cost) + sizeof(double) * PB::NCOST + 7) / 8);
  static constexpr int ABP = NZ + 1, WP = NZ + 1, PP = NX + 1;   // row pitches of the padded blocks (see Lds)

  // Storage of the iterate: LDS (default) or, for problems that do not fit (long horizons, wide stages), a per-instance
  // workspace in global memory (PB::BIG; L2-resident, same code path, longer latencies).  Problem constants, the pivot
  // block, reduction scratch and the per-instance data always stay in LDS.
  // (the workspace pointers carry the GLOBAL address space: through generic pointers every access would be a flat instruction,
  // which counts on both the memory and the LDS counter and so can only be waited for with both at zero - one access in flight)
  static constexpr bool BIG = PB::BIG;
  using dp = cond_t<BIG, gbl_double*, lds_double*>;
  using cdp = cond_t<BIG, gbl_cdouble*, lds_cdouble*>;
  // Mixed layout of the workspace mode: when the horizon is a compile-time constant, the vectors every small phase walks over
  // stay in LDS as far as FOUR instances per CU still fit (40 KB each, constants included) - in this order: primal point, trial
  // point, step, bound multipliers and effective box (7 slot vectors); equality multipliers and the two defect vectors; the
  // gradient.  The matrices of the stages go to the workspace.  (Measured on C5: all eleven vectors in LDS cost a fourth resident
  // instance per CU and lost more than the shorter phases gained - 59.9 k against 67.5 k steps/s.)
  static constexpr int VEC_N = pb_vec_n<PB>::value;
  __host__ __device__ static constexpr size_t vec_doubles_level(int N, int level) {
    return (level >= 1 ? 7 * (size_t)(N + 1) * NZ : 0) + (level >= 2 ? 3 * (size_t)N * NX : 0) + (level >= 3 ? (size_t)(N + 1) * NZ : 0);
  }
  static constexpr size_t VEC_BUDGET = 40 * 1024 - 64;
  static constexpr bool vec_fits(int level) {
    return (ocp_fixed_doubles(NX, NU, NCONST, NPAR, NSD, NEXT, VEC_N, PREPL) + vec_doubles_level(VEC_N, level)) * sizeof(double) <= VEC_BUDGET;
  }
  // (a policy may cap the level - PB::VEC_MAX: a problem whose long phases are latency bound trades the LDS-resident vectors for
  // a second wave per SIMD)
  static constexpr int VEC_FIT = !(BIG && VEC_N > 0) ? 0 : (vec_fits(3) ? 3 : (vec_fits(2) ? 2 : (vec_fits(1) ? 1 : 0)));
  static constexpr int VEC_LEVEL = VEC_FIT < pb_vec_max<PB>::value ? VEC_FIT : pb_vec_max<PB>::value;
  static constexpr bool VEC_LDS = VEC_LEVEL > 0;
  __host__ __device__ static constexpr size_t vec_doubles(int N) { return vec_doubles_level(N, VEC_LEVEL); }
  using vp = cond_t<BIG && VEC_LEVEL < 1, gbl_double*, lds_double*>;      // Z Zt D zL zU lbA ubA
  using cvp = cond_t<BIG && VEC_LEVEL < 1, gbl_cdouble*, lds_cdouble*>;
  using ep = cond_t<BIG && VEC_LEVEL < 2, gbl_double*, lds_double*>;      // lam c ct
  using gp_t = cond_t<BIG && VEC_LEVEL < 3, gbl_double*, lds_double*>;    // grad
  struct Lds {
    const __attribute__((address_space(3))) OcpConst* pc;
    // AB: [N][NX][ABP] = [A B | -c];  W: [N][NZ][WP] = [Hessian block (+ Sigma on the diagonal after kkt_pass) | rhs];
    // P: [N+1][NX][PP] = [P_k | p_k];  Kg: [N][NU][PP] = [K_k | kff_k];  Acl: [N][NX][PP] = [A + B K | B kff - c];  rbN: rhs of x_N
    vp Z, Zt, D, zL, zU;
    vp lbA, ubA;  // effective box of every slot: -inf / +inf where there is no bound or the slot is not a variable
    ep lam, c, ct;
    gp_t grad;
    dp lamn, AB, W, Qd, P, Kg, sig, rbN, Acl;
    dp Xs;   // SYM policies: Runge-Kutta stage points [N][4][NX] of the last values-only evaluation (reused by the derivative phase)
    dp prep; // [N][PREP]: what PB::prepare leaves for the Taylor tasks of an interval
    dp xc;   // [N][XCW]: collocation states of the last values-only evaluation (cooperative collocation)
    lds_double* prepl;   // [PREP]: the block of the interval whose directions are being swept (staged in LDS: every lane reads all of it)
    lds_double* gpc;   // cooperative models: [N][4][6] value / gradient / Hessian of the learned term at the stage points (GpExt::cache)
    dp cs, cst, cnu, cnun, cvL, cvU, cdvL, cdvU, cds, cd, csig, crb, Jd;  // [N][NC] (Jd: [N][NC][NZ])
    dp c0, cd0, cdt;  // second-order correction: saved defects / row values, row values at the trial point
    lds_double *Mm, *mm, *fk, *filt, *red, *par, *sd, *ext;
    __attribute__((address_space(3))) int* dirs;   // [0] = number of swept directions per interval, [1..] = their indices d
  };
  static_assert(OCP_FILTER == 16, "ocp_fixed_doubles assumes a filter of 16 entries");
  // direction table of the Hessian blocks: policies with symbolic derivatives write the blocks directly and keep only the
  // terminal cost's directions; stage points kept for reuse: SYM policies only
  __host__ __device__ static constexpr size_t qd_doubles(int N) { return (SYM || SYM_MHE) ? (size_t)NXDIR : (size_t)(N + 1) * NDIR; }
  __host__ __device__ static constexpr size_t xs_doubles(int N) { return SYM ? (size_t)N * 4 * NX : (COOP ? (size_t)N * 24 : 0); }
  __host__ __device__ static constexpr size_t iter_doubles(int N) {  // the iterate (LDS or workspace); <= ocp_iter_doubles + xs
    return ocp_iter_doubles(NX, NU, NC, N) - (size_t)(N + 1) * NDIR - (size_t)N * 4 * NX + qd_doubles(N) + xs_doubles(N) +
           (size_t)N * PREP + (size_t)N * XCW;
  }
  __device__ static dp qd_term(const Lds l, int N) { return (SYM || SYM_MHE) ? l.Qd : l.Qd + (size_t)N * NDIR; }
  __host__ __device__ static constexpr size_t fixed_doubles(int N) {  // always LDS
    return ocp_fixed_doubles(NX, NU, NCONST, NPAR, NSD, NEXT, N, PREPL);
  }
  __host__ __device__ static constexpr size_t lds_doubles(int N) {
    return fixed_doubles(N) + (BIG ? (VEC_LDS ? vec_doubles(N) : 0) : iter_doubles(N));
  }
  __host__ __device__ static constexpr size_t ws_doubles(int N) { return BIG ? iter_doubles(N) - (VEC_LDS ? vec_doubles(N) : 0) : 0; }
  // The non-inlined phases take (LDS base, workspace) and re-derive the pointer table: a struct argument would be
  // passed through scratch memory per lane (measured: 380 MB of scratch writes per 1024-instance launch).
  __device__ static int horizon_of(lds_double* base) {
    return reinterpret_cast<const __attribute__((address_space(3))) OcpConst*>(base)->N;
  }
  __device__ static Lds carve(lds_double* base, double* ws) { return carve(base, ws, horizon_of(base)); }
  __device__ static Lds carve(lds_double* base, double* ws, int N) {
    Lds l;
    const size_t S = (size_t)(N + 1) * NZ;
    lds_double* q = base;
    auto take = [&](size_t n) { lds_double* r = q; q += n; return r; };
    l.pc = reinterpret_cast<const __attribute__((address_space(3))) OcpConst*>(take(NCONST));
    l.Mm = take(NZ * NZ); l.mm = take(NZ);
    l.fk = take(N + 1); l.filt = take(2 * OCP_FILTER); l.red = take(16); l.par = take(NPAR);
    l.sd = take((size_t)(N + 1) * (NSD > 0 ? NSD : 0) + 1);
    l.ext = take(NEXT);
    l.dirs = reinterpret_cast<__attribute__((address_space(3))) int*>(take((NDIR + 2) / 2));
    l.prepl = take(PREPL);
    dp w;
    const size_t V = (size_t)N * NX;
    if constexpr (VEC_LEVEL >= 1) {   // the vectors first, in LDS; everything else in the workspace
      l.Z = take(S); l.Zt = take(S); l.D = take(S); l.zL = take(S); l.zU = take(S); l.lbA = take(S); l.ubA = take(S);
    }
    if constexpr (VEC_LEVEL >= 2) { l.lam = take(V); l.c = take(V); l.ct = take(V); }
    if constexpr (VEC_LEVEL >= 3) l.grad = take(S);
    if constexpr (BIG) w = (gbl_double*)ws; else w = q;
    auto big = [&](size_t n) { dp r = w; w += n; return r; };
    if constexpr (VEC_LEVEL < 1) {   // (dp and the vector types coincide where these run)
      l.Z = big(S); l.Zt = big(S); l.D = big(S); l.zL = big(S); l.zU = big(S); l.lbA = big(S); l.ubA = big(S);
    }
    if constexpr (VEC_LEVEL < 2) { l.lam = big(V); l.c = big(V); l.ct = big(V); }
    if constexpr (VEC_LEVEL < 3) l.grad = big(S);
    l.lamn = big((size_t)N * NX);
    l.AB = big((size_t)N * NX * ABP); l.W = big((size_t)N * NZ * WP); l.Qd = big(qd_doubles(N));
    l.P = big((size_t)(N + 1) * NX * PP);
    l.Kg = big((size_t)N * NU * PP);
    l.sig = big(S); l.rbN = big(NX); l.Acl = big((size_t)N * NX * PP);
    const size_t R = (size_t)N * NC;
    l.cs = big(R); l.cst = big(R); l.cnu = big(R); l.cnun = big(R); l.cvL = big(R); l.cvU = big(R);
    l.cdvL = big(R); l.cdvU = big(R); l.cds = big(R); l.cd = big(R); l.csig = big(R); l.crb = big(R);
    l.Jd = big(R * NZ);
    l.c0 = big((size_t)N * NX); l.cd0 = big(R); l.cdt = big(R);
    if constexpr (COOP) {
      static_assert(!COOP || !BIG, "cooperative models keep the iterate in LDS");
      l.Xs = nullptr;
      l.gpc = (lds_double*)big(xs_doubles(N));
    } else {
      l.Xs = big(xs_doubles(N));
      l.gpc = nullptr;
    }
    l.prep = big((size_t)N * PREP);
    l.xc = big((size_t)N * XCW);
    return l;
  }

  // a slot (k, i) of the stage-major primal layout is a variable unless it is a pinned x_0 or the unused u_N
  __device__ static bool x0_pinned(const OcpConst& pc, int i) { return FIX_X0 && i < NX && !((pc.x0_free_mask >> i) & 1u); }
  __device__ static bool is_free(const OcpConst& pc, int k, int i) {
    if constexpr (NH > 0) {
      if (i >= NX && k >= pc.Nc) return false;   // beyond the control horizon the input is not a variable
    }
    return !((k == 0 && x0_pinned(pc, i)) || (k == pc.N && i >= NX));
  }
  // the same test with the problem constants it reads already in (scalar) registers: inside a slot loop the reads of `pc` sit
  // in conditional blocks (k == 0, ...) and would be waited for in the middle of the slot's arithmetic
  struct FreeTest {
    unsigned pinm;
    int N, Nc;
    __device__ bool operator()(int k, int i) const {
      if constexpr (NH > 0) {
        if (i >= NX && k >= Nc) return false;
      }
      return !((k == 0 && i < NX && ((pinm >> i) & 1u)) || (k == N && i >= NX));
    }
  };
  __device__ static FreeTest free_test(const OcpConst& pc) {
    FreeTest f;
    f.pinm = uni((int)(FIX_X0 ? (~pc.x0_free_mask) & ((1u << NX) - 1u) : 0u));
    f.N = uni(pc.N);
    f.Nc = NH > 0 ? uni(pc.Nc) : 0;
    return f;
  }
  // inequality row m exists at stage k
  __device__ static bool row_on(const OcpConst& pc, int k, int m) {
    // (a row without a finite bound constrains nothing: the copies at the node of rows that only exist at the collocation
    // points - bounds on algebraic states, hilo_nmpc_user.hip - are switched off this way)
    return (m < pc.nc || (k == pc.N - 1 && m < pc.nc + pc.nc_term)) && (pc.dlb[m] > -INFINITY || pc.dub[m] < INFINITY);
  }
  __device__ static double lb_of(const OcpConst& pc, int k, int i) {
    if (k == 0 && (pc.flags & 2) && i < NX0) return pc.x0lb[i];
    return (k > 0 && ((pc.k0_only_mask >> i) & 1u)) ? -INFINITY : pc.lbz[i];
  }
  __device__ static double ub_of(const OcpConst& pc, int k, int i) {
    if (k == 0 && (pc.flags & 2) && i < NX0) return pc.x0ub[i];
    return (k > 0 && ((pc.k0_only_mask >> i) & 1u)) ? INFINITY : pc.ubz[i];
  }

  // d >= n -> (i < j) among n slots.  A fixed trip count with a predicated body instead of `while (r >= n - 1 - i)`: the exit of
  // a lane-dependent loop is a join block, and the register allocator of ROCm 7.2 put live-range copies of i in front of that
  // block's EXEC restore (tools/check_exec_prologue.py found it in a run-time compiled estimator: lanes that skip the loop kept
  // a stale index)
  __device__ __forceinline__ static void pair_of(int d, int n, int& i, int& j) {
    int r = d - n, ii = 0;
    for (int t = 0; t + 1 < n; ++t) {
      const int w = n - 1 - ii;
      const bool go = r >= w;
      r -= go ? w : 0;
      ii += go ? 1 : 0;
    }
    i = ii;
    j = ii + 1 + r;
  }
  __device__ static int dir_of(int i, int j, int n) { return n + i * (n - 1) - i * (i - 1) / 2 + (j - i - 1); }
  __device__ static bool pair_on(const OcpConst& pc, int d) {   // d >= NZ: pair direction
    const int p = d - NZ;
    return p >= 256 || ((pc.pair_mask[p >> 6] >> (p & 63)) & 1ull) != 0ull;
  }
  // the swept directions of an interval (all NZ unit directions + the active pairs), once per solve
  __device__ static void build_dirs(const Lds l) {
    const OcpConst& pc = *(const OcpConst*)l.pc;
    if (threadIdx.x == 0) {
      int n = 0;
      for (int d = 0; d < NDIR; ++d)
        if (d < NZ || pair_on(pc, d)) l.dirs[1 + n++] = d;
      l.dirs[0] = n;
    }
    __syncthreads();
  }

  __device__ static const double* sd_of(const Lds l, int k) { return (const double*)(l.sd + (NSD > 0 ? k * NSD : 0)); }

  // ---- values only at a point Zp: defects cp_k = x_{k+1} - F_k(x_k,u_k), returns (f, theta = |c|_1) -------------
  // with inequality rows: theta also counts |d_k - sp_k| for the slacks sp; `dstore` (optional) receives d_k
  // `extra`: a per-lane value of the caller's that is summed over the lanes together with f and theta (the line search's
  // -sum log(slacks) of the trial point: one three-value reduction instead of a reduction of its own in front of this call)
  __device__ OCP_PHASE static FTheta eval_values_call(lds_double* lbase, double* ws, cvp Zp, ep cp, cdp sp, dp dstore, double extra) {
    lbase = uni(lbase); ws = uni(ws); Zp = uni(Zp); cp = uni(cp); sp = uni(sp); dstore = uni(dstore);
    const Lds l = carve(lbase, ws);
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N;
    double fpart = 0.0, tpart = 0.0;
    if constexpr (COOP) {
      // groups of gs lanes per interval: the same point in every lane of a group, the kernel sum split among them
      const int lane = threadIdx.x;
      int gs = 64 / N;
      gs = gs < 1 ? 1 : gs;
      const int ng = 64 / gs, g = lane / gs, gl = lane - g * gs;
      const double* gp = (const double*)pc.ext;
      for (int r = 0; r * ng < N; ++r) {
        const int k = r * ng + g;
        const bool active = g < ng && k < N;
        const int kk = active ? k : N - 1;
        int nev = 0;
        const GpExt ext{gp, l.ext, gs, active ? gl : 0, active ? g * gs : lane, !active, (lds_cdouble*)(l.ext + NEXT_SCR),
                        pc.nsub == 1 ? l.gpc + kk * 24 : nullptr, &nev, false};
        double x[NX], u[NU > 0 ? NU : 1], xn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = Zp[kk * NZ + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = Zp[kk * NZ + NX + i];
        PB::dyn(pc, (const double*)l.par, sd_of(l, kk), kk, x, u, xn, ext);
        if (active && gl == 0) {
          fpart += PB::stage_cost(pc, (const double*)l.par, sd_of(l, k), k, x, u);
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            const double ci = Zp[(k + 1) * NZ + i] - xn[i];
            cp[k * NX + i] = ci;
            tpart += fabs(ci);
          }
        }
      }
      if (lane == 63) {
        double x[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = Zp[N * NZ + i];
        fpart += PB::term_cost(pc, (const double*)l.par, sd_of(l, N), x);
      }
    } else {
    // (a trial point - anything but the iterate itself - is evaluated after the derivative phase: the iterate's collocation
    // states are in l.prep and start the Newton iteration)
    if constexpr (XCW > 0) PB::template coll_pass<false>(pc, (const double*)l.par, (const double*)l.sd, Zp, l.lam, l.cnu, N, l.xc, l.prep, l.prepl,
                                                         uni((size_t)Zp != (size_t)l.Z));
#ifdef HILO_DBG_TWICE_VALS   // developer knob (tools/dbg/c5dae_twice.sh): the same work again - the launch's extra time is this part's cost
    if constexpr (XCW > 0) PB::template coll_pass<false>(pc, (const double*)l.par, (const double*)l.sd, Zp, l.lam, l.cnu, N, l.xc, l.prep, l.prepl);
#endif
    OCP_FOR(k, (N) + 1) {
      double x[NX], u[NU > 0 ? NU : 1];
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] = Zp[k * NZ + i];
      if (k < N) {
        double xn[NX];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = Zp[k * NZ + NX + i];
        double dvf[NC > 0 ? NC : 1];
        if constexpr (XCW > 0) {
          fpart += PB::dyn_cost_xc(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, dvf, l.xc + (size_t)k * XCW);
        } else if constexpr (FUSED_CON) {
          static_assert(!FUSED_CON || FUSED, "fused rows need the fused cost");
          fpart += PB::dyn_cost_con(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, dvf, NoExt{});
        } else if constexpr (FUSED) {
          fpart += PB::dyn_cost(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, NoExt{});
        } else if constexpr (SYM) {
          // the derivative phase's own stage-point routine; the points are kept (l.Xs) for the derivative phase that follows
          // when this trial point is accepted
          using M = typename PB::Model;
          double xp[NX], up[NU > 0 ? NU : 1], X[4][NX], phi[NX];
#pragma unroll
          for (int i = 0; i < NX; ++i) xp[i] = x[i] * pc.sz[i];
#pragma unroll
          for (int i = 0; i < NU; ++i) up[i] = u[i] * pc.sz[NX + i];
          sym_points<M>(M::DISCRETE ? 1 : pc.order, pc.dt, xp, up, (const double*)l.par, X, phi);
          dp xs = l.Xs + (size_t)k * 4 * NX;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < NX; ++q) xs[i * NX + q] = X[i][q];
#pragma unroll
          for (int i = 0; i < NX; ++i) xn[i] = phi[i] * rcp_fast(pc.sz[i]);
          fpart += PB::stage_cost(pc, (const double*)l.par, sd_of(l, k), k, x, u);
        } else {
          if constexpr (SYM_MHE) {   // the arithmetic of the derivative phase (division = x * rcp_fast(y)): same defects in both
            FastD xf[NX], uf[NU > 0 ? NU : 1], xnf[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) xf[i] = FastD(x[i]);
#pragma unroll
            for (int i = 0; i < NU; ++i) uf[i] = FastD(u[i]);
            PB::dyn(pc, (const double*)l.par, sd_of(l, k), k, xf, uf, xnf, NoExt{});
#pragma unroll
            for (int i = 0; i < NX; ++i) xn[i] = xnf[i].v;
          } else {
            PB::dyn(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, NoExt{});
          }
          fpart += PB::stage_cost(pc, (const double*)l.par, sd_of(l, k), k, x, u);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const double ci = Zp[(k + 1) * NZ + i] - xn[i];
          cp[k * NX + i] = ci;
          tpart += fabs(ci);
        }
        if constexpr (NC > 0) {
          double dv[NC];
          if constexpr (FUSED_CON) {
#pragma unroll
            for (int m = 0; m < NC; ++m) dv[m] = dvf[m];
          } else PB::con(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, dv);
#pragma unroll
          for (int m = 0; m < NC; ++m) {
            if (row_on(pc, k, m)) {
              if (dstore) dstore[k * NC + m] = dv[m];
              if (sp) tpart += fabs(dv[m] - sp[k * NC + m]);
            }
          }
        }
      } else {
        fpart += PB::term_cost(pc, (const double*)l.par, sd_of(l, N), x);
      }
    }
    }
    if constexpr (OCP_TPB == 64) {
      double rv[3] = {fpart, tpart, extra};
      WaveReduceN<R_SUM, R_SUM, R_SUM>::run(rv);
      return FTheta{rv[0], rv[1], rv[2]};
    } else {
      const double fr = block_reduce<OpSum>(fpart, l.red);
      const double tr = block_reduce<OpSum>(tpart, l.red);
      return FTheta{fr, tr, block_reduce<OpSum>(extra, l.red)};
    }
  }

  __device__ __forceinline__ static FTheta eval_values(lds_double* lbase, double* ws, cvp Zp, ep cp, cdp sp = nullptr,
                                                      dp dstore = nullptr, double extra = 0.0) {
    const FTheta r = eval_values_call(lbase, ws, Zp, cp, sp, dstore, extra);
    return FTheta{uni(r.f), uni(r.theta), uni(r.x)};
  }

  // this lane's part of -sum log(slacks) over the slots of a point - TRIAL: of Zt = Z + alpha D, which is formed and stored on the
  // way.  Two slots per trip, reads first, and ONE logarithm for the (up to four) slacks of the two slots: the f64 logarithm is
  // a ~120-instruction dependent chain.  (Four slacks between 1e-75 and 1e75 cannot leave the range of a double.)
  template <bool TRIAL>
  __device__ __forceinline__ static double slots_log(const Lds l, cvp Zp, double alpha, int N) {
    constexpr int U = 2;
    const int SLT = (N + 1) * NZ;
    double part = 0.0;
    for (int base = 0; base < SLT; base += OCP_TPB * U) {
      double lbv[U], ubv[U], zv[U], dv[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e0 = base + u * OCP_TPB + (int)threadIdx.x;
        ok[u] = e0 < SLT;
        const int e = ok[u] ? e0 : 0;
        lbv[u] = l.lbA[e]; ubv[u] = l.ubA[e]; zv[u] = Zp[e];
        dv[u] = TRIAL ? l.D[e] : 0.0;
      }
      double prod = 1.0;
      bool pos = true;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double z = TRIAL ? zv[u] + alpha * dv[u] : zv[u];
        if constexpr (TRIAL) {
          if (ok[u]) l.Zt[base + u * OCP_TPB + (int)threadIdx.x] = z;
        }
        const bool hl = ok[u] && lbv[u] > -INFINITY, hu = ok[u] && ubv[u] < INFINITY;
        const double sl = hl ? z - lbv[u] : 1.0, su = hu ? ubv[u] - z : 1.0;
        prod *= sl * su;
        pos = pos && sl > 0.0 && su > 0.0;       // (two negative slacks must not cancel; a NaN slack fails the test as well)
      }
      const double r = -log(prod);
      part += pos ? r : __builtin_nan("");
    }
    return part;
  }
  // -sum log(slacks) of a point (the barrier function is mu times this; the sum itself does not depend on mu, so the value of
  // an accepted trial point is carried into the next iteration instead of being recomputed)
  __device__ static double barrier_logs(const Lds l, cvp Zp, cdp sp = nullptr) {
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N;
    double part = slots_log<false>(l, Zp, 0.0, N);
    if constexpr (NC > 0) {
      OCP_FOR(e, N * NC) {
        const int m = e % NC;
        if (!row_on(pc, e / NC, m)) continue;
        if (pc.dlb[m] > -INFINITY) part -= log(sp[e] - pc.dlb[m]);
        if (pc.dub[m] < INFINITY) part -= log(pc.dub[m] - sp[e]);
      }
    }
    return uni(block_reduce<OpSum>(part, l.red));
  }
  // trial point Zt = Z + alpha D (and the slack rows), formed in the same pass as its -sum log(slacks)
  // (returns this lane's part of the sum: reduced by the evaluation of the point that follows - eval_values' `extra`)
  __device__ static double form_trial_part(const Lds l, double alpha) {
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N;
    double part = slots_log<true>(l, l.Z, alpha, N);
    if constexpr (NC > 0) {
      OCP_FOR(e, N * NC) {
        const int m = e % NC;
        const double sv = l.cs[e] + alpha * l.cds[e];
        l.cst[e] = sv;
        if (!row_on(pc, e / NC, m)) continue;
        if (pc.dlb[m] > -INFINITY) part -= log(sv - pc.dlb[m]);
        if (pc.dub[m] < INFINITY) part -= log(pc.dub[m] - sv);
      }
    }
    __syncthreads();
    return part;
  }

  // ---- full derivative evaluation at Z: c, AB, grad, per-stage cost values, Lagrangian Hessian blocks --------
  // inlined at its call sites: as a real call its ~170 live registers cost 66 callee-saved VGPR saves per call (17 KB of
  // scratch per wave and call, 120 MB of HBM writes per 1024-instance launch)
  __device__ __attribute__((always_inline)) static double eval_derivs_body(lds_double* lbase, double* ws, bool reuse = false) {
    const Lds l = carve(lbase, ws);
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N;
    const unsigned pinm = FIX_X0 ? (~pc.x0_free_mask) & ((1u << NX) - 1u) : 0u;  // pinned slots of x_0 (bit i)
    constexpr int GPW = COOP ? 64 / NDIR : 1;  // cooperative: lane groups of NDIR directions, GPW intervals per round
    const int nact = COOP ? NDIR : uni(l.dirs[0]);   // swept directions per interval (pair_mask)
    // PREP policies: ONE interval per pass of the wave (its prepared block staged in LDS), PASSES passes per interval
    const int PASSES = (nact + OCP_TPB - 1) / OCP_TPB;
    const int ntask = COOP ? ((N + GPW - 1) / GPW) * 64 + NXDIR : (PREP > 0 ? N * PASSES * OCP_TPB + NXDIR : N * nact + NXDIR);
    const int tbase = ntask - NXDIR;
    if constexpr (XCW > 0) {    // cooperative collocation: states, tangents and the adjoint weights of every interval
      PB::template coll_pass<true>(pc, (const double*)l.par, (const double*)l.sd, l.Z, l.lam, l.cnu, N, l.xc, l.prep, l.prepl);
#ifdef HILO_DBG_TWICE_COLL
      PB::template coll_pass<true>(pc, (const double*)l.par, (const double*)l.sd, l.Z, l.lam, l.cnu, N, l.xc, l.prep, l.prepl);
#endif
    } else if constexpr (PREP > 0) {   // once per interval: what all its directions share (PB::prepare)
      OCP_FOR(k, N) {
        double x[NX], u[NU > 0 ? NU : 1];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = l.Z[k * NZ + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = l.Z[k * NZ + NX + i];
        PB::prepare(pc, (const double*)l.par, sd_of(l, k), k, x, u, l.prep + (size_t)k * PREP);
      }
      __syncthreads();
    }
    // Cooperative collocation: the directions of the horizon are PACKED into the passes of the wave - pass p sweeps the tasks
    // [xbase, xend) of the list (interval k, swept direction s) -> k nact + s, which may belong to up to PREPB intervals; their
    // blocks are staged side by side.  (One interval per pass left 28 of 64 lanes idle for configuration 5's 36 directions.)
    // xbase == N nact: the pass of the terminal cost's directions.
    const int xtotal = N * nact;
#ifdef HILO_DBG_TWICE_DIRS
    for (int dbg_rep = 0; dbg_rep < 2; ++dbg_rep) {
#endif
    int xbase = 0, xend = 0, xklo = 0;
    for (int task0_base = 0; XCW > 0 ? xbase <= xtotal : task0_base < ntask; task0_base += OCP_TPB, xbase = XCW > 0 ? (xbase < xtotal ? xend : xtotal + 1) : 0) {
      int xtask0 = 0, xk = 0, xslot = 0;
      if constexpr (XCW > 0) {
        if (xbase < xtotal) {
          xklo = xbase / nact;
          const int kend = xklo + PREPB < N ? xklo + PREPB : N;
          xend = xbase + OCP_TPB < kend * nact ? xbase + OCP_TPB : kend * nact;
          const int nblk = (xend - 1) / nact - xklo + 1;
          __syncthreads();
          for (int q = (int)threadIdx.x; q < nblk * PREP; q += OCP_TPB) l.prepl[q] = l.prep[(size_t)xklo * PREP + q];
          __syncthreads();
          const int my = xbase + (int)threadIdx.x;
          xk = my / nact;
          xslot = my - xk * nact;
          xtask0 = my < xend ? 0 : ntask;                 // (any value below tbase: an interval's task; ntask: no task in this pass)
        } else xtask0 = tbase + (int)threadIdx.x;          // the terminal cost's directions
      } else if constexpr (PREP > 0) {
        if (task0_base < tbase && (task0_base / OCP_TPB) % PASSES == 0) {   // a new interval: stage its block
          const int kk = task0_base / (OCP_TPB * PASSES);
          __syncthreads();
          for (int q = (int)threadIdx.x; q < PREP; q += OCP_TPB) l.prepl[q] = l.prep[(size_t)kk * PREP + q];
          __syncthreads();
        }
      }
      if (const int task0 = XCW > 0 ? xtask0 : task0_base + (int)threadIdx.x; task0 < ntask) {
      if (task0 < tbase) {
        int k, d, task = task0;
        bool active = true;
        if constexpr (COOP) {
          const int lane = threadIdx.x, g = lane / NDIR;
          d = lane - g * NDIR;
          k = (task0 / 64) * GPW + g;
          active = g < GPW && k < N;
          if (!active) { k = N - 1; d = 0; }
          task = k * NDIR + d;
        } else if constexpr (XCW > 0) {
          k = xk;
          d = l.dirs[1 + xslot];
          task = k * NDIR + d;
        } else if constexpr (PREP > 0) {
          k = task0 / (OCP_TPB * PASSES);
          const int slot = task0 - k * OCP_TPB * PASSES;
          if (slot >= nact) continue;
          d = l.dirs[1 + slot];
          task = k * NDIR + d;
        } else {
          k = task0 / nact;
          d = l.dirs[1 + (task0 - k * nact)];
          task = k * NDIR + d;
        }
        int di = d, dj = -1;
        if (d >= NZ) pair_of(d, NZ, di, dj);
        const bool dead = k == 0 && ((((pinm >> di) | (dj >= 0 ? pinm >> dj : 0u)) & 1u) != 0u);  // touches the pinned x_0
        if (dead && active) {
          l.Qd[task] = 0.0;
          if (d < NZ) {
            l.grad[d] = 0.0;
#pragma unroll
            for (int m = 0; m < NX; ++m) l.AB[m * ABP + d] = 0.0;
            if constexpr (NC > 0) {
#pragma unroll
              for (int m = 0; m < NC; ++m) l.Jd[m * NZ + d] = 0.0;
            }
          }
          if (d != 0 && !COOP) continue;
        }
        Jet2 x[NX], u[NU > 0 ? NU : 1], xn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = Jet2(l.Z[k * NZ + i], (!dead && (i == di || i == dj)) ? 1.0 : 0.0, 0.0);
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = Jet2(l.Z[k * NZ + NX + i], (NX + i == di || NX + i == dj) ? 1.0 : 0.0, 0.0);
        // what the task needs from the rest of the iterate is requested here, in front of the sweep through the shooting map
        // (in workspace mode the multipliers come from global memory: the round trip hides behind the sweep)
        double zn[NX], lamv[NX], cnuv[NC > 0 ? NC : 1];
#pragma unroll
        for (int m = 0; m < NX; ++m) {
          zn[m] = l.Z[(k + 1) * NZ + m];
          lamv[m] = l.lam[k * NX + m];
        }
        if constexpr (NC > 0) {
#pragma unroll
          for (int m = 0; m < NC; ++m) cnuv[m] = l.cnu[k * NC + m];
        }
        if constexpr (COOP) {
          const int lane = threadIdx.x, g = lane / NDIR;
          int nev = 0;
          const GpExt ext{(const double*)pc.ext, l.ext, NDIR, active ? d : 0, active ? g * NDIR : lane, !active,
                          (lds_cdouble*)(l.ext + NEXT_SCR), l.gpc + k * 24, &nev, reuse && pc.nsub == 1};
          PB::dyn(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, ext);
          if (!active) continue;
        } else {
          if constexpr (!FUSED) PB::dyn(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, NoExt{});
        }
        // a policy with a purely quadratic stage cost supplies value / gradient / (constant) Hessian in closed form
        Jet2 lc(0.0);
        Jet2 dvf[NC > 0 ? NC : 1];
        if constexpr (XCW > 0) {
          lc = PB::dyn_cost_prep(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, dvf, l.prepl + (k - xklo) * PREP);
        } else if constexpr (PREP > 0) {
          lc = PB::dyn_cost_prep(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, dvf, l.prepl);
        } else if constexpr (FUSED_CON) {
          lc = PB::dyn_cost_con(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, dvf, NoExt{});
        } else if constexpr (FUSED) {
          static_assert(!FUSED || (!COOP && !PB::QUAD_COST), "fused cost: Taylor evaluation, no cooperative model");
          lc = PB::dyn_cost(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, NoExt{});
        } else if constexpr (!PB::QUAD_COST) lc = PB::stage_cost(pc, (const double*)l.par, sd_of(l, k), k, x, u);
        double q = lc.b;
        {  // the two store groups each under ONE condition
#pragma unroll
          for (int m = 0; m < NX; ++m) q -= lamv[m] * xn[m].b;
          if (d == 0) {
#pragma unroll
            for (int m = 0; m < NX; ++m) l.c[k * NX + m] = zn[m] - xn[m].v;
          }
          if (d < NZ && !dead) {
#pragma unroll
            for (int m = 0; m < NX; ++m) l.AB[(k * NX + m) * ABP + d] = xn[m].a;
          }
        }
        if constexpr (!PB::QUAD_COST) {
          if (d == 0) l.fk[k] = lc.v;
        }
        if constexpr (NC > 0) {  // inequality rows: value, Jacobian column, nu-weighted second-order term
          Jet2 dv[NC];
          if constexpr (FUSED_CON) {
#pragma unroll
            for (int m = 0; m < NC; ++m) dv[m] = dvf[m];
          } else PB::con(pc, (const double*)l.par, sd_of(l, k), k, x, u, xn, dv);
#pragma unroll
          for (int m = 0; m < NC; ++m) {
            if (row_on(pc, k, m)) {
              if (d == 0) l.cd[k * NC + m] = dv[m].v;
              if (d < NZ && !dead) l.Jd[(k * NC + m) * NZ + d] = dv[m].a;
              q += cnuv[m] * dv[m].b;
            }
          }
        }
        if (!dead) {
          l.Qd[task] = q;
          if constexpr (!PB::QUAD_COST) {
            if (d < NZ) l.grad[k * NZ + d] = lc.a;
          }
        }
      } else if (task0 < ntask) {  // terminal cost V(x_N): directions over the NX state slots
        const int d = task0 - tbase;
        int di = d, dj = -1;
        if (d >= NX) pair_of(d, NX, di, dj);
        Jet2 x[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = Jet2(l.Z[N * NZ + i], (i == di || i == dj) ? 1.0 : 0.0, 0.0);
        const Jet2 v = PB::term_cost(pc, (const double*)l.par, sd_of(l, N), x);
        qd_term(l, N)[d] = v.b;
        if (d < NX) l.grad[N * NZ + d] = v.a;
        if (d == 0) l.fk[N] = v.v;
      }
      }
    }
#ifdef HILO_DBG_TWICE_DIRS
    }
#endif
    __syncthreads();
    // Hessian blocks by polarisation: H_ii = q(e_i), H_ij = (q(e_i+e_j) - q(e_i) - q(e_j)) / 2
    if constexpr (BIG) {
      // workspace mode: four entries per lane and trip with their twelve table reads requested together (a trip of the plain
      // loop is one global-memory round trip: 95 of them for N = 50, n_z = 11); a direction that was not swept is read
      // (whatever the table holds there) and dropped by the select
      constexpr int U = 4;
      const int total = N * NZ * NZ;
      for (int base = 0; base < total; base += OCP_TPB * U) {
        double qd[U], qa[U], qb[U];
        int kk[U], ii[U], jj[U];
        bool on[U], ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e0 = base + u * OCP_TPB + (int)threadIdx.x;
          ok[u] = e0 < total;
          const int e = ok[u] ? e0 : 0;
          const int k = e / (NZ * NZ), r = e - k * NZ * NZ, i = r / NZ, j = r - i * NZ;
          const int a = i < j ? i : j, b = i < j ? j : i;
          const int dd = i == j ? i : dir_of(a, b, NZ);
          cdp Q = l.Qd + k * NDIR;
          qd[u] = Q[dd]; qa[u] = Q[a]; qb[u] = Q[b];
          kk[u] = k; ii[u] = i; jj[u] = j;
          on[u] = i == j || COOP || pair_on(pc, dd);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = kk[u], i = ii[u], j = jj[u];
          double h = i == j ? qd[u] : (on[u] ? 0.5 * (qd[u] - qa[u] - qb[u]) : 0.0);
          if constexpr (PB::QUAD_COST) h += PB::cost_hess(pc, k, i, j);
          if (k == 0 && (((pinm >> i) | (pinm >> j)) & 1u)) h = 0.0;
          if (ok[u]) l.W[(k * NZ + i) * WP + j] = h;
        }
      }
    } else {
      OCP_FOR(e, N * NZ * NZ) {
        const int k = e / (NZ * NZ), r = e - k * NZ * NZ, i = r / NZ, j = r - i * NZ;
        cdp Q = l.Qd + k * NDIR;
        double h;
        if (i == j) h = Q[i];
        else {
          const int a = i < j ? i : j, b = i < j ? j : i, dd = dir_of(a, b, NZ);
          h = (COOP || pair_on(pc, dd)) ? 0.5 * (Q[dd] - Q[a] - Q[b]) : 0.0;
        }
        if constexpr (PB::QUAD_COST) h += PB::cost_hess(pc, k, i, j);
        if (k == 0 && (((pinm >> i) | (pinm >> j)) & 1u)) h = 0.0;
        l.W[(k * NZ + i) * WP + j] = h;
      }
    }
    if constexpr (PB::QUAD_COST) {
      OCP_FOR(e, N * NZ) {
        const int k = e / NZ, i = e - k * NZ;
        if (is_free(pc, k, i)) l.grad[e] = PB::cost_grad(pc, (const double*)l.par, sd_of(l, k), k, i, (const double*)(l.Z + k * NZ));
      }
      OCP_FOR(k, N) {
        double x[NX], u[NU > 0 ? NU : 1];
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = l.Z[k * NZ + i];
#pragma unroll
        for (int i = 0; i < NU; ++i) u[i] = l.Z[k * NZ + NX + i];
        l.fk[k] = PB::stage_cost(pc, (const double*)l.par, sd_of(l, k), k, x, u);
      }
      __syncthreads();
    }
    double fpart = 0.0;
    OCP_FOR(k, (N) + 1) fpart += l.fk[k];
    OCP_FOR(a, NU) l.grad[N * NZ + NX + a] = 0.0;
    const double f = block_reduce<OpSum>(fpart, l.red);
    __syncthreads();
    return f;
  }


  // ---- derivative evaluation from symbolic model derivatives (SYM policies: quadratic cost, no inequality rows) ----------
  // What CasADi hands IPOPT in the reference - the Hessian of the Lagrangian of the discretised model from a symbolic graph
  // with shared sub-expressions (mpc.py:1778-1787) - instead of NZ (NZ + 1) / 2 second-order Taylor sweeps per interval:
  //   stages     X_i = x + h sum_j a_ij k_j,  k_i = f(X_i, u);         Phi = x + h sum_i b_i k_i
  //   adjoint    kb_i = h b_i lam' + h sum_{j > i} a_ji f_x(X_j)^T kb_j            (lam' = lam / s_x: scaled variables)
  //   tangent    dX_i = [I 0] + h sum_j a_ij dk_j,  dk_i = J_i dW_i,  dW_i = [dX_i; 0 I],  J_i = f_w(X_i, u)
  //   Hessian    lam'^T Phi_ww = sum_i dW_i^T H_i dW_i,  H_i = sum_m kb_i[m] d2 f_m / dw2 (X_i, u)   (second-order adjoint: the
  //              stage points are affine in the slopes, f is the only nonlinearity)
  // J_i and H_i come from ModelSym<M>::jh (generated, ~150 operations for the chemostat against ~240 per Taylor sweep and
  // RK stage, 21 sweeps).  A lane owns CPL columns (directions) of one interval; the lanes of an interval exchange the tangent
  // columns of the current stage through the interval's block of l.W, which receives the Hessian at the end.
  // SYM policies: the Hessian of the terminal cost is constant - written once per solve in the direction form the Riccati
  // start reads (Q[e_i] = H_ii, Q[e_i + e_j] = H_ii + H_jj + 2 H_ij)
  // Runge-Kutta stage points X_i and the integrated state Phi of one interval (SYM policies: the tableau of orders 1..4 written
  // out, model divisions as x * rcp_fast(y)): ONE routine for the values-only pass of the line search and for the derivative
  // phase, so that a trial point the line search accepted leaves exactly the stage points the derivative phase needs (l.Xs)
  template <class M>
  __device__ __forceinline__ static void sym_points(int order, double h, const double* x, const double* u, const double* par,
                                                    double (*X)[NX], double* phi) {
    constexpr bool DISC = M::DISCRETE;
    const double a10 = order >= 2 ? 0.5 : 0.0, a20 = order == 3 ? -1.0 : 0.0, a21 = order == 3 ? 2.0 : (order == 4 ? 0.5 : 0.0),
                 a32 = order == 4 ? 1.0 : 0.0;
    const double hb[4] = {DISC ? 1.0 : h * erk_b<0>(order), DISC ? 0.0 : h * erk_b<1>(order), DISC ? 0.0 : h * erk_b<2>(order),
                          DISC ? 0.0 : h * erk_b<3>(order)};
    FastD kk[4][NX], uf[NU > 0 ? NU : 1];
#pragma unroll
    for (int i = 0; i < NU; ++i) uf[i] = FastD(u[i]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      FastD Xf[NX];
#pragma unroll
      for (int s = 0; s < NX; ++s) {
        double acc = x[s];
        if (i == 1) acc += (h * a10) * kk[0][s].v;
        if (i == 2) acc += (h * a20) * kk[0][s].v + (h * a21) * kk[1][s].v;
        if (i == 3) acc += (h * a32) * kk[2][s].v;
        X[i][s] = acc;
        Xf[s] = FastD(acc);
      }
      if (i < order) {
        M::ode(Xf, uf, par, h, kk[i]);
      } else {
#pragma unroll
        for (int s = 0; s < NX; ++s) kk[i][s] = FastD(0.0);
      }
    }
#pragma unroll
    for (int s = 0; s < NX; ++s)
      phi[s] = (DISC ? 0.0 : x[s]) + hb[0] * kk[0][s].v + hb[1] * kk[1][s].v + hb[2] * kk[2][s].v + hb[3] * kk[3][s].v;
  }

  __device__ static void term_hess_dirs(const Lds l) {
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N;
    OCP_FOR(d, NXDIR) {
      int di = d, dj = -1;
      if (d >= NX) pair_of(d, NX, di, dj);
      double q = PB::term_hess(pc, di, di);
      if (dj >= 0) q += PB::term_hess(pc, dj, dj) + 2.0 * PB::term_hess(pc, di, dj);
      qd_term(l, N)[d] = q;
    }
  }

  // `reuse`: the iterate is the trial point the line search just evaluated (eval_values): its stage points are in l.Xs, its
  // defects were copied to l.c and its objective is known - the Runge-Kutta slopes and the cost value are not recomputed
  __device__ __attribute__((always_inline)) static double eval_derivs_sym(lds_double* lbase, double* ws, bool reuse = false) {
    using M = typename PB::Model;
    using MS = ModelSym<M>;
    using SM = sym_masks<MS>;
    static_assert(PB::QUAD_COST && NC == 0 && !COOP && !FUSED && M::NX == NX && M::NU == NU, "SYM: plain tracking policies");
    constexpr bool DISC = M::DISCRETE;
    constexpr int CPL = 2, LPI = (NZ + CPL - 1) / CPL;
    const Lds l = carve(lbase, ws);
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N;
    const int order = DISC ? 1 : pc.order;
    const double h = pc.dt;
    const unsigned pinm = FIX_X0 ? (~pc.x0_free_mask) & ((1u << NX) - 1u) : 0u;
    // tableau entries that can be non-zero for orders 1..4 (modeling.py:1239-1250): a10, a20, a21, a32
    const double a10 = order >= 2 ? 0.5 : 0.0, a20 = order == 3 ? -1.0 : 0.0, a21 = order == 3 ? 2.0 : (order == 4 ? 0.5 : 0.0),
                 a32 = order == 4 ? 1.0 : 0.0;
    const double hb[4] = {DISC ? 1.0 : h * erk_b<0>(order), DISC ? 0.0 : h * erk_b<1>(order), DISC ? 0.0 : h * erk_b<2>(order),
                          DISC ? 0.0 : h * erk_b<3>(order)};
    const double* par = (const double*)l.par;
    DTICK0
    double fpart = 0.0;
    OCP_FOR(a, NU) l.grad[N * NZ + NX + a] = 0.0;
    DTICK(7)
    constexpr int IPP = OCP_TPB / LPI;             // intervals per pass: the LPI lanes of an interval work in the same pass
    static_assert(IPP >= 1, "SYM: more column groups than lanes");
    for (int kbase = 0; kbase < N; kbase += IPP) {
      const int li = (int)threadIdx.x / LPI;
      const bool act = li < IPP && kbase + li < N;
      const int k = act ? kbase + li : N - 1, g = act ? (int)threadIdx.x - li * LPI : 0, c0 = g * CPL;
      double zs[NZ], sz[NZ], isz[NX], x[NX], u[NU > 0 ? NU : 1], lamp[NX], X[4][NX], kb[4][NX];
#pragma unroll
      for (int i = 0; i < NZ; ++i) {
        zs[i] = l.Z[k * NZ + i];
        sz[i] = pc.sz[i];
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        isz[i] = rcp_fast(sz[i]);
        x[i] = zs[i] * sz[i];
        lamp[i] = l.lam[k * NX + i] * isz[i];
      }
#pragma unroll
      for (int i = 0; i < NU; ++i) u[i] = zs[NX + i] * sz[NX + i];
      DTICK(4)
      // stage cost: value (one lane of the interval) and the gradient rows of this lane's columns, closed form; the lanes of the
      // last interval add the terminal cost V(x_N) (its constant Hessian is written once per solve: term_hess_dirs)
      double ch[NZ][CPL];
      {
        double xN[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) xN[i] = l.Z[N * NZ + i];
        if constexpr (pb_symtab<PB>::value) {
          double gc[CPL];
          PB::template cost_cols<CPL>(pc, par, k, c0, zs, gc, ch);
          if (act) {
            if (g == 0 && !reuse) fpart += PB::stage_cost(pc, par, sd_of(l, k), k, zs, zs + NX);
            if (g == 0 && k == N - 1 && !reuse) fpart += PB::term_cost(pc, par, sd_of(l, N), xN);
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
              const int col = c0 + c;
              if (col < NZ) l.grad[k * NZ + col] = is_free(pc, k, col) ? gc[c] : 0.0;
              if (col < NX && k == N - 1) l.grad[N * NZ + col] = PB::term_grad(pc, col, xN);
            }
          }
        } else {
        if (act) {
          if (g == 0 && !reuse) fpart += PB::stage_cost(pc, par, sd_of(l, k), k, zs, zs + NX);
          if (g == 0 && k == N - 1 && !reuse) fpart += PB::term_cost(pc, par, sd_of(l, N), xN);
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const int col = c0 + c;
            if (col < NZ) l.grad[k * NZ + col] = is_free(pc, k, col) ? PB::cost_grad(pc, par, sd_of(l, k), k, col, zs) : 0.0;
            if (col < NX && k == N - 1) l.grad[N * NZ + col] = PB::term_grad(pc, col, xN);
          }
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
          for (int r = 0; r < NZ; ++r) ch[r][c] = (c0 + c < NZ && r >= c0 + c) ? PB::cost_hess(pc, k, r, c0 + c) : 0.0;
        }
      }
      DTICK(5)
      if (!reuse) {  // stage points, slopes, Phi, defect
        double phi[NX];
        sym_points<M>(order, h, x, u, par, X, phi);
        if (act && g == 0) {
#pragma unroll
          for (int s = 0; s < NX; ++s) l.c[k * NX + s] = l.Z[(k + 1) * NZ + s] - phi[s] * isz[s];
        }
      } else {
        cdp xs = l.Xs + (size_t)k * 4 * NX;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int s = 0; s < NX; ++s) X[i][s] = xs[i * NX + s];
      }
      DTICK(0)
      // adjoint weights of the slopes (reverse over the stages)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < NX; ++s) kb[i][s] = hb[i] * lamp[s];
#pragma unroll
      for (int j = 3; j >= 1; --j) {
        if (j < order) {
          double fx[NX * NX], t[NX];
          MS::jx(X[j], u, par, fx);
#pragma unroll
          for (int n = 0; n < NX; ++n) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < NX; ++m)
              if ((SM::X >> (m * NX + n)) & 1ull) acc += fx[m * NX + n] * kb[j][m];
            t[n] = acc;
          }
#pragma unroll
          for (int n = 0; n < NX; ++n) {
            if (j == 3) kb[2][n] += (h * a32) * t[n];
            if (j == 2) { kb[1][n] += (h * a21) * t[n]; kb[0][n] += (h * a20) * t[n]; }
            if (j == 1) kb[0][n] += (h * a10) * t[n];
          }
        }
      }
      DTICK(1)
      // tangent columns and the second-order adjoint, stage by stage
      dp scr = l.W + (size_t)k * NZ * WP;            // [NX][NZ] tangent block dX_i of the interval (exchange between its lanes)
      double dXc[NX][CPL], dX2[NX][CPL], dPhi[NX][CPL], G[NZ][CPL];
      // the input rows of this lane's tangent columns dW = d(x_i, u)/dz[:, col] are constant: 1 where the column IS that input -
      // as factors (exact: products with 0 and 1) instead of a compare-and-select per term of the products below
      double eU[NU > 0 ? NU : 1][CPL];
#pragma unroll
      for (int a = 0; a < NU; ++a)
#pragma unroll
        for (int c = 0; c < CPL; ++c) eU[a][c] = (c0 + c == NX + a) ? 1.0 : 0.0;
#pragma unroll
      for (int s = 0; s < NX; ++s)
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          dXc[s][c] = (c0 + c == s) ? 1.0 : 0.0;
          dX2[s][c] = dXc[s][c];
          dPhi[s][c] = DISC ? 0.0 : dXc[s][c];
        }
#pragma unroll
      for (int r = 0; r < NZ; ++r)
#pragma unroll
        for (int c = 0; c < CPL; ++c) G[r][c] = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < order) {
          if (act) {
#pragma unroll
            for (int s = 0; s < NX; ++s)
#pragma unroll
              for (int c = 0; c < CPL; ++c)
                if (c0 + c < NZ) scr[s * NZ + c0 + c] = dXc[s][c];
          }
          __syncthreads();
          double dK[NX][CPL], v[NZ][CPL];
          {
            double J[NX * NZ], H[NZ * (NZ + 1) / 2];
            MS::jh(X[i], u, par, kb[i], J, H);
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
#pragma unroll
              for (int m = 0; m < NX; ++m) {
                double acc = 0.0;
#pragma unroll
                for (int n = 0; n < NZ; ++n)
                  if ((SM::J >> (m * NZ + n)) & 1ull) acc += J[m * NZ + n] * (n < NX ? dXc[n < NX ? n : 0][c] : eU[n >= NX ? n - NX : 0][c]);
                dK[m][c] = acc;
              }
#pragma unroll
              for (int a = 0; a < NZ; ++a) {   // v = H_i dW_i[:, col]
                double acc = 0.0;
#pragma unroll
                for (int n = 0; n < NZ; ++n) {
                  const int q = a >= n ? a * (a + 1) / 2 + n : n * (n + 1) / 2 + a;
                  if ((SM::H >> q) & 1ull) acc += H[q] * (n < NX ? dXc[n < NX ? n : 0][c] : eU[n >= NX ? n - NX : 0][c]);
                }
                v[a][c] = acc;
              }
            }
          }
          // G[r][col] += dW_i[:, r]^T v = sum_{a < NX} dX_i[a][r] v[a] + (r >= NX ? v[r] : 0)
#pragma unroll
          for (int r = 0; r < NZ; ++r) {
            double col_r[NX];
#pragma unroll
            for (int a = 0; a < NX; ++a) col_r[a] = scr[a * NZ + r];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
              double acc = r >= NX ? v[r][c] : 0.0;
#pragma unroll
              for (int a = 0; a < NX; ++a) acc += col_r[a] * v[a][c];
              G[r][c] += acc;
            }
          }
#pragma unroll
          for (int s = 0; s < NX; ++s)
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
              const double e = (c0 + c == s) ? 1.0 : 0.0;
              dPhi[s][c] += hb[i] * dK[s][c];
              if (i == 0) { dXc[s][c] = e + (h * a10) * dK[s][c]; dX2[s][c] = e + (h * a20) * dK[s][c]; }
              if (i == 1) dXc[s][c] = dX2[s][c] + (h * a21) * dK[s][c];
              if (i == 2) dXc[s][c] = e + (h * a32) * dK[s][c];
            }
          __syncthreads();   // every lane of the interval has read the block before the next stage overwrites it
        }
      }
      DTICK(2)
      if (act) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int col = c0 + c;
          if (col < NZ) {
            const bool pin_c = k == 0 && ((pinm >> col) & 1u);
            double sc = sz[0];
#pragma unroll
            for (int r = 1; r < NZ; ++r) sc = col == r ? sz[r] : sc;
#pragma unroll
            for (int m = 0; m < NX; ++m) l.AB[(k * NX + m) * ABP + col] = pin_c ? 0.0 : dPhi[m][c] * sc * isz[m];
            // the interval's block of W: entries (r, col) and (col, r) for r >= col from ONE value (exact symmetry)
#pragma unroll
            for (int r = 0; r < NZ; ++r) {
              if (r >= col) {
                const bool pin = pin_c || (k == 0 && ((pinm >> r) & 1u));
                const double hv = pin ? 0.0 : ch[r][c] - sz[r] * sc * G[r][c];
                l.W[((size_t)k * NZ + r) * WP + col] = hv;
                l.W[((size_t)k * NZ + col) * WP + r] = hv;
              }
            }
          }
        }
      }
      DTICK(3)
    }
    __syncthreads();
    const double f = block_reduce<OpSum>(fpart, l.red);
    __syncthreads();
    DTICK(6)
    return f;
  }

  // ---- the same derivative evaluation for the moving-horizon estimator (policy MheNoise, csrc/hilo_mhe.hip) ----------------
  // Engine variables of stage k: z = (x_k, w_k) with x_{k+1} = Phi(x_k; u_meas_k, p) / s_x + w_k.  Only x enters nonlinearly:
  // tangent columns for the NX state directions (NX / 2 lanes per interval: N = 30 is one pass), B_k = I, and a Hessian with the
  // blocks  W_xx = l_xx - lam'^T Phi_xx,  W_ww = l_ww,  W_xw = 0.  Cost of the reference (mhe.py:742-748): arrival
  // (x_0 - x_a)^T Wx (.) at k = 0; for k >= 1  r^T Wy r + w^T Ww w  with r = h(x_k) - y_k on UN-scaled quantities - value,
  // gradient and Hessian from the symbolic measurement derivatives ModelSym<M>::mjh (Jy, and Hy contracted with 2 Wy r).
  __device__ __attribute__((always_inline)) static double eval_derivs_sym_mhe(lds_double* lbase, double* ws) {
    using M = typename PB::Model;
    using MS = ModelSym<M>;
    constexpr int MU = PB::MU, NY = PB::NY, MZ = NX + MU, MU1 = MU > 0 ? MU : 1, NP = M::NP;
    static_assert(NU == NX && NC == 0 && !COOP && !FIX_X0 && NH == 0 && !M::DISCRETE && NX % 2 == 0, "SYM_MHE: the MheNoise policy");
    constexpr int CPL = 2, LPI = NX / CPL;
    const Lds l = carve(lbase, ws);
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N, order = pc.order;
    const double h = pc.dt;
    const double a10 = order >= 2 ? 0.5 : 0.0, a20 = order == 3 ? -1.0 : 0.0, a21 = order == 3 ? 2.0 : (order == 4 ? 0.5 : 0.0),
                 a32 = order == 4 ? 1.0 : 0.0;
    const double hb[4] = {h * erk_b<0>(order), h * erk_b<1>(order), h * erk_b<2>(order), h * erk_b<3>(order)};
    const double* par = (const double*)l.par;
    double fpart = 0.0;
    OCP_FOR(d, NXDIR) qd_term(l, N)[d] = 0.0;          // no terminal cost (mhe.py: the window ends with the last measurement)
    OCP_FOR(a, NZ) l.grad[N * NZ + a] = 0.0;
    constexpr int IPP = OCP_TPB / LPI;
    for (int kbase = 0; kbase < N; kbase += IPP) {
      const int li = (int)threadIdx.x / LPI;
      const bool act = li < IPP && kbase + li < N;
      const int k = act ? kbase + li : N - 1, g = act ? (int)threadIdx.x - li * LPI : 0, c0 = g * CPL;
      const double* sd = sd_of(l, k);
      double xsv[NX], wv[NX], sz[NX], sw[NX], isz[NX], x[NX], ue[MU1], lamp[NX], X[4][NX], kb[4][NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        xsv[i] = l.Z[k * NZ + i];
        wv[i] = l.Z[k * NZ + NX + i];
        sz[i] = pc.sz[i];
        sw[i] = pc.sz[NX + i];
        isz[i] = rcp_fast(sz[i]);
        x[i] = xsv[i] * sz[i];
        lamp[i] = l.lam[k * NX + i] * isz[i];
      }
#pragma unroll
      for (int i = 0; i < MU; ++i) ue[i] = sd[i] * pc.cost[PB::O_SU + i];
      // ---- cost of the stage: value (one lane), gradient entries and Hessian columns of this lane ----
      double chx[NX][CPL], chw[NX][CPL], gx[CPL], gw[CPL];
      {
        double Jy[NY * NX], Hy[NX * (NX + 1) / 2], kby[NY], r[NY], yv[NY], ws_[NX];
        double val = 0.0;
        if (k == 0) {   // arrival cost on the un-scaled state
          double d[NX];
#pragma unroll
          for (int i = 0; i < NX; ++i) d[i] = x[i] - par[NP + i];
#pragma unroll
          for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j < NX; ++j) val += d[i] * pc.cost[PB::O_WX + i * NX + j] * d[j];
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const int col = c0 + c;
            double gacc = 0.0;
#pragma unroll
            for (int j = 0; j < NX; ++j) gacc += (pc.cost[PB::O_WX + col * NX + j] + pc.cost[PB::O_WX + j * NX + col]) * d[j];
            double scol = sz[0];
#pragma unroll
            for (int q2 = 1; q2 < NX; ++q2) scol = col == q2 ? sz[q2] : scol;
            gx[c] = gacc * scol;
            gw[c] = 0.0;
#pragma unroll
            for (int rr = 0; rr < NX; ++rr) {
              chx[rr][c] = (pc.cost[PB::O_WX + rr * NX + col] + pc.cost[PB::O_WX + col * NX + rr]) * sz[rr] * scol;
              chw[rr][c] = 0.0;
            }
          }
        } else {
          // r = h(x) - y, kb = (Wy + Wy^T) r, then Jy and the kb-contracted Hessian of h
#pragma unroll
          for (int a = 0; a < NY; ++a) kby[a] = 0.0;
          MS::mjh(x, ue, par, kby, yv, Jy, Hy);   // first call: only y is used
#pragma unroll
          for (int a = 0; a < NY; ++a) r[a] = yv[a] - sd[MU + a];
#pragma unroll
          for (int a = 0; a < NY; ++a) {
            double acc = 0.0;
#pragma unroll
            for (int b2 = 0; b2 < NY; ++b2) {
              acc += (pc.cost[PB::O_WY + a * NY + b2] + pc.cost[PB::O_WY + b2 * NY + a]) * r[b2];
              val += r[a] * pc.cost[PB::O_WY + a * NY + b2] * r[b2];
            }
            kby[a] = acc;
          }
          MS::mjh(x, ue, par, kby, yv, Jy, Hy);
#pragma unroll
          for (int i = 0; i < NX; ++i) ws_[i] = wv[i] * sw[i];
#pragma unroll
          for (int i = 0; i < NX; ++i)
#pragma unroll
            for (int j = 0; j < NX; ++j) val += ws_[i] * pc.cost[PB::O_WW + i * NX + j] * ws_[j];
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const int col = c0 + c;
            double scol = sz[0], swcol = sw[0];
#pragma unroll
            for (int q2 = 1; q2 < NX; ++q2) { scol = col == q2 ? sz[q2] : scol; swcol = col == q2 ? sw[q2] : swcol; }
            double gacc = 0.0, gwacc = 0.0;
#pragma unroll
            for (int a = 0; a < NY; ++a) {
              double jac = Jy[a * NX + 0];
#pragma unroll
              for (int q2 = 1; q2 < NX; ++q2) jac = col == q2 ? Jy[a * NX + q2] : jac;
              gacc += kby[a] * jac;
            }
#pragma unroll
            for (int j = 0; j < NX; ++j) gwacc += (pc.cost[PB::O_WW + col * NX + j] + pc.cost[PB::O_WW + j * NX + col]) * ws_[j];
            gx[c] = gacc * scol;
            gw[c] = gwacc * swcol;
#pragma unroll
            for (int rr = 0; rr < NX; ++rr) {
              // (Jy^T (Wy + Wy^T) Jy)[rr][col] + Hy(kb)[rr][col]
              double acc = 0.0;
#pragma unroll
              for (int a = 0; a < NY; ++a) {
                double t2 = 0.0;
#pragma unroll
                for (int b2 = 0; b2 < NY; ++b2) {
                  double jb = Jy[b2 * NX + 0];
#pragma unroll
                  for (int q2 = 1; q2 < NX; ++q2) jb = col == q2 ? Jy[b2 * NX + q2] : jb;
                  t2 += (pc.cost[PB::O_WY + a * NY + b2] + pc.cost[PB::O_WY + b2 * NY + a]) * jb;
                }
                acc += Jy[a * NX + rr] * t2;
              }
              double hy = 0.0;
#pragma unroll
              for (int q2 = 0; q2 < NX; ++q2) {
                const double hv2 = Hy[rr >= q2 ? rr * (rr + 1) / 2 + q2 : q2 * (q2 + 1) / 2 + rr];
                hy = col == q2 ? hv2 : hy;
              }
              chx[rr][c] = (acc + hy) * sz[rr] * scol;
              chw[rr][c] = (pc.cost[PB::O_WW + rr * NX + col] + pc.cost[PB::O_WW + col * NX + rr]) * sw[rr] * swcol;
            }
          }
        }
        if (act && g == 0) fpart += val;
      }
      {  // stage points, slopes, Phi, defect
        FastD kk[4][NX], uf[MU1];
#pragma unroll
        for (int i = 0; i < MU; ++i) uf[i] = FastD(ue[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          FastD Xf[NX];
#pragma unroll
          for (int s2 = 0; s2 < NX; ++s2) {
            double acc = x[s2];
            if (i == 1) acc += (h * a10) * kk[0][s2].v;
            if (i == 2) acc += (h * a20) * kk[0][s2].v + (h * a21) * kk[1][s2].v;
            if (i == 3) acc += (h * a32) * kk[2][s2].v;
            X[i][s2] = acc;
            Xf[s2] = FastD(acc);
          }
          if (i < order) {
            M::ode(Xf, uf, par, h, kk[i]);
          } else {
#pragma unroll
            for (int s2 = 0; s2 < NX; ++s2) kk[i][s2] = FastD(0.0);
          }
        }
        if (act && g == 0) {
#pragma unroll
          for (int s2 = 0; s2 < NX; ++s2) {
            const double phi = x[s2] + hb[0] * kk[0][s2].v + hb[1] * kk[1][s2].v + hb[2] * kk[2][s2].v + hb[3] * kk[3][s2].v;
            l.c[k * NX + s2] = l.Z[(k + 1) * NZ + s2] - (phi * isz[s2] + wv[s2]);
          }
        }
      }
      // adjoint weights of the slopes (reverse over the stages)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s2 = 0; s2 < NX; ++s2) kb[i][s2] = hb[i] * lamp[s2];
#pragma unroll
      for (int j = 3; j >= 1; --j) {
        if (j < order) {
          double fx[NX * NX], t2[NX];
          MS::jx(X[j], ue, par, fx);
#pragma unroll
          for (int n = 0; n < NX; ++n) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < NX; ++m) acc += fx[m * NX + n] * kb[j][m];
            t2[n] = acc;
          }
#pragma unroll
          for (int n = 0; n < NX; ++n) {
            if (j == 3) kb[2][n] += (h * a32) * t2[n];
            if (j == 2) { kb[1][n] += (h * a21) * t2[n]; kb[0][n] += (h * a20) * t2[n]; }
            if (j == 1) kb[0][n] += (h * a10) * t2[n];
          }
        }
      }
      // tangent columns (state directions only) and the second-order adjoint, stage by stage
      dp scr = l.W + (size_t)k * NZ * WP;            // [NX][NX] tangent block of the interval
      double dXc[NX][CPL], dX2[NX][CPL], dPhi[NX][CPL], G[NX][CPL];
#pragma unroll
      for (int s2 = 0; s2 < NX; ++s2)
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          dXc[s2][c] = (c0 + c == s2) ? 1.0 : 0.0;
          dX2[s2][c] = dXc[s2][c];
          dPhi[s2][c] = dXc[s2][c];
          G[s2][c] = 0.0;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < order) {
          if (act) {
#pragma unroll
            for (int s2 = 0; s2 < NX; ++s2)
#pragma unroll
              for (int c = 0; c < CPL; ++c) scr[s2 * NX + c0 + c] = dXc[s2][c];
          }
          __syncthreads();
          double dK[NX][CPL], v[NX][CPL];
          {
            double J[NX * MZ], H[MZ * (MZ + 1) / 2];
            MS::jh(X[i], ue, par, kb[i], J, H);
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
#pragma unroll
              for (int m = 0; m < NX; ++m) {
                double acc = 0.0;
#pragma unroll
                for (int n = 0; n < NX; ++n) acc += J[m * MZ + n] * dXc[n][c];
                dK[m][c] = acc;
              }
#pragma unroll
              for (int a = 0; a < NX; ++a) {
                double acc = 0.0;
#pragma unroll
                for (int n = 0; n < NX; ++n) acc += H[a >= n ? a * (a + 1) / 2 + n : n * (n + 1) / 2 + a] * dXc[n][c];
                v[a][c] = acc;
              }
            }
          }
#pragma unroll
          for (int r = 0; r < NX; ++r) {
            double col_r[NX];
#pragma unroll
            for (int a = 0; a < NX; ++a) col_r[a] = scr[a * NX + r];
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
              double acc = 0.0;
#pragma unroll
              for (int a = 0; a < NX; ++a) acc += col_r[a] * v[a][c];
              G[r][c] += acc;
            }
          }
#pragma unroll
          for (int s2 = 0; s2 < NX; ++s2)
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
              const double e = (c0 + c == s2) ? 1.0 : 0.0;
              dPhi[s2][c] += hb[i] * dK[s2][c];
              if (i == 0) { dXc[s2][c] = e + (h * a10) * dK[s2][c]; dX2[s2][c] = e + (h * a20) * dK[s2][c]; }
              if (i == 1) dXc[s2][c] = dX2[s2][c] + (h * a21) * dK[s2][c];
              if (i == 2) dXc[s2][c] = e + (h * a32) * dK[s2][c];
            }
          __syncthreads();
        }
      }
      if (act) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int col = c0 + c;
          double scol = sz[0];
#pragma unroll
          for (int q2 = 1; q2 < NX; ++q2) scol = col == q2 ? sz[q2] : scol;
          l.grad[k * NZ + col] = gx[c];
          l.grad[k * NZ + NX + col] = gw[c];
#pragma unroll
          for (int m = 0; m < NX; ++m) {
            l.AB[(k * NX + m) * ABP + col] = dPhi[m][c] * scol * isz[m];       // A = d(Phi / s_x) / d x_s
            l.AB[(k * NX + m) * ABP + NX + col] = m == col ? 1.0 : 0.0;        // B = I (additive noise)
          }
#pragma unroll
          for (int r = 0; r < NX; ++r) {
            if (r >= col) {   // one value for (r, col) and (col, r): exact symmetry
              const double hxx = chx[r][c] - sz[r] * scol * G[r][c], hww = chw[r][c];
              l.W[((size_t)k * NZ + r) * WP + col] = hxx;
              l.W[((size_t)k * NZ + col) * WP + r] = hxx;
              l.W[((size_t)k * NZ + NX + r) * WP + NX + col] = hww;
              l.W[((size_t)k * NZ + NX + col) * WP + NX + r] = hww;
            }
            l.W[((size_t)k * NZ + NX + r) * WP + col] = 0.0;             // state / noise cross block
            l.W[((size_t)k * NZ + col) * WP + NX + r] = 0.0;
          }
        }
      }
    }
    __syncthreads();
    const double f = block_reduce<OpSum>(fpart, l.red);
    __syncthreads();
    return f;
  }

  __device__ OCP_PHASE static double eval_derivs_call(lds_double* lbase, double* ws) {
    return eval_derivs_body(uni(lbase), uni(ws));
  }
  // Cooperative models exchange partial sums between lanes through LDS with workgroup barriers in between; as a real
  // function (barriers kept as instructions) that is what was validated, so they keep the call.
  __device__ __forceinline__ static double eval_derivs(lds_double* lbase, double* ws, bool reuse = false) {
#ifdef HILO_COOP_CALL
    if constexpr (COOP) return uni(eval_derivs_call(lbase, ws));
#else
    // inlined: the kernel's register budget (one wave per SIMD) instead of a callee's; `reuse`: the iterate is the trial point the
    // line search evaluated last - the learned term's value / gradient / Hessian at its stage points are in l.gpc
    if constexpr (COOP) return uni(eval_derivs_body(uni(lbase), uni(ws), uni(reuse)));
#endif
    // cooperative collocation: as a real call - the phase is long (N / CG elimination rounds, N direction passes), and inlined it
    // shares the register file with everything the solver loop keeps alive: spills inside its loops, which one wave per SIMD sends
    // to HBM (1.5 TB of scratch writes per 8192-instance launch of configuration 5's DAE problem)
    else if constexpr (XCW > 0) return uni(eval_derivs_call(lbase, ws));
    else if constexpr (SYM) return eval_derivs_sym(lbase, ws);
    else if constexpr (SYM_MHE) return eval_derivs_sym_mhe(lbase, ws);
    else return eval_derivs_body(lbase, ws);
  }

  // dual residual of slot e: grad + J^T lam - zL + zU
  __device__ static double dual_res(const Lds l, int N, int e) {
    const int k = e / NZ, i = e - k * NZ;
    double r = l.grad[e] - l.zL[e] + l.zU[e];
    if (i < NX && k >= 1) r += l.lam[(k - 1) * NX + i];
    if (k < N) {
#pragma unroll
      for (int m = 0; m < NX; ++m) r -= l.AB[(k * NX + m) * ABP + i] * l.lam[k * NX + m];
      if constexpr (NC > 0) {
#pragma unroll
        for (int m = 0; m < NC; ++m) r += l.Jd[(k * NC + m) * NZ + i] * l.cnu[k * NC + m];  // inactive rows hold zeros
      }
    }
    return r;
  }

  // ---- one pass over the iterate per iteration: scaled optimality error (W&B eq. 5) AND the barrier terms of the Newton system.
  // Per slot: dual residual, multiplier sums, the complementarity products p = slack * multiplier (their largest and smallest
  // value give max |p - mu| for EVERY barrier parameter: max(pmax - mu, mu - pmin) - the barrier update needs no further pass),
  // and with the reciprocal slacks already at hand  sig = zL/(z - lb) + zU/(ub - z)  (added to the diagonal of the stage's
  // Hessian block, stage_rhs) and  q = 1/(z - lb) - 1/(ub - z)  (parked in the right-hand-side column: rb = grad - mu q is
  // completed by finish_rhs once the barrier parameter of this iteration is known).  Keeps the divisions out of the recursion.
  struct KktErr { double dual_s, prim, s_c, i_s_c, pmax, pmin, theta; };
  __device__ static void stage_rhs(const Lds l, int N, int e, double sg, double r, bool replace) {
    const int k = e / NZ, i = e - k * NZ;
    l.sig[e] = sg;
    if (k < N) {
      dp w = l.W + (size_t)(k * NZ + i) * WP;
      w[i] = replace ? sg : w[i] + sg;
      w[NZ] = r;
    } else if (i < NX) l.rbN[i] = r;
  }
  __device__ __forceinline__ static KktErr kkt_pass(const Lds l, double nb) {
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N;
    double dmax = 0.0, lsum = 0.0, zsum = 0.0, pmax = 0.0, cmax = 0.0, cmin = INFINITY, th = 0.0;
    DTICK0
    {
      // U slots per lane and trip, everything a slot reads (gradient, multipliers, its column of [A B], inequality rows, bounds,
      // the diagonal entry of W it updates) requested before the first use and the slot's arithmetic written without branches:
      // a trip of the plain predicated loop is a chain of ten dependent memory round trips (LDS: ~100 clocks each with one wave
      // per SIMD; workspace mode: global memory)
      constexpr int U = BIG ? 3 : 2;
      const int SLT = (N + 1) * NZ;
      const FreeTest free_at = free_test(pc);
      for (int base = 0; base < SLT; base += OCP_TPB * U) {
        double gr[U], lp[U], ab[U][NX], lm[U][NX], wd[U], jd[U][NC > 0 ? NC : 1], cn[U][NC > 0 ? NC : 1];
        double lbv[U], ubv[U], zv[U], zlv[U], zuv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e0 = base + u * OCP_TPB + (int)threadIdx.x;
          ok[u] = e0 < SLT;
          const int e = ok[u] ? e0 : 0, k = e / NZ, i = e - k * NZ, kc = k < N ? k : N - 1, kp = k >= 1 ? k - 1 : 0;
          const int ix = i < NX ? i : 0;
          gr[u] = l.grad[e];
          lp[u] = l.lam[kp * NX + ix];
#pragma unroll
          for (int m = 0; m < NX; ++m) { ab[u][m] = l.AB[(kc * NX + m) * ABP + i]; lm[u][m] = l.lam[kc * NX + m]; }
          wd[u] = l.W[(size_t)(kc * NZ + i) * WP + i];
          if constexpr (NC > 0) {
#pragma unroll
            for (int m = 0; m < NC; ++m) { jd[u][m] = l.Jd[(kc * NC + m) * NZ + i]; cn[u][m] = l.cnu[kc * NC + m]; }
          }
          lbv[u] = l.lbA[e]; ubv[u] = l.ubA[e]; zv[u] = l.Z[e]; zlv[u] = l.zL[e]; zuv[u] = l.zU[e];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e0 = base + u * OCP_TPB + (int)threadIdx.x;
          const int e = ok[u] ? e0 : 0, k = e / NZ, i = e - k * NZ;
          const bool fr = ok[u] && free_at(k, i);
          const double zl = zlv[u], zu = zuv[u];
          double r = gr[u] - zl + zu;                                   // dual_res, same order of the sum
          r = (i < NX && k >= 1) ? r + lp[u] : r;
          double r2 = r;
#pragma unroll
          for (int m = 0; m < NX; ++m) r2 -= ab[u][m] * lm[u][m];
          if constexpr (NC > 0) {
#pragma unroll
            for (int m = 0; m < NC; ++m) r2 += jd[u][m] * cn[u][m];
          }
          r = k < N ? r2 : r;
          dmax = fr ? nmax(dmax, fabs(r)) : dmax;
          zsum += fr ? fabs(zl) + fabs(zu) : 0.0;
          const bool hl = fr && lbv[u] > -INFINITY, hu = fr && ubv[u] < INFINITY;
          const double sl = zv[u] - lbv[u], su = ubv[u] - zv[u];
          const double isl = rcp_fast(hl ? sl : 1.0), isu = rcp_fast(hu ? su : 1.0), pl = sl * zl, pu = su * zu;
          cmax = hl ? fmax(cmax, pl) : cmax;
          cmin = hl ? fmin(cmin, pl) : cmin;
          cmax = hu ? fmax(cmax, pu) : cmax;
          cmin = hu ? fmin(cmin, pu) : cmin;
          double sg = hl ? zl * isl : 0.0, q = hl ? isl : 0.0;
          sg = hu ? sg + zu * isu : sg;
          q = hu ? q - isu : q;
          if (ok[u]) {                                                  // stage_rhs(l, N, e, sg, q, false)
            l.sig[e] = sg;
            if (k < N) {
              dp w = l.W + (size_t)(k * NZ + i) * WP;
              w[i] = wd[u] + sg;
              w[NZ] = q;
            } else if (i < NX) l.rbN[i] = q;
          }
        }
      }
      const int NV = N * NX;
      for (int base = 0; base < NV; base += OCP_TPB * U) {
        double cv[U], lv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e0 = base + u * OCP_TPB + (int)threadIdx.x;
          ok[u] = e0 < NV;
          cv[u] = l.c[ok[u] ? e0 : 0];
          lv[u] = l.lam[ok[u] ? e0 : 0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double ca = fabs(cv[u]);
          pmax = ok[u] ? nmax(pmax, ca) : pmax;
          th += ok[u] ? ca : 0.0;
          lsum += ok[u] ? fabs(lv[u]) : 0.0;
        }
      }
    }
    double ncon = 0.0;
    if constexpr (NC > 0) {  // slack block: dual residual -nu - vL + vU, primal residual d - s; csig, and q_s parked in crb
      OCP_FOR(e, N * NC) {
        const int m = e % NC;
        double sg = 0.0, q = 0.0;
        if (row_on(pc, e / NC, m)) {
          dmax = nmax(dmax, fabs(-l.cnu[e] - l.cvL[e] + l.cvU[e]));
          pmax = nmax(pmax, fabs(l.cd[e] - l.cs[e]));
          th += fabs(l.cd[e] - l.cs[e]);
          lsum += fabs(l.cnu[e]);
          zsum += fabs(l.cvL[e]) + fabs(l.cvU[e]);
          if (pc.dlb[m] > -INFINITY) {
            const double sl = l.cs[e] - pc.dlb[m], is = rcp_fast(sl), p = sl * l.cvL[e];
            cmax = fmax(cmax, p);
            cmin = fmin(cmin, p);
            sg += l.cvL[e] * is;
            q += is;
          }
          if (pc.dub[m] < INFINITY) {
            const double su = pc.dub[m] - l.cs[e], is = rcp_fast(su), p = su * l.cvU[e];
            cmax = fmax(cmax, p);
            cmin = fmin(cmin, p);
            sg += l.cvU[e] * is;
            q -= is;
          }
        }
        l.csig[e] = sg;
        l.crb[e] = q;
      }
      ncon = (double)N * pc.nc + pc.nc_term;
    }
    KktErr r;
    DTICK(12)
    if constexpr (OCP_TPB == 64) {
      double rv[7] = {th, cmax, cmin, dmax, pmax, lsum, zsum};
      WaveReduceN<R_SUM, R_MAX2, R_MIN, R_MAX, R_MAX, R_SUM, R_SUM>::run(rv);
      r.theta = rv[0]; r.pmax = rv[1]; r.pmin = rv[2]; dmax = rv[3]; pmax = rv[4]; lsum = rv[5]; zsum = rv[6];
    } else {
      r.theta = uni(block_reduce<OpSum>(th, l.red));
      r.pmax = uni(block_reduce<OpMax2>(cmax, l.red));
      r.pmin = uni(block_reduce<OpMin>(cmin, l.red));
      dmax = block_reduce<OpMax>(dmax, l.red);
      pmax = block_reduce<OpMax>(pmax, l.red);
      lsum = block_reduce<OpSum>(lsum, l.red);
      zsum = block_reduce<OpSum>(zsum, l.red);
    }
    // (scalings with reciprocals - v_rcp_f64 and two Newton steps, <= 1 ulp - instead of IEEE divisions: five dependent
    // 13-instruction sequences on wave-uniform values otherwise)
    const double i_smax = rcp_fast(pc.s_max);
    const double s_d = fmax(pc.s_max, (lsum + zsum) * rcp_fast(N * NX + ncon + nb)) * i_smax;
    // a NaN multiplier or defect must reach the error measure (fmax would drop it): through the sums
    const double poison = (lsum + zsum + r.theta) * 0.0;      // 0, or NaN
    r.s_c = uni(fmax(pc.s_max, zsum * rcp_fast(nb)) * i_smax + poison);
    r.i_s_c = uni(rcp_fast(r.s_c));
    r.dual_s = uni(dmax * rcp_fast(s_d));
    r.prim = uni(pmax);
    DTICK(13)
    return r;
  }
  // complementarity error for barrier parameter mu from the extreme products
  __device__ __forceinline__ static double compl_of(const KktErr& r, double mu) { return fmax(fmax(r.pmax - mu, mu - r.pmin), 0.0); }
  // switching condition of the filter line search (W&B eq. 19):  alpha (-dphi)^s_phi > delta_ls th0^s_theta  with nd = -dphi > 0.
  // Two f64 pow() are 540 instructions on one dependent chain per trial point; decided here in the logarithm with the hardware's
  // f32 log2 of the mantissas (error of the difference < 2e-6 in units of log2) whenever the two sides are further apart than
  // 2^(1e-4) - otherwise (and for zero / non-finite / extreme arguments) by the pow() expression itself: the same decision in every
  // case (tests/test_switching_rule.py restates the rule in numpy and compares the decisions).
  __device__ __forceinline__ static bool switching(const OcpConst& pc, double alpha, double nd, double th0) {
    const double dls = pc.delta_ls;
    // (inside these ranges neither power nor the products can overflow or underflow - outside them the pow() expression has a
    // semantics of its own, inf > inf or 0 > 0, which only the expression itself reproduces)
    const bool plain = alpha > 1e-30 && alpha < 1e30 && nd > 1e-100 && nd < 1e100 && th0 > 1e-100 && th0 < 1e100 && dls > 1e-30 && dls < 1e30;
    if (plain) {
      auto lg2 = [](double x) {
        return (double)__builtin_amdgcn_frexp_exp(x) + (double)__builtin_amdgcn_logf((float)__builtin_amdgcn_frexp_mant(x));
      };
      const double d = lg2(alpha) + pc.s_phi * lg2(nd) - lg2(dls) - pc.s_theta * lg2(th0);
      if (fabs(d) > 1e-4) return d > 0.0;
    }
    return alpha * pow(nd, pc.s_phi) > dls * pow(th0, pc.s_theta);
  }
  // right-hand sides of the Newton system once mu is fixed: rb = grad - mu q (q parked by kkt_pass), crb = -mu q_s
  __device__ static void finish_rhs(const Lds l, double mu) {
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N;
    {   // U slots per lane and trip, reads first (see kkt_pass)
      constexpr int U = BIG ? 3 : 2;
      const int SLT = (N + 1) * NZ;
      for (int base = 0; base < SLT; base += OCP_TPB * U) {
        double gr[U], qv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e0 = base + u * OCP_TPB + (int)threadIdx.x;
          ok[u] = e0 < SLT;
          const int e = ok[u] ? e0 : 0, k = e / NZ, i = e - k * NZ;
          gr[u] = l.grad[e];
          qv[u] = k < N ? l.W[(size_t)(k * NZ + i) * WP + NZ] : l.rbN[i < NX ? i : 0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = base + u * OCP_TPB + (int)threadIdx.x;
          if (!ok[u]) continue;
          const int k = e / NZ, i = e - k * NZ;
          if (k < N) l.W[(size_t)(k * NZ + i) * WP + NZ] = gr[u] - mu * qv[u];
          else if (i < NX) l.rbN[i] = gr[u] - mu * qv[u];
        }
      }
    }
    if constexpr (NC > 0) {
      OCP_FOR(e, N * NC) l.crb[e] = -mu * l.crb[e];
    }
    __syncthreads();
  }
  // number of finite bounds of the free slots and the inequality rows: a constant of the problem instance
  __device__ static double count_bounds(const Lds l) {
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N;
    double nb = 0.0;
    OCP_FOR(e, (N + 1) * NZ) {
      if (!is_free(pc, e / NZ, e % NZ)) continue;
      nb += (l.lbA[e] > -INFINITY ? 1.0 : 0.0) + (l.ubA[e] < INFINITY ? 1.0 : 0.0);
    }
    if constexpr (NC > 0) {
      OCP_FOR(e, N * NC) {
        const int m = e % NC;
        if (!row_on(pc, e / NC, m)) continue;
        nb += (pc.dlb[m] > -INFINITY ? 1.0 : 0.0) + (pc.dub[m] < INFINITY ? 1.0 : 0.0);
      }
    }
    return uni(fmax(1.0, block_reduce<OpSum>(nb, l.red)));
  }

  // lower Cholesky factor of the n x n block M (row pitch ld) with reciprocal pivots: L (n x n), invd = 1/diag(L)
  template <int n>
  __device__ __forceinline__ static bool small_chol(lds_cdouble* M, int ld, double* L, double* invd) {
    bool pd = true;
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double s = M[j * ld + j];
#pragma unroll
      for (int q = 0; q < j; ++q) s -= L[j * n + q] * L[j * n + q];
      if (!(s > 0.0)) { pd = false; s = 1.0; }
      const double id = rsq_fast(s);
      invd[j] = id;
      L[j * n + j] = s * id;
#pragma unroll
      for (int i = j + 1; i < n; ++i) {
        double v = M[i * ld + j];
#pragma unroll
        for (int q = 0; q < j; ++q) v -= L[i * n + q] * L[j * n + q];
        L[i * n + j] = v * id;
      }
    }
    return pd;
  }
  // the same factorisation of a block held in registers (row-major n x n, lower triangle used)
  template <int n>
  __device__ __forceinline__ static bool small_chol_reg(const double* M, double* L, double* invd) {
    if constexpr (n == 2) {
      // L11 = sqrt(r00), L21 = r10 / L11, L22 = sqrt(det / r00): rsqrt(r00) and rsqrt(det) do not depend on each other - one
      // reciprocal-square-root chain on the critical path of a Riccati stage instead of two (same cancellation as r11 - L21^2)
      const double r00 = M[0], r10 = M[2], r11 = M[3];
      const double det = fma(r00, r11, -r10 * r10);
      const bool ok0 = r00 > 0.0, ok1 = det > 0.0;
      const double i0 = rsq_fast(ok0 ? r00 : 1.0), idt = rsq_fast(ok1 ? det : 1.0);
      const double s0 = (ok0 ? r00 : 1.0) * i0;          // sqrt(r00)
      invd[0] = i0;
      L[0] = s0;
      L[2] = r10 * i0;
      invd[1] = idt * s0;                                 // 1 / sqrt(det / r00)
      L[3] = (ok1 ? det : 1.0) * idt * i0;                // sqrt(det / r00)
      return ok0 && ok1;
    }
    bool pd = true;
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double s = M[j * n + j];
#pragma unroll
      for (int q = 0; q < j; ++q) s -= L[j * n + q] * L[j * n + q];
      if (!(s > 0.0)) { pd = false; s = 1.0; }
      const double id = rsq_fast(s);
      invd[j] = id;
      L[j * n + j] = s * id;
#pragma unroll
      for (int i = j + 1; i < n; ++i) {
        double v = M[i * n + j];
#pragma unroll
        for (int q = 0; q < j; ++q) v -= L[i * n + q] * L[j * n + q];
        L[i * n + j] = v * id;
      }
    }
    return pd;
  }
  // y <- (L L^T)^-1 y
  template <int n>
  __device__ __forceinline__ static void small_solve(const double* L, const double* invd, double* y) {
#pragma unroll
    for (int a = 0; a < n; ++a) {
      double s = y[a];
#pragma unroll
      for (int q = 0; q < a; ++q) s -= L[a * n + q] * y[q];
      y[a] = s * invd[a];
    }
#pragma unroll
    for (int a = n - 1; a >= 0; --a) {
      double s = y[a];
#pragma unroll
      for (int q = a + 1; q < n; ++q) s -= L[q * n + a] * y[q];
      y[a] = s * invd[a];
    }
  }

  // ---- backward recursion with the cost-to-go in registers (MFMA_STAGE policies with inputs, no held inputs) ----------------
  // The accumulator layout of v_mfma_f64_16x16x4 puts entry (row g + 4 r, column q) of the stage matrix
  //     M = [A B | -c]^T (P_{k+1} [A B | -c] + [0 | p_{k+1}]) + [H_k | r_k]          (column NZ = right-hand side)
  // into register r of lane 16 g + q.  The same lane then computes entry (g + 4 r, q) of P_k (q < NX) or of p_k (q = NZ):
  //     P_k[i][j] = sym(M_xx)[i][j] - w_i . w_j,   w_j = L^-1 M_ux[:, j],   R_k = M_uu = L L^T     (p_k: column NZ)
  // - with the columns of M_ux fetched from the lanes that hold them (ds_bpermute: the LDS crossbar without a store / barrier /
  // load round trip) - and that register IS the A operand P_k[q][4 kb + g] of the next stage's first product (P symmetric, and
  // kept bitwise symmetric: both lanes of a pair run the same arithmetic on swapped factors).  So the recursion never waits
  // for LDS: feedback, closed-loop matrices, P_k and p_k are stored for the forward sweep on the side, the operands of the
  // next stage are fetched while the pivot block of this one is factored (one basic block per stage: the positivity of the
  // pivots is collected and tested after the loop).
  // Round 3: the stage blocks sit in LDS in the PADDED form the products read - [A B | -c] with pitch NZ + 1, [H + Sigma | r]
  // with pitch NZ + 1 (stage_rhs) - so a lane fetches each operand with ONE unconditional load at a clamped column
  // (columns beyond NZ and rows beyond NZ of M are duplicates that nothing reads; the same register is the B operand of the
  // first product, the A operand of the second and the open-loop coefficient of the closed-loop update), and every output of
  // a stage ([P | p], [K | kff], [Acl | bcl], pitch NX + 1) leaves through one predicated store.
  static constexpr int KB_ = (NX + 3) / 4, RB_ = (NZ + 3) / 4;
  // Round 5: DUPLICATED INPUT ROWS.  The 16 x 16 tile of the second product has rows to spare (NZ + 1 <= 4 RB_ of them are in use):
  // the lanes q >= 4 RB_ feed the input columns of [A B | -c] once more as A operand - rows 4 (RB_ + a) + g of the tile, g = 0..3, all
  // equal row NX + a of M - so that accumulator register RB_ + a of EVERY lane (g, q) holds M_ux[a][q]: the lane's own column of the
  // input rows and (lanes NX + c) the pivot block arrive without a cross-lane fetch.  (The same register is the B operand of the first
  // product: columns >= 4 RB_ of T become copies of P B - nothing reads them.  NZ % 4 != 0 keeps the right-hand-side column NZ below
  // 4 RB_.)  With at most two inputs the pivot block R = L D L^T is then factored WITHOUT square roots and with its two reciprocals
  // in parallel (pivot_inverse): 1 / d_0 = 1 / r00 and 1 / d_1 = r00 / det R, l = r10 / r00,
  //     P_k[i][j] = sym(M_xx)[i][j] - m_i0 m_j0 / d_0 - (m_i1 - l m_i0)(m_j1 - l m_j0) / d_1,      m_j = M_ux[:, j]
  // - the arithmetic (and the rounding behaviour) of the Cholesky form P = M_xx - w_i . w_j, w = L^-1 D^-1/2 m, with one reciprocal
  // chain on the critical path instead of two dependent reciprocal square roots.  (The closed form m_i^T adj(R) m_j / det R was
  // tried first and is NOT usable: with a state constraint active at stage k + 1 the cost-to-go carries sigma j j^T, sigma ~ 1e10,
  // R and M_ux both contain a rank-one term of that size and the adjugate form cancels terms of order sigma^3 down to sigma -
  // no digit left; found by the constrained parity tests.)  Positivity of the pivots = r00 > 0 and det R > 0.
#ifndef HILO_RIC_DUP
#define HILO_RIC_DUP 1
#endif
#ifndef HILO_RIC_TADD
#define HILO_RIC_TADD 1
#endif
#ifndef HILO_FWD_SGPR
#define HILO_FWD_SGPR 1
#endif
  static constexpr bool DUPU = HILO_RIC_DUP && OCP_TPB == 64 && NZ + 1 <= 16 && NX <= 16 && NU >= 1 && NU <= 2 && NH == 0 && NZ % 4 != 0 &&
                               RB_ + NU <= 4;
  static constexpr int RD_ = DUPU ? RB_ + NU : RB_;   // accumulator registers of the stage matrix in use
  // row of the stage matrix that register r of row group g holds (clamped where the tile has no row)
  __device__ __forceinline__ static int stage_row(int r, int g) {
    if (DUPU && r >= RB_) return NX + (r - RB_);
    const int i = g + 4 * r;
    return (NZ % 4 == 0 || i < NZ) ? i : NZ - 1;
  }
  struct StageOps {
    double b1[KB_], bi[KB_][NU > 0 ? NU : 1];
    v4d M0;
  };
  // R = [r00 r10; r10 r11] = L D L^T:  i0 = 1 / d_0,  i1 = 1 / d_1 = r00 / det R,  l = r10 / r00  (the two reciprocals are independent)
  struct PivotInv { double i0, i1, l; bool ok; };
  // reciprocal of a pivot: v_rcp_f64 (about 2^-26 relative) + HILO_RCP_PIVOT_NEWTON Newton steps; the lanes (i, j) and (j, i) run the
  // same instructions on the same bits, so the cost-to-go stays bitwise symmetric whatever the last bits are
#ifndef HILO_RCP_PIVOT_NEWTON
#define HILO_RCP_PIVOT_NEWTON 2
#endif
  __device__ __forceinline__ static double rcp_pivot(double x) {
    double r = __builtin_amdgcn_rcp(x);
#pragma unroll
    for (int q = 0; q < HILO_RCP_PIVOT_NEWTON; ++q) r = fma(fma(-x, r, 1.0), r, r);
    return r;
  }
  __device__ __forceinline__ static PivotInv pivot_inverse(double r00, double r10, double r11) {
    PivotInv p;
    if constexpr (NU == 2) {
      const double det = fma(r00, r11, -r10 * r10);
      p.ok = r00 > 0.0 && det > 0.0;
      p.i0 = rcp_pivot(r00);
      p.i1 = r00 * rcp_pivot(det);
      p.l = r10 * p.i0;
    } else {
      p.ok = r00 > 0.0;
      p.i0 = rcp_pivot(r00);
      p.i1 = 0.0;
      p.l = 0.0;
    }
    return p;
  }
  // what a stage reads from the iterate, before any arithmetic: in workspace mode these loads are issued TWO stages ahead
  // (a global-memory round trip is about two stage times at one wave per SIMD) and turned into operands one stage ahead
  struct StageRaw {
    double b1[KB_], bi[KB_][NU > 0 ? NU : 1], w[RD_];
    double csig[NC > 0 ? NC : 1], cdv[NC > 0 ? NC : 1], csv[NC > 0 ? NC : 1], crb[NC > 0 ? NC : 1], jq[NC > 0 ? NC : 1];
    double ji[RD_][NC > 0 ? NC : 1];
  };
  __device__ __forceinline__ static void stage_load(const Lds l, int k, int q, int g, StageRaw& w) {
    const int qc = q < NZ ? q : NZ;
    int qa = qc;   // column of [A B | -c] this lane feeds to the products (DUPU: lanes q >= 4 RB_ feed the input columns again)
    if constexpr (DUPU) {
      if (q >= 4 * RB_) qa = ((q - 4 * RB_) >> 2) < NU ? NX + ((q - 4 * RB_) >> 2) : NZ;
    }
    cdp AB = l.AB + (size_t)k * NX * ABP;
    cdp Wk = l.W + (size_t)k * NZ * WP;
#pragma unroll
    for (int kb = 0; kb < KB_; ++kb) {
      const int kk = 4 * kb + g;
      const int kc = (NX % 4 == 0 || kk < NX) ? kk : NX - 1;
      w.b1[kb] = AB[kc * ABP + qa];
#pragma unroll
      for (int a = 0; a < NU; ++a) w.bi[kb][a] = AB[kc * ABP + NX + a];
    }
#pragma unroll
    for (int r = 0; r < RD_; ++r) {
      const int ic = stage_row(r, g);
      w.w[r] = Wk[ic * WP + qc];
      if constexpr (NC > 0) {
#pragma unroll
        for (int m = 0; m < NC; ++m) w.ji[r][m] = l.Jd[(k * NC + m) * NZ + ic];
      }
    }
    if constexpr (NC > 0) {
      const int jj = q < NZ ? q : 0;
#pragma unroll
      for (int m = 0; m < NC; ++m) {
        const int rr = k * NC + m;
        w.csig[m] = l.csig[rr]; w.cdv[m] = l.cd[rr]; w.csv[m] = l.cs[rr]; w.crb[m] = l.crb[rr]; w.jq[m] = l.Jd[rr * NZ + jj];
      }
    }
  }
  __device__ __forceinline__ static void stage_finish(const StageRaw& w, double delta, int q, int g, StageOps& o) {
#pragma unroll
    for (int kb = 0; kb < KB_; ++kb) {
      const int kk = 4 * kb + g;
      o.b1[kb] = (NX % 4 == 0 || kk < NX) ? w.b1[kb] : 0.0;
#pragma unroll
      for (int a = 0; a < NU; ++a) o.bi[kb][a] = w.bi[kb][a];
    }
    o.M0 = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < RD_; ++r) {
      const int ic = stage_row(r, g);
      double h = w.w[r];
      if (ic == q) h += delta;
      if constexpr (NC > 0) {  // eliminated slack rows: + Jd^T (Sigma_s + delta) Jd, rhs + Jd^T ((Sigma_s + delta)(d - s) + crb)
        const bool rhs = q >= NZ;
#pragma unroll
        for (int m = 0; m < NC; ++m) {
          const double wgt = delta + w.csig[m];
          const double ds = w.cdv[m] - w.csv[m];
          const double right = rhs ? wgt * ds + w.crb[m] : wgt * w.jq[m];
          h += w.ji[r][m] * right;
        }
      }
      o.M0[r] = h;
    }
  }
  __device__ __forceinline__ static void stage_ops(const Lds l, int k, double delta, int q, int g, StageOps& o) {
    StageRaw w;
    stage_load(l, k, q, g, w);
    stage_finish(w, delta, q, g, o);
  }

  __device__ __forceinline__ static bool backward_reg(const Lds l, int N, double delta) {
    const int t = threadIdx.x, q = t & 15, g = t >> 4;
    constexpr int KB = KB_;
    const int qx = q < NX ? q : NX - 1;
    const bool colP = q < NX, colR = q == NZ, use = colP || colR;
    const int qk = colR ? NX : qx;                 // this lane's column in the [. | rhs] outputs of pitch PP
    double Pr[KB], pr[KB];
#pragma unroll
    for (int r = 0; r < KB; ++r) {
      const int row = g + 4 * r, rc = row < NX ? row : NX - 1;
      const double v = l.P[((size_t)N * NX + rc) * PP + qk];
      Pr[r] = (row < NX && colP) ? v : 0.0;
      pr[r] = (row < NX && colR) ? v : 0.0;
    }
    StageOps o;
    stage_ops(l, N - 1, delta, q, g, o);
    bool pd = true;
    // one stage: the two chained products (operands `o`), then - with the operands of the coming stages being fetched in the
    // meantime by `fetch` - the pivot block, feedback, cost-to-go and closed-loop coefficients
    auto stage = [&](int k, auto&& fetch) __attribute__((always_inline)) {
      // T = P_{k+1} [A B | -c] + [0 | p_{k+1}];  M = [A B | -c]^T T + [H_k | r_k]
      v4d Tacc = {0.0, 0.0, 0.0, 0.0}, Macc = o.M0;
#if HILO_RIC_TADD
      // the cost-to-go vector joins the right-hand-side column after the product: the product starts from the constant zero tile
      // (no accumulator registers to clear per stage)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) Tacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pr[kb], o.b1[kb], Tacc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < KB; ++r) Tacc[r] += pr[r];
#else
#pragma unroll
      for (int r = 0; r < KB; ++r) Tacc[r] = pr[r];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) Tacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pr[kb], o.b1[kb], Tacc, 0, 0, 0);
#endif
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) Macc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.b1[kb], Tacc[kb], Macc, 0, 0, 0);
      const StageOps c = o;                           // this stage's closed-loop operands
      fetch();
      if constexpr (DUPU) {
        // own column of the input rows (duplicated rows: registers RB_ + a), the transposed partner's for the row of this lane
        double mq[NU], yq[NU], mg[KB][NU], mji[KB];
#pragma unroll
        for (int a = 0; a < NU; ++a) mq[a] = Macc[RB_ + a];
#pragma unroll
        for (int r = 0; r < KB; ++r) {
          const int i = g + 4 * r, ic = i < NX ? i : NX - 1;
          const int src = 16 * (qx % 4) + ic;          // lane (qx % 4, i): M[qx][i] in register qx / 4, M_ux[:, i] in the duplicates
#pragma unroll
          for (int a = 0; a < NU; ++a) mg[r][a] = __shfl(Macc[RB_ + a], src);
          mji[r] = 0.0;
#pragma unroll
          for (int rr = 0; rr < KB; ++rr) {
            const double v = __shfl(Macc[rr], src);
            mji[r] = (KB == 1 || qx / 4 == rr) ? v : mji[r];
          }
        }
        // pivot block out of the duplicated rows of lanes NX + c (scalar registers), inverted in closed form
        const double r00 = read_lane(Macc[RB_], NX);
        PivotInv pv;
        if constexpr (NU == 2) pv = pivot_inverse(r00, read_lane(Macc[RB_ + 1], NX), read_lane(Macc[RB_ + 1], NX + 1));
        else pv = pivot_inverse(r00, 0.0, 1.0);
        pd = pd && pv.ok;
        double tq = 0.0;
        if constexpr (NU == 2) {
          tq = fma(-pv.l, mq[0], mq[1]);
          yq[1] = tq * pv.i1;
          yq[0] = fma(-pv.l, yq[1], mq[0] * pv.i0);
        } else yq[0] = mq[0] * pv.i0;
        if (g == 0 && use) {   // feedback K[:, q] = -y_q, feed-forward kff = -y_NZ
#pragma unroll
          for (int a = 0; a < NU; ++a) l.Kg[((size_t)k * NU + a) * PP + qk] = -yq[a];
        }
#pragma unroll
        for (int r = 0; r < KB; ++r) {
          const int i = g + 4 * r, ic = i < NX ? i : NX - 1;
          const bool valid = NX % 4 == 0 || i < NX;
          const double sym = colR ? Macc[r] : 0.5 * (Macc[r] + mji[r]);
          double sv = fma(-(mg[r][0] * mq[0]), pv.i0, sym);      // every product: the same bits in the lanes (i, q) and (q, i)
          if constexpr (NU == 2) sv = fma(-(fma(-pv.l, mg[r][0], mg[r][1]) * tq), pv.i1, sv);
          double cl = c.b1[r];                          // A[i][q] (q < NX) or -c[i] (q = NZ)
#pragma unroll
          for (int a = 0; a < NU; ++a) cl -= c.bi[r][a] * yq[a];
          Pr[r] = (valid && colP) ? sv : 0.0;
          pr[r] = (valid && colR) ? sv : 0.0;
          if (valid && use) {
            l.P[((size_t)k * NX + ic) * PP + qk] = sv;
            l.Acl[((size_t)k * NX + ic) * PP + qk] = cl;
          }
        }
        return;
      }
      // cross-lane fetches of the M_ux columns and the mirrored M_xx entries, all issued before they are needed
      double wq[NU], yq[NU], wi[KB][NU], mji[KB];
#pragma unroll
      for (int a = 0; a < NU; ++a) wq[a] = __shfl(Macc[(NX + a) / 4], 16 * ((NX + a) % 4) + q);
#pragma unroll
      for (int r = 0; r < KB; ++r) {
        const int i = g + 4 * r, ic = i < NX ? i : NX - 1;
#pragma unroll
        for (int a = 0; a < NU; ++a) wi[r][a] = __shfl(Macc[(NX + a) / 4], 16 * ((NX + a) % 4) + ic);
        mji[r] = 0.0;   // M[q][i]: register q / 4 of lane 16 (q % 4) + i
#pragma unroll
        for (int rr = 0; rr < KB; ++rr) {
          const double v = __shfl(Macc[rr], 16 * (qx % 4) + ic);
          mji[r] = (KB == 1 || qx / 4 == rr) ? v : mji[r];
        }
      }
      // reduced pivot block R_k = M_uu out of the accumulators, factored redundantly per lane
      double Rl[NU * NU], Lc[NU * NU], invd[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a)
#pragma unroll
        for (int c2 = 0; c2 <= a; ++c2) {
          const int i = NX + a, j = NX + c2;
          Rl[a * NU + c2] = read_lane(Macc[i / 4], 16 * (i % 4) + j);
        }
      pd = small_chol_reg<NU>(Rl, Lc, invd) && pd;
      // this lane's column: w_q = L^-1 M_ux[:, q] (column NZ: m_u), y_q = L^-T w_q
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        double sv = wq[a];
#pragma unroll
        for (int c2 = 0; c2 < a; ++c2) sv -= Lc[a * NU + c2] * wq[c2];
        wq[a] = sv * invd[a];
      }
#pragma unroll
      for (int a = NU - 1; a >= 0; --a) {
        double sv = wq[a];
#pragma unroll
        for (int c2 = a + 1; c2 < NU; ++c2) sv -= Lc[c2 * NU + a] * yq[c2];
        yq[a] = sv * invd[a];
      }
      if (g == 0 && use) {   // feedback K[:, q] = -y_q, feed-forward kff = -y_NZ
#pragma unroll
        for (int a = 0; a < NU; ++a) l.Kg[((size_t)k * NU + a) * PP + qk] = -yq[a];
      }
#pragma unroll
      for (int r = 0; r < KB; ++r) {
        const int i = g + 4 * r, ic = i < NX ? i : NX - 1;
        const bool valid = NX % 4 == 0 || i < NX;
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          double sv = wi[r][a];
#pragma unroll
          for (int c2 = 0; c2 < a; ++c2) sv -= Lc[a * NU + c2] * wi[r][c2];
          wi[r][a] = sv * invd[a];
        }
        double sv = colR ? Macc[r] : 0.5 * (Macc[r] + mji[r]);
#pragma unroll
        for (int a = 0; a < NU; ++a) sv -= wi[r][a] * wq[a];
        double cl = c.b1[r];                          // A[i][q] (q < NX) or -c[i] (q = NZ)
#pragma unroll
        for (int a = 0; a < NU; ++a) cl -= c.bi[r][a] * yq[a];
        Pr[r] = (valid && colP) ? sv : 0.0;
        pr[r] = (valid && colR) ? sv : 0.0;
        if (valid && use) {
          l.P[((size_t)k * NX + ic) * PP + qk] = sv;
          l.Acl[((size_t)k * NX + ic) * PP + qk] = cl;
        }
      }
    };
    if constexpr (BIG) {
      // workspace mode: raw operands two stages ahead (ping-pong buffers, no register moves), finished one stage ahead
      StageRaw ra, rb;
      stage_load(l, N >= 2 ? N - 2 : 0, q, g, ra);
      int k = N - 1;
      for (; k >= 1; k -= 2) {
        stage(k, [&]() __attribute__((always_inline)) { stage_load(l, k >= 2 ? k - 2 : 0, q, g, rb); stage_finish(ra, delta, q, g, o); });
        stage(k - 1, [&]() __attribute__((always_inline)) { stage_load(l, k >= 3 ? k - 3 : 0, q, g, ra); stage_finish(rb, delta, q, g, o); });
      }
      if (k == 0) stage(0, [&]() __attribute__((always_inline)) {});
    } else {
      for (int k = N - 1; k >= 0; --k)
        stage(k, [&]() __attribute__((always_inline)) { stage_ops(l, k > 0 ? k - 1 : 0, delta, q, g, o); });   // next stage's operands: in flight during the factorisation
    }
    return uni(pd);
  }

  // an LDS pointer the compiler must take as it is (it would otherwise split it into a uniform base and a lane offset and add the
  // two at every access)
  template <class P>
  __device__ __forceinline__ static void opaque_lds(P& ptr) {
    unsigned a = (unsigned)(size_t)ptr;
    asm volatile("" : "+v"(a));
    ptr = (P)(size_t)a;
  }

  // ---- the same recursion, software-pipelined by hand (DUPU policies, iterate in LDS) ------------------------------------------
  // One wave per SIMD issues one instruction per four clocks whatever it is, and each of the two f64 products occupies the matrix
  // core for 16 passes during which only INDEPENDENT instructions can issue.  The stage is therefore laid out in source order, with
  // scheduling barriers, as:  product 1 | outputs of the PREVIOUS stage (feedback, closed-loop row, cost-to-go stores) and the raw
  // loads of the NEXT stage | product 2 | operands of the next stage finished | pivot block, reciprocal, this lane's entry of P_k.
  // What lies on the recursion's critical path is the two products, six v_readlane, one reciprocal and about ten multiply-adds.
#ifndef HILO_RIC_PIPE
#define HILO_RIC_PIPE 1
#endif
  // D0: no inertia correction (delta = 0: every factorisation of a converging solve) - the next stage's tile is then the loaded
  // [H + Sigma | r] as it is.  Nothing in the shadow of the second product may be an f64 vector instruction: the f64 products run on
  // the same arithmetic units, and the first such instruction (the closed-loop row's multiply-adds and these diagonal adds in the
  // first version: 120 clocks per stage, measured) waits for the whole product, and with it everything behind it.
  template <bool D0>
  __device__ __forceinline__ static bool backward_dup(const Lds l, int N, double delta) {
    static_assert(DUPU && !BIG, "backward_dup: duplicated-row policies with the iterate in LDS");
    const int t = threadIdx.x, q = t & 15, g = t >> 4;
    constexpr int KB = KB_;
    const int qx = q < NX ? q : NX - 1;
    const bool colP = q < NX, colR = q == NZ;
    const bool use = ((((1u << NX) - 1u) | (1u << NZ)) >> q) & 1u;   // colP or colR, as ONE lane predicate
    const int qk = colR ? NX : qx;                 // this lane's column in the [. | rhs] outputs of pitch PP
    const int qc = q < NZ ? q : NZ;
    int qa = qc;                                   // column of [A B | -c] this lane feeds to the products (see stage_load)
    if (q >= 4 * RB_) qa = ((q - 4 * RB_) >> 2) < NU ? NX + ((q - 4 * RB_) >> 2) : NZ;
    double Pr[KB], pr[KB], dl[RD_];
    int src[KB], srcm[KB];
#pragma unroll
    for (int r = 0; r < KB; ++r) {
      const int row = g + 4 * r, rc = row < NX ? row : NX - 1;
      const double v = l.P[((size_t)N * NX + rc) * PP + qk];
      Pr[r] = (row < NX && colP) ? v : 0.0;
      pr[r] = (row < NX && colR) ? v : 0.0;
      src[r] = 16 * (qx % 4) + rc;                 // lane (qx % 4, row): M[qx][row] in register qx / 4, M_ux[:, row] in the duplicates
      srcm[r] = colP ? src[r] : t;                 // the right-hand-side column has no mirrored entry: its own (0.5 (m + m) = m exactly)
    }
#pragma unroll
    for (int r = 0; r < RD_; ++r) dl[r] = stage_row(r, g) == q ? delta : 0.0;   // the inertia correction on this lane's diagonal entries
    // running pointers of what a lane reads and writes per stage (one decrement each per stage, immediate offsets otherwise).
    // The loads run one stage ahead: at k = 0 they touch the doubles in front of the arrays - inside the iterate, never used.
    cdp pB1[KB], pBi[KB], pW[RD_];
    dp pP[KB], pAcl[KB], pKg;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int kk = 4 * kb + g, kc = (NX % 4 == 0 || kk < NX) ? kk : NX - 1;
      pB1[kb] = l.AB + ((size_t)(N - 1) * NX + kc) * ABP + qa;
      pBi[kb] = l.AB + ((size_t)(N - 1) * NX + kc) * ABP + NX;
      pP[kb] = l.P + ((size_t)(N - 1) * NX + kc) * PP + qk;
      pAcl[kb] = l.Acl + ((size_t)(N - 1) * NX + kc) * PP + qk;
    }
#pragma unroll
    for (int r = 0; r < RD_; ++r) pW[r] = l.W + ((size_t)(N - 1) * NZ + stage_row(r, g)) * WP + qc;
    pKg = l.Kg + (size_t)(N - 1) * NU * PP + qk;
    // (the compiler would split every pointer into a uniform base and a lane offset and add the two at every access)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) { opaque_lds(pB1[kb]); opaque_lds(pBi[kb]); opaque_lds(pP[kb]); opaque_lds(pAcl[kb]); }
#pragma unroll
    for (int r = 0; r < RD_; ++r) opaque_lds(pW[r]);
    opaque_lds(pKg);
    StageOps oA, oB;      // operands of the current and of the next stage, roles alternating (loop unrolled by two: no register copies)
    StageRaw raw;
    auto load = [&]() __attribute__((always_inline)) {   // raw operands of the stage the pointers stand on, then one stage down
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        raw.b1[kb] = pB1[kb][0];
#pragma unroll
        for (int a = 0; a < NU; ++a) raw.bi[kb][a] = pBi[kb][a];
        pB1[kb] -= NX * ABP;
        pBi[kb] -= NX * ABP;
      }
#pragma unroll
      for (int r = 0; r < RD_; ++r) {
        raw.w[r] = pW[r][0];
        pW[r] -= NZ * WP;
      }
    };
    auto finish = [&](StageOps& o, int kn) __attribute__((always_inline)) {
      if constexpr (NC > 0) {
        stage_load(l, kn, q, g, raw);   // (the rows of the inequality block; policies with rows take the general routine)
        stage_finish(raw, delta, q, g, o);
      } else {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const int kk = 4 * kb + g;
          o.b1[kb] = (NX % 4 == 0 || kk < NX) ? raw.b1[kb] : 0.0;
#pragma unroll
          for (int a = 0; a < NU; ++a) o.bi[kb][a] = raw.bi[kb][a];
        }
        o.M0 = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < RD_; ++r) o.M0[r] = D0 ? raw.w[r] : raw.w[r] + dl[r];
      }
    };
    bool pd = true;
    // outputs of a finished stage (with the operands `po` it ran on), stored during the next stage's second product
    double d_ny[NU], d_sv[KB], d_cl[KB];
#pragma unroll
    for (int a = 0; a < NU; ++a) d_ny[a] = 0.0;
#pragma unroll
    for (int r = 0; r < KB; ++r) { d_sv[r] = 0.0; d_cl[r] = 0.0; }
    auto emit = [&]() __attribute__((always_inline)) {
      // feedback K[:, q] = -y_q (kff = -y_NZ): every row group writes the same value; closed-loop row and cost-to-go entry
      if (use) {
#pragma unroll
        for (int a = 0; a < NU; ++a) pKg[a * PP] = d_ny[a];
#pragma unroll
        for (int r = 0; r < KB; ++r) {
          if (NX % 4 == 0 || g + 4 * r < NX) {
            pP[r][0] = d_sv[r];
            pAcl[r][0] = d_cl[r];
          }
        }
      }
      pKg -= NU * PP;
#pragma unroll
      for (int r = 0; r < KB; ++r) { pP[r] -= NX * PP; pAcl[r] -= NX * PP; }
    };
    // one stage on the operands `o`; `on` holds the previous stage's operands on entry (for its outputs) and the next stage's on exit
    auto stage = [&](int k, const StageOps& o, StageOps& on, bool have) __attribute__((always_inline)) {
      // T = P_{k+1} [A B | -c] (+ [0 | p_{k+1}] below);  M = [A B | -c]^T T + [H_k | r_k]
      v4d Tacc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) Tacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Pr[kb], o.b1[kb], Tacc, 0, 0, 0);
#if HILO_RIC_PIPE
      __builtin_amdgcn_sched_barrier(0);
#endif
      load();
#if HILO_RIC_PIPE
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(Tacc));   // the whole tile stays allocated while the product runs (no register of it is reused in its shadow)
#endif
#pragma unroll
      for (int r = 0; r < KB; ++r) Tacc[r] += pr[r];
      v4d Macc = o.M0;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) Macc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.b1[kb], Tacc[kb], Macc, 0, 0, 0);
#if HILO_RIC_PIPE
      __builtin_amdgcn_sched_barrier(0);
#endif
      if (have) emit();
      finish(on, k > 0 ? k - 1 : 0);
#if HILO_RIC_PIPE
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(Macc));   // (as above: rows of the tile nothing reads would otherwise be handed out as temporaries of the shadow)
#endif
      // own column of the input rows (registers RB_ + a), the transposed partner's for the row(s) of this lane
      double mq[NU], mg[KB][NU], mji[KB];
#pragma unroll
      for (int a = 0; a < NU; ++a) mq[a] = Macc[RB_ + a];
#pragma unroll
      for (int r = 0; r < KB; ++r) {
#pragma unroll
        for (int a = 0; a < NU; ++a) mg[r][a] = __shfl(Macc[RB_ + a], src[r]);
        mji[r] = 0.0;
#pragma unroll
        for (int rr = 0; rr < KB; ++rr) {
          const double v = __shfl(Macc[rr], srcm[r]);
          mji[r] = (KB == 1 || (colP ? qx / 4 : r) == rr) ? v : mji[r];
        }
      }
      const double r00 = read_lane(Macc[RB_], NX);
      PivotInv pv;
      if constexpr (NU == 2) pv = pivot_inverse(r00, read_lane(Macc[RB_ + 1], NX), read_lane(Macc[RB_ + 1], NX + 1));
      else pv = pivot_inverse(r00, 0.0, 1.0);
      pd = pd && pv.ok;           // (a pivot block that is not positive definite: the factorisation is abandoned after the loop)
      double tq = 0.0;
      if constexpr (NU == 2) {    // -y_q = -R^-1 m_q by the two substitutions
        tq = fma(-pv.l, mq[0], mq[1]);
        d_ny[1] = -(tq * pv.i1);
        d_ny[0] = fma(-pv.l, d_ny[1], -(mq[0] * pv.i0));
      } else d_ny[0] = -(mq[0] * pv.i0);
#pragma unroll
      for (int r = 0; r < KB; ++r) {
        const int i = g + 4 * r;
        const bool valid = NX % 4 == 0 || i < NX;
        const double sym = 0.5 * (Macc[r] + mji[r]);
        double sv = fma(-(mg[r][0] * mq[0]), pv.i0, sym);        // every product: the same bits in the lanes (i, q) and (q, i)
        if constexpr (NU == 2) sv = fma(-(fma(-pv.l, mg[r][0], mg[r][1]) * tq), pv.i1, sv);
        d_sv[r] = sv;
        double cl = o.b1[r];                            // closed-loop row: A[i][q] (q < NX) or -c[i] (q = NZ), + B[i] (-y_q)
#pragma unroll
        for (int a = 0; a < NU; ++a) cl = fma(o.bi[r][a], d_ny[a], cl);
        d_cl[r] = cl;
        // rows of T beyond NX are read by nothing when NX fills its blocks of four: lanes q >= NX may keep their (finite) value
        Pr[r] = (NX % 4 == 0) ? sv : ((valid && colP) ? sv : 0.0);
        pr[r] = (valid && colR) ? sv : 0.0;
      }
    };
    load();
    int k = N - 1;
    bool have = false;      // a finished stage's outputs are waiting to be stored
    if (N & 1) {            // odd horizon: one stage in front of the pairs
      finish(oB, k);
      stage(k, oB, oA, false);
      --k;
      have = true;
    } else finish(oA, k);
    for (; k >= 1; k -= 2) {
      stage(k, oA, oB, have);
      stage(k - 1, oB, oA, true);
      have = true;
    }
    emit();
    return uni(pd);
  }

  // broadcast of lane j's value inside every quad of lanes (DPP quad_perm [j, j, j, j]): the forward sweep's state exchange
  template <int J>
  __device__ __forceinline__ static double quad_bcast(double v) { return dpp_mov<J * 0x55>(v); }

  // ---- Riccati factor + solve of the Newton system; false when a reduced pivot is not positive ---------------
  // Per stage: (1) M = H_k + [A B]^T P_{k+1} [A B] and its right-hand side on the f64 matrix cores, the reduced pivot
  // block broadcast from the accumulators and factored (redundantly per lane); (2) feedback K, feed-forward kff, P_k, p_k and
  // the closed-loop coefficients.  The forward sweep keeps dx in registers (DPP / v_readlane broadcasts, no LDS round trip).
  // The stage matrices [H_k + Sigma | r_k] were completed by kkt_pass / finish_rhs; `delta` is added to their diagonal here.
  // `resto`: feasibility-restoration step (the caller has replaced the blocks by the barrier diagonal and passes delta = 1: H = I,
  // least-norm d with J d = -c); here it only switches off the terminal cost's Hessian and the new row multipliers
  __device__ OCP_PHASE static bool riccati(lds_double* lbase, double* ws, double mu, double delta, bool resto = false) {
    lbase = uni(lbase); ws = uni(ws); mu = uni(mu); delta = uni(delta); resto = uni(resto);
    const Lds l = carve(lbase, ws);
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N, t = threadIdx.x;
    (void)mu;
    DTICK0
    // terminal: P_N = hess V + Sigma + delta, p_N = grad V + barrier rhs; the column -c of the padded [A B | -c] (the defects
    // change between factorisations of one iteration: second-order correction)
    OCP_FOR(e, NX * PP) {
      const int i = e / PP, j = e - i * PP;
      double v;
      if (j < NX) {
        v = 0.0;
        if (!resto) {
          cdp Q = qd_term(l, N);
          if (i == j) v = Q[i];
          else {
            const int a = i < j ? i : j, b = i < j ? j : i;
            v = 0.5 * (Q[dir_of(a, b, NX)] - Q[a] - Q[b]);
          }
        }
        if (i == j) v += delta + l.sig[N * NZ + i];
      } else v = l.rbN[i];
      l.P[((size_t)N * NX) * PP + e] = v;
    }
    OCP_FOR(e, N * NX) l.AB[(size_t)e * ABP + NZ] = -l.c[e];
    __syncthreads();
#ifndef HILO_RICCATI_LDS
    if constexpr (MFMA_STAGE && NU > 0 && NH == 0) {
      bool okb;
      if constexpr (DUPU && !BIG) okb = delta == 0.0 ? backward_dup<true>(l, N, delta) : backward_dup<false>(l, N, delta);
      else okb = backward_reg(l, N, delta);
      if (!okb) return false;
      __syncthreads();
      DTICK(8)
    } else
#endif
    for (int k = N - 1; k >= 0; --k) {
      cdp Pn = l.P + (size_t)(k + 1) * NX * PP;       // [P_{k+1} | p_{k+1}]
      cdp AB = l.AB + (size_t)k * NX * ABP;
      cdp Wk = l.W + (size_t)k * NZ * WP;
      double Lc[NU > 0 ? NU * NU : 1], invd[NU > 0 ? NU : 1];   // Cholesky factor of the reduced pivot block R_k
      bool pd = true, factored = false;
      // (1) Mm = H_k + [A B]^T P_{k+1} [A B];  mm = r_k + [A B]^T (p_{k+1} - P_{k+1} c_k).
#ifndef HILO_RICCATI_VALU
      if constexpr (MFMA_STAGE) {
        // Two chained v_mfma_f64_16x16x4 per 4 rows of the inner dimension, ONE matrix element per lane (lane = 16 g + q):
        //   T = P_{k+1} [A B | -c] + [0 | p_{k+1}]      A-operand P[q][4kb+g], B-operand [A B | -c][4kb+g][q]
        //   M = [A B | -c]^T T + [H_k | r_k]             A-operand = the same register, B-operand T rows 4kb+g = accumulator kb of T
        // The accumulator of the f64 form holds rows g + 4r, column q in register r - exactly the B-operand layout of the
        // second product, so T never leaves the registers.
        const int q = t & 15, g = t >> 4;
        constexpr int KB = KB_, RB = RB_;
        const int qx = q < NX ? q : NX - 1, qc = q < NZ ? q : NZ;
        v4d Tacc = {0.0, 0.0, 0.0, 0.0}, Macc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < KB; ++r) {
          const int row = g + 4 * r, rc = row < NX ? row : NX - 1;
          const double pv_ = Pn[rc * PP + NX];
          Tacc[r] = (q == NZ && row < NX) ? pv_ : 0.0;
        }
        double a1[KB], b1[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const int kk = 4 * kb + g;
          const bool kv = kk < NX;
          const int kc = kv ? kk : NX - 1;
          const double pe = Pn[qx * PP + kc], ab = AB[kc * ABP + qc];
          a1[kb] = (kv && q < NX) ? pe : 0.0;
          b1[kb] = kv ? ab : 0.0;
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int i = g + 4 * r;
          const bool iv = i < NZ, rhs = q >= NZ;
          const int ic = iv ? i : NZ - 1, jj = q < NZ ? q : 0;
          double h = Wk[ic * WP + qc];
          if (ic == q) h += delta;
          if constexpr (NC > 0) {  // eliminated slack rows: + Jd^T (Sigma_s + delta) Jd, rhs + Jd^T ((Sigma_s + delta)(d - s) + crb)
#pragma unroll
            for (int m = 0; m < NC; ++m) {
              const int rr = k * NC + m;
              const double wgt = delta + l.csig[rr];
              const double ds = l.cd[rr] - l.cs[rr], cr = l.crb[rr], jr = l.Jd[rr * NZ + jj];
              const double right = rhs ? wgt * ds + cr : wgt * jr;
              h += l.Jd[rr * NZ + ic] * right;
            }
          }
          Macc[r] = h;
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) Tacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kb], b1[kb], Tacc, 0, 0, 0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) Macc = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[kb], Tacc[kb], Macc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int i = g + 4 * r;
          if (i < NZ && q <= NZ) {
            lds_double* dst = q == NZ ? l.mm + i : l.Mm + i * NZ + q;
            *dst = Macc[r];
          }
        }
        if constexpr (NU > 0) {
          // the pivot block R_k = M_uu straight out of the accumulators (entry (i, j) sits in register i / 4 of lane
          // 16 (i % 4) + j): v_readlane broadcasts instead of waiting for the LDS round trip of Mm
          double Rl[NU * NU];
#pragma unroll
          for (int a = 0; a < NU; ++a)
#pragma unroll
            for (int c2 = 0; c2 <= a; ++c2) {
              const int i = NX + a, j = NX + c2;
              Rl[a * NU + c2] = read_lane(Macc[i / 4], 16 * (i % 4) + j);
            }
          pd = small_chol_reg<NU>(Rl, Lc, invd);
          factored = true;
        }
      } else
#endif
      {
      // One uniform code path for the NZ x (NZ+1) entries: column NZ is the right-hand side, i.e. the column -c_k
      // of [A B | -c] with p_{k+1} added.  All operands are fetched before the arithmetic (one LDS wait).
      OCP_FOR(e, NZ * (NZ + 1)) {
        const int i = e / (NZ + 1), j = e - i * (NZ + 1);
        const bool rhs = j == NZ;
        const int jj = rhs ? 0 : j;
        double Pl[NX * NX], ai[NX], aj[NX], pl[NX];
#pragma unroll
        for (int q = 0; q < NX * NX; ++q) Pl[q] = Pn[(q / NX) * PP + q % NX];
#pragma unroll
        for (int n = 0; n < NX; ++n) {
          ai[n] = AB[n * ABP + i];
          aj[n] = AB[n * ABP + j];       // column NZ: -c
          pl[n] = Pn[n * PP + NX];
        }
        double s = Wk[i * WP + j];       // column NZ: r_k
        const double dg = (i == j) ? delta : 0.0;
        if constexpr (NC > 0) {  // eliminated slack rows: + Jd^T (Sigma_s + delta) Jd, rhs + Jd^T ((Sigma_s + delta)(d - s) + crb)
#pragma unroll
          for (int m = 0; m < NC; ++m) {
            const int r = k * NC + m;
            const double wgt = delta + l.csig[r];
            const double ds = l.cd[r] - l.cs[r], cr = l.crb[r], jr = l.Jd[r * NZ + jj];
            const double right = rhs ? wgt * ds + cr : wgt * jr;
            s += l.Jd[r * NZ + i] * right;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < NX; ++m) {
          double tm = rhs ? pl[m] : 0.0;
#pragma unroll
          for (int n = 0; n < NX; ++n) tm += Pl[m * NX + n] * aj[n];
          s += ai[m] * tm;
        }
        lds_double* dst = rhs ? l.mm + i : l.Mm + i * NZ + j;
        *dst = s + dg;   // dg = 0 in the right-hand-side column (i != NZ)
      }
      }
      if constexpr (NH > 0) {
        // beyond the control horizon the inputs of the stage are not variables (mpc.py:1629-1630): their rows / columns of
        // the stage matrix are replaced by the identity and a zero right-hand side, so K = 0, kff = 0, P_k = M_xx
        if (k >= pc.Nc) {
          __syncthreads();
          OCP_FOR(e, NU * (NZ + 1)) {
            const int a = e / (NZ + 1), j = e - a * (NZ + 1);
            if (j == NZ) l.mm[NX + a] = 0.0;
            else {
              l.Mm[(NX + a) * NZ + j] = (j == NX + a) ? 1.0 : 0.0;
              l.Mm[j * NZ + NX + a] = (j == NX + a) ? 1.0 : 0.0;
            }
          }
          factored = false;   // phase (2) factors the identity block from LDS
        }
      }
      __syncthreads();
      // (2) pivot block (factored redundantly per lane), feedback, cost-to-go: lane (i, j), j = 0..NX:
      //   y_j = R^-1 M_ux[:, j] (j < NX) or R^-1 m_u (j = NX);  P_k[i][j] = sym(M_xx)[i][j] - M_xu[i] y_j;
      //   p_k[i] = m_x[i] - M_xu[i] y_NX;  lanes with i = 0 also store K[:, j] = -y_j and kff = -y_NX
      if constexpr (NU > 0) {
        if (!factored) pd = small_chol<NU>(l.Mm + NX * NZ + NX, NZ, Lc, invd);
        if (!pd) return false;  // wave-uniform: every lane factors the same block
        OCP_FOR(e, NX * (NX + 1)) {
          const int i = e / (NX + 1), j = e - i * (NX + 1);
          const bool rhs = j == NX;
          const int jj = rhs ? 0 : j;
          double y[NU], xu[NU];
#pragma unroll
          for (int a = 0; a < NU; ++a) {
            const double ym = l.mm[NX + a], yM = l.Mm[(NX + a) * NZ + jj];
            y[a] = rhs ? ym : yM;
            xu[a] = l.Mm[(NX + a) * NZ + i];   // M_ux^T: the same block on both sides of R^-1
          }
          const double sm = l.mm[i], s1 = l.Mm[i * NZ + jj], s2 = l.Mm[jj * NZ + i];
          double s = rhs ? sm : 0.5 * (s1 + s2);
          // closed-loop coefficient of the forward sweep, same lane: Acl[i][j] = A[i][j] + B[i] K[:, j], bcl[i] = B[i] kff - c[i]
          // (K[:, j] = -y_j, kff = -y_NX are this lane's solve)
          double bi[NU];
#pragma unroll
          for (int a = 0; a < NU; ++a) bi[a] = AB[i * ABP + NX + a];
          double cl = AB[i * ABP + (rhs ? NZ : jj)];
          __builtin_amdgcn_sched_barrier(0);
          small_solve<NU>(Lc, invd, y);
#pragma unroll
          for (int a = 0; a < NU; ++a) s -= xu[a] * y[a];
          // P_k is kept EXACTLY symmetric: entry (i, j), i <= j, is computed once and stored twice.  Rounding-level
          // asymmetry is not damped by the recursion - for unstable dynamics it grows like the open loop and destroyed the
          // pivots after ~38 stages of the chemostat.
#pragma unroll
          for (int a = 0; a < NU; ++a) cl -= bi[a] * y[a];
          l.Acl[((size_t)k * NX + i) * PP + j] = cl;
          if (rhs) l.P[((size_t)k * NX + i) * PP + NX] = s;
          else if (i <= jj) {
            l.P[((size_t)k * NX + i) * PP + jj] = s;
            l.P[((size_t)k * NX + jj) * PP + i] = s;
          }
          if (i == 0) {
#pragma unroll
            for (int a = 0; a < NU; ++a) l.Kg[((size_t)k * NU + a) * PP + j] = -y[a];
          }
        }
      } else {
        OCP_FOR(e, NX * PP) {
          const int i = e / PP, j = e - i * PP;
          l.P[((size_t)k * NX) * PP + e] = j < NX ? 0.5 * (l.Mm[i * NZ + j] + l.Mm[j * NZ + i]) : l.mm[i];
        }
      }
      __syncthreads();
    }
    // initial state: pinned (dx_0 = 0) or (partly) free: dx_0 = -P_0^-1 p_0 on the free slots (P_0 must be positive
    // definite there); pinned rows / columns of P_0 are replaced by the identity
    if (FIX_X0 && pc.x0_free_mask == 0u) {
      OCP_FOR(i, NX) l.D[i] = 0.0;
    } else {
      OCP_FOR(e, NX * NX) {
        const int i = e / NX, j = e - i * NX;
        const bool pin = x0_pinned(pc, i) || x0_pinned(pc, j);
        l.Mm[e] = pin ? (i == j ? 1.0 : 0.0) : l.P[i * PP + j];
      }
      __syncthreads();
      double Lc[NX * NX], invd[NX], y[NX];
      const bool pd = small_chol<NX>(l.Mm, NX, Lc, invd);
      if (!pd) return false;
#pragma unroll
      for (int a = 0; a < NX; ++a) y[a] = x0_pinned(pc, a) ? 0.0 : l.P[a * PP + NX];
      small_solve<NX>(Lc, invd, y);
      __syncthreads();
      if (t == 0) {
#pragma unroll
        for (int a = 0; a < NX; ++a) l.D[a] = -y[a];
      }
    }
    DTICK(9)
    // closed-loop matrices for the forward sweep: Acl = A + B K, bcl = B kff - c: written by the stage loop above when there
    // are inputs; without inputs Acl = A, bcl = -c
    if constexpr (NU == 0) OCP_FOR(e, N * NX * PP) {
      const int row = e / PP, j = e - row * PP;
      l.Acl[e] = l.AB[(size_t)row * ABP + (j < NX ? j : NZ)];
    }
    __syncthreads();
    // forward sweep on the first wave: lane i < NX carries dx[i] in a register, exchanged by DPP quad broadcasts (NX <= 4) or
    // v_readlane; the coefficients of the next stage are fetched while the current one is computed
#if HILO_FWD_SGPR
    if (t < 64) {
      // Round 5: the state step lives in SCALAR registers.  Lane i < NX holds row i of [Acl | bcl] of a chunk of FU stages and computes
      // dx_{k+1}[i]; v_readlane hands the NX results to every lane as scalar operands of the next stage's multiply-adds (no DPP
      // moves, no LDS trip).  Two chunk buffers with alternating roles: the next chunk's rows are in flight while the current one
      // is applied, and no register copies between chunks.  Lanes >= NX duplicate row 0 (same value to the same address).
      constexpr int FU = 4;
      const int i = t < NX ? t : 0;
      double dxs[NX];
      {
        const double d0 = l.D[i];
#pragma unroll
        for (int j = 0; j < NX; ++j) dxs[j] = read_lane(d0, j);
      }
      double ca[FU][NX + 1], cb[FU][NX + 1];
      auto fetch = [&](double (&dst)[FU][NX + 1], int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < FU; ++q) {
          const int kq = k0 + q < N ? k0 + q : N - 1;
          cdp row = l.Acl + ((size_t)kq * NX + i) * PP;
#pragma unroll
          for (int j = 0; j <= NX; ++j) dst[q][j] = row[j];
        }
      };
      // (whole chunks run without a test per stage - `tail` is wave-uniform: only the last chunk of a horizon that is no multiple
      // of FU looks at the stage index)
      auto run = [&](const double (&c)[FU][NX + 1], int k0) __attribute__((always_inline)) {
        auto one = [&](int q, int k) __attribute__((always_inline)) {
          double s = c[q][NX], s2 = 0.0;   // two accumulation chains of half the length
#pragma unroll
          for (int j = 0; j < NX; ++j) {
            if (j & 1) s2 = fma(c[q][j], dxs[j], s2);
            else s = fma(c[q][j], dxs[j], s);
          }
          s += s2;
          l.D[(k + 1) * NZ + i] = s;
#pragma unroll
          for (int j = 0; j < NX; ++j) dxs[j] = read_lane(s, j);
        };
        if (k0 + FU <= N) {
#pragma unroll
          for (int q = 0; q < FU; ++q) one(q, k0 + q);
        } else {
#pragma unroll
          for (int q = 0; q < FU; ++q)
            if (k0 + q < N) one(q, k0 + q);
        }
      };
      fetch(ca, 0);
      for (int k0 = 0; k0 < N; k0 += 2 * FU) {
        fetch(cb, k0 + FU);
        run(ca, k0);
        fetch(ca, k0 + 2 * FU);
        run(cb, k0 + FU);
      }
    }
#else
    if (t < 64) {
      // chunks of FU stages: the coefficients of the NEXT chunk are fetched while the current one is computed (one LDS
      // latency per chunk instead of one per stage - a stage is a handful of dependent multiply-adds)
      constexpr int FU = 4;
      const int i = t < NX ? t : 0;
      double dxi = l.D[i];
      double cur[FU][NX + 1], nxt[FU][NX + 1];
      auto fetch = [&](double (*dst)[NX + 1], int k0) {
#pragma unroll
        for (int q = 0; q < FU; ++q) {
          const int kq = k0 + q < N ? k0 + q : N - 1;
          cdp row = l.Acl + ((size_t)kq * NX + i) * PP;
#pragma unroll
          for (int j = 0; j <= NX; ++j) dst[q][j] = row[j];
        }
      };
      fetch(cur, 0);
      for (int k0 = 0; k0 < N; k0 += FU) {
        fetch(nxt, k0 + FU);
#pragma unroll
        for (int q = 0; q < FU; ++q) {
          const int k = k0 + q;
          double s = cur[q][NX];
          if constexpr (NX <= 4) {
            double dj[4];
            dj[0] = quad_bcast<0>(dxi); dj[1] = quad_bcast<1>(dxi); dj[2] = quad_bcast<2>(dxi); dj[3] = quad_bcast<3>(dxi);
            double s2 = 0.0;   // two accumulation chains of half the length
#pragma unroll
            for (int j = 0; j < NX; ++j) {
              if (j & 1) s2 += cur[q][j] * dj[j];
              else s += cur[q][j] * dj[j];
            }
            s += s2;
          } else {
#pragma unroll
            for (int j = 0; j < NX; ++j) s += cur[q][j] * read_lane(dxi, j);   // v_readlane: scalar broadcast, no LDS crossbar trip
          }
          if (k < N) {
            dxi = s;
            if (t < NX) l.D[(k + 1) * NZ + i] = dxi;
          }
        }
#pragma unroll
        for (int q = 0; q < FU; ++q)
#pragma unroll
          for (int j = 0; j <= NX; ++j) cur[q][j] = nxt[q][j];
      }
    }
#endif
    __syncthreads();
    DTICK(10)
    // inputs and new equality multipliers, parallel over stages:
    //   du_k = K dx_k + kff,   lam_{k+1} = -(P_{k+1} dx_{k+1} + p_{k+1})
    OCP_FOR(e, N * (NU + NX)) {
      const int k = e / (NU + NX), r = e - k * (NU + NX);
      if (r < NU) {
        cdp Kr = l.Kg + ((size_t)k * NU + r) * PP;
        double s = Kr[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) s += Kr[j] * l.D[k * NZ + j];
        l.D[k * NZ + NX + r] = s;
      } else {
        const int i = r - NU;
        cdp Prow = l.P + ((size_t)(k + 1) * NX + i) * PP;
        double s = Prow[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) s += Prow[j] * l.D[(k + 1) * NZ + j];
        l.lamn[k * NX + i] = -s;
      }
    }
    OCP_FOR(a, NU) l.D[N * NZ + NX + a] = 0.0;
    __syncthreads();
    DTICK(11)
    if constexpr (NC > 0) {  // recover the eliminated slack step and the new multipliers of d - s = 0
      OCP_FOR(e, N * NC) {
        const int k = e / NC, m = e - k * NC;
        double ds = 0.0, nun = 0.0;
        if (row_on(pc, k, m)) {
          ds = l.cd[e] - l.cs[e];
#pragma unroll
          for (int i = 0; i < NZ; ++i) ds += l.Jd[e * NZ + i] * l.D[k * NZ + i];
          nun = resto ? 0.0 : (l.csig[e] + delta) * ds + l.crb[e];   // restoration resets the multipliers afterwards
        }
        l.cds[e] = ds;
        l.cnun[e] = nun;
      }
      __syncthreads();
    }
    return true;
  }

  // ---- feasibility restoration, simplified from W&B sec. 3.3 (same statement as oracle/nmpc.py::_restore) ------
  // Returns 0: restored (Z, cs hold a point acceptable to the filter), 1: failed, 2: converged to a point of locally minimal
  // infeasibility (IPOPT's 'Infeasible_Problem_Detected').  The step solves
  //     min 1/2 |d|^2 - mu_R sum ln(slacks)   s.t.   J d = -c,      mu_R = max(mu, |c|_inf)
  // - IPOPT's restoration NLP keeps its iterates inside the bounds with a barrier of that parameter (W&B sec. 3.3); without the
  // barrier's Newton terms the least-norm step runs into the bounds and the fraction-to-the-boundary rule stalls it.
  __device__ OCP_PHASE static int restore(lds_double* lbase, double* ws, double mu, double tau, int nfilt, double theta_max) {
    lbase = uni(lbase); ws = uni(ws); mu = uni(mu); tau = uni(tau); nfilt = uni(nfilt); theta_max = uni(theta_max);
    const Lds l = carve(lbase, ws);
    const OcpConst& pc = *(const OcpConst*)l.pc;
    const int N = pc.N, SL = (N + 1) * NZ;
    double th = 0.0;
    OCP_FOR(e, N * NX) th += fabs(l.c[e]);
    if constexpr (NC > 0)
      OCP_FOR(e, N * NC) th += row_on(pc, e / NC, e % NC) ? fabs(l.cd[e] - l.cs[e]) : 0.0;
    th = block_reduce<OpSum>(th, l.red);
    const double th_start = th;
    double th_ref = th;
    for (int it = 0; it < 50; ++it) {
      // ten iterations without reducing the violation by 1e-4 in total: the iterates sit at a point of locally minimal
      // infeasibility (IPOPT's restoration NLP would converge there and report Infeasible_Problem_Detected)
      if (it % 10 == 9) {
        if (th > (1.0 - 1e-4) * th_ref && th > 1e-6) return 2;
        th_ref = th;
      }
      double cmax = 0.0;
      OCP_FOR(e, N * NX) cmax = nmax(cmax, fabs(l.c[e]));
      if constexpr (NC > 0)
        OCP_FOR(e, N * NC) cmax = nmax(cmax, row_on(pc, e / NC, e % NC) ? fabs(l.cd[e] - l.cs[e]) : 0.0);
      const double mu_r = uni(fmax(mu, block_reduce<OpMax>(cmax, l.red)));
      OCP_FOR(e, N * NZ * NZ) {   // the restoration step's Hessian is the identity (added as delta = 1) + the barrier's diagonal
        const int row = e / NZ;
        l.W[(size_t)row * WP + (e - row * NZ)] = 0.0;
      }
      __syncthreads();
      OCP_FOR(e, SL) {
        const double lb = l.lbA[e], ub = l.ubA[e], z = l.Z[e];
        double sg = 0.0, r = 0.0;
        if (lb > -INFINITY) { const double is = 1.0 / (z - lb); sg += mu_r * is * is; r -= mu_r * is; }
        if (ub < INFINITY) { const double is = 1.0 / (ub - z); sg += mu_r * is * is; r += mu_r * is; }
        stage_rhs(l, N, e, sg, r, true);
      }
      if constexpr (NC > 0) {
        OCP_FOR(e, N * NC) {
          const int m = e % NC;
          double sg = 0.0, r = 0.0;
          if (row_on(pc, e / NC, m)) {
            if (pc.dlb[m] > -INFINITY) { const double is = 1.0 / (l.cs[e] - pc.dlb[m]); sg += mu_r * is * is; r -= mu_r * is; }
            if (pc.dub[m] < INFINITY) { const double is = 1.0 / (pc.dub[m] - l.cs[e]); sg += mu_r * is * is; r += mu_r * is; }
          }
          l.csig[e] = sg;
          l.crb[e] = r;
        }
      }
      __syncthreads();
      riccati(lbase, ws, mu, 1.0, true);
      double a = 1.0, dmax = 0.0;
      if constexpr (NC > 0) {
        OCP_FOR(e, N * NC) {
          const int m = e % NC;
          if (!row_on(pc, e / NC, m)) continue;
          const double d = l.cds[e];
          if (pc.dlb[m] > -INFINITY && d < 0.0) a = fmin(a, -tau * (l.cs[e] - pc.dlb[m]) / d);
          if (pc.dub[m] < INFINITY && d > 0.0) a = fmin(a, tau * (pc.dub[m] - l.cs[e]) / d);
        }
      }
      OCP_FOR(e, SL) {
        const double d = l.D[e], lb = l.lbA[e], ub = l.ubA[e], z = l.Z[e];
        dmax = nmax(dmax, fabs(d));
        if (lb > -INFINITY && d < 0.0) a = fmin(a, -tau * (z - lb) / d);
        if (ub < INFINITY && d > 0.0) a = fmin(a, tau * (ub - z) / d);
      }
      double alpha = block_reduce<OpMin>(a, l.red);
      dmax = block_reduce<OpMax>(dmax, l.red);
      // the restoration problem is stationary while the constraints are still violated: locally infeasible
      if (dmax <= 1e-9 && th > 1e-6) return 2;
      bool ok = false;
      double tht = 0.0, ft = 0.0;
      while (alpha > 1e-10) {
        OCP_FOR(e, SL) l.Zt[e] = l.Z[e] + alpha * l.D[e];
        if constexpr (NC > 0)
          OCP_FOR(e, N * NC) l.cst[e] = l.cs[e] + alpha * l.cds[e];
        __syncthreads();
        const FTheta trial = eval_values(lbase, ws, l.Zt, l.ct, l.cst);
        ft = trial.f; tht = trial.theta;
        if (isfinite(tht) && tht <= (1.0 - 1e-4 * alpha) * th) { ok = true; break; }
        alpha = uni(alpha * 0.5);
      }
      // no step length reduces the violation along the (barrier-deflected) Newton direction of the constraints: a point of
      // locally minimal infeasibility inside the box
      if (!ok) return th > 1e-6 ? 2 : 1;
      OCP_FOR(e, SL) l.Z[e] = l.Zt[e];
      if constexpr (NC > 0)
        OCP_FOR(e, N * NC) l.cs[e] = l.cst[e];
      __syncthreads();
      th = tht;
      if (th <= 0.9 * th_start && th <= theta_max) {
        const double ph = ft + mu * barrier_logs(l, l.Z, l.cs);
        bool acc = true;
        for (int q = 0; q < nfilt; ++q)
          if (th >= l.filt[2 * q] && ph >= l.filt[2 * q + 1]) { acc = false; break; }
        if (acc) return 0;
      }
      eval_derivs(lbase, ws);
    }
    return 1;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// The solve kernel.  v layout (device, per instance, scaled): [prefix (v_prefix doubles, untouched) | x_0..x_N | u_0..u_{N-1}]
// ---------------------------------------------------------------------------------------------------------------
// The whole solve of one instance: a device function, so that the same body serves the kernels compiled into the library
// (dynamic LDS) and the kernels compiled at run time for user models (static LDS sized for their horizon, hilo_jit.hip).
template <class PB, int TPB>
__device__ __forceinline__ void ocp_solve_body(lds_double* lds_raw, const OcpConst* __restrict__ pcg, int64_t batch,
                                               const double* __restrict__ x0, const double* __restrict__ par,
                                               int64_t par_stride, const double* __restrict__ sdata,
                                               int64_t sd_stride, const double* __restrict__ v0,
                                               int64_t v0_stride, int v0_prefix, int v_prefix, double* __restrict__ v_opt,
                                               double* __restrict__ f_opt, double* __restrict__ lam_g,
                                               double* __restrict__ first, int first_kind,
                                               int32_t* __restrict__ status, int32_t* __restrict__ iters,
                                               double* __restrict__ kkt, long long* __restrict__ prof,
                                               double* __restrict__ ws, const OcpExtra ex = OcpExtra(), int64_t b_in = -1,
                                               int64_t ws_slot = -1) {
  using S = Ocp<PB>;
  constexpr int NX = S::NX, NU = S::NU, NZ = S::NZ, NC = S::NC, NXV = S::NXV, NH = S::NH, NTAIL = NX - NXV - NH;
  const int t = threadIdx.x;
  const int64_t b = b_in >= 0 ? b_in : (int64_t)blockIdx.x;     // instance; workspace slot (workspace mode: one per WORKGROUP)
  if (b >= batch) return;
  const int N = pcg->N;
  double* const wsb = ws ? ws + (ws_slot >= 0 ? ws_slot : b) * (int64_t)S::ws_doubles(N) : nullptr;
  typename S::Lds l = S::carve(lds_raw, wsb, N);
  {  // problem constants into LDS: every later access is an LDS read instead of a global load
    const double* src = reinterpret_cast<const double*>(pcg);
    lds_double* dst = lds_raw;
    OCP_FOR(i, S::NCONST) dst[i] = src[i];
  }
  {
    const int n1 = ex.par2 || ex.npar1 > 0 ? ex.npar1 : PB::NPAR;   // default: the whole row from `par`
    OCP_FOR(i, PB::NPAR) {
      double v = 0.0;
      if (i < n1) v = par[b * par_stride + i];
      else if (ex.par2) v = ex.par2[b * (int64_t)(PB::NPAR - n1) + (i - n1)];
      l.par[i] = v;
    }
  }
  if constexpr (PB::NSD > 0)
    OCP_FOR(i, (N + 1) * PB::NSD) l.sd[i] = sdata[b * sd_stride + i];
  if constexpr (S::COOP) {   // the learned term's table into LDS: every kernel-sum term is an LDS read instead of an L2 round trip
    const double* gsrc = pcg->ext;
    const int nt = GP2_HDR + 3 * (int)gsrc[0];
    OCP_FOR(i, nt) l.ext[S::NEXT_SCR + i] = gsrc[i];
  }
  __syncthreads();
  const OcpConst& pc = *(const OcpConst*)l.pc;
  const int SL = (N + 1) * NZ;
  long long tprof[PH_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = prof ? clock64() : 0;
#define OCP_TICK(ph) if (prof) { const long long tn = clock64(); tprof[ph] += tn - tlast; tlast = tn; }

  // ---- load: warm start in the [x-block | u-block] layout; x_0 pinned to the measurement (mpc.py:801-802) ----
  const double* vb = v0 + b * v0_stride + v0_prefix;
  OCP_FOR(e, SL) {
    const int k = e / NZ, i = e - k * NZ;
    double v;
    // reference layout [x (NXV per stage) | u | shared tail]; the tail is carried as constant states (same value in every stage)
    // with a control horizon Nc < N the reference holds Nc input blocks; the held copy of stage k is the input of stage
    // min(k - 1, Nc - 1) (zero at stage 0, where it is pinned and never read)
    const int Ncv = NH > 0 ? pc.Nc : N;
    if (i < NXV) v = (k == 0 && S::x0_pinned(pc, i)) ? x0[b * S::NX0 + i] / pc.sz[i] : vb[k * NXV + i];
    else if (i < NXV + NTAIL) v = vb[(pc.tail_off > 0 ? pc.tail_off : (N + 1) * NXV + Ncv * NU) + (i - NXV)];
    else if (i < NX) v = k == 0 ? 0.0 : vb[(N + 1) * NXV + ((k - 1 < Ncv ? k - 1 : Ncv - 1)) * NU + (i - NXV - NTAIL)];
    else v = (k < Ncv) ? vb[(N + 1) * NXV + k * NU + (i - NX)] : 0.0;
    const bool fr = S::is_free(pc, k, i);
    double lb = fr ? S::lb_of(pc, k, i) : -INFINITY, ub = fr ? S::ub_of(pc, k, i) : INFINITY;
    if (ex.lbx && fr && !(k == 0 && (pc.flags & 2) && i < S::NX0)) {   // bounds of this call (the box of a free x_0 stays its own): the slot's entry of the reference's lbx / ubx (same index as in v)
      int src = -1;
      if (i < NXV) src = k * NXV + i;
      else if (i < NXV + NTAIL) { if (k == 0) src = (pc.tail_off > 0 ? pc.tail_off : (N + 1) * NXV + Ncv * NU) + (i - NXV); }
      else if (i >= NX && k < Ncv) src = (N + 1) * NXV + k * NU + (i - NX);
      if (src >= 0) {
        const double lo = ex.lbx[b * ex.bx_stride + v_prefix + src], up = ex.ubx[b * ex.bx_stride + v_prefix + src];
        lb = lo > -INFINITY ? lo - pc.bound_relax * fmax(1.0, fabs(lo)) : lo;
        ub = up < INFINITY ? up + pc.bound_relax * fmax(1.0, fabs(up)) : up;
      }
    }
    l.lbA[e] = lb;
    l.ubA[e] = ub;
    if (fr) {  // IPOPT start: push into the interior (W&B sec. 3.6)
      const bool hl = lb > -INFINITY, hu = ub < INFINITY;
      if (hl) {
        double pl = pc.bound_push * fmax(1.0, fabs(lb));
        if (hu) pl = fmin(pl, pc.bound_frac * (ub - lb));
        v = fmax(v, lb + pl);
      }
      if (hu) {
        double pu = pc.bound_push * fmax(1.0, fabs(ub));
        if (hl) pu = fmin(pu, pc.bound_frac * (ub - lb));
        v = fmin(v, ub - pu);
      }
      l.zL[e] = hl ? 1.0 : 0.0;
      l.zU[e] = hu ? 1.0 : 0.0;
    } else {
      l.zL[e] = 0.0;
      l.zU[e] = 0.0;
    }
    l.Z[e] = v;
    l.D[e] = 0.0;
  }
  OCP_FOR(e, N * NX) l.lam[e] = 0.0;
  if constexpr (NC > 0) {
    OCP_FOR(e, N * NC) {
      l.cs[e] = 0.0; l.cst[e] = 0.0; l.cnu[e] = 0.0; l.cnun[e] = 0.0; l.cvL[e] = 0.0; l.cvU[e] = 0.0; l.cdvL[e] = 0.0;
      l.cdvU[e] = 0.0; l.cds[e] = 0.0; l.cd[e] = 0.0; l.csig[e] = 0.0; l.crb[e] = 0.0;
    }
    OCP_FOR(e, N * NC * NZ) l.Jd[e] = 0.0;
  }
  __syncthreads();
  if constexpr (NC > 0) {  // IPOPT: slacks start at d(w_0), pushed into the interior of their bounds
    S::eval_values(lds_raw, wsb, l.Z, l.ct, nullptr, l.cd);
    __syncthreads();
    OCP_FOR(e, N * NC) {
      const int m = e % NC;
      if (!S::row_on(pc, e / NC, m)) continue;
      double v = l.cd[e];
      const double lb = pc.dlb[m], ub = pc.dub[m];
      const bool hl = lb > -INFINITY, hu = ub < INFINITY;
      if (hl) {
        double pl = pc.bound_push * fmax(1.0, fabs(lb));
        if (hu) pl = fmin(pl, pc.bound_frac * (ub - lb));
        v = fmax(v, lb + pl);
      }
      if (hu) {
        double pu = pc.bound_push * fmax(1.0, fabs(ub));
        if (hl) pu = fmin(pu, pc.bound_frac * (ub - lb));
        v = fmin(v, ub - pu);
      }
      l.cs[e] = v;
      l.cvL[e] = hl ? 1.0 : 0.0;
      l.cvU[e] = hu ? 1.0 : 0.0;
    }
    __syncthreads();
  }

  // smallest barrier parameter: W&B eq. (7) writes eps_tol / 10; IPOPT's MonotoneMuUpdate uses
  // min(tol, compl_inf_tol) / (barrier_tol_factor + 1) - the value behind the last printed digit of the reference's CSTR notebook
  const double mu_min = uni(fmin(pc.tol, 1e-4) / (pc.kappa_eps + 1.0));
  double mu = uni(pc.mu_init), tau = uni(fmax(pc.tau_min, 1.0 - mu));
  double delta_last = 0.0;
  int nfilt = 0, acc_count = 0, it = 0, st = 0;
  double theta_min = 0.0, theta_max = INFINITY;
  double E0 = INFINITY, fval = 0.0;
  double blog = 0.0;       // -sum log(slacks) at the current iterate, carried over from the accepted trial point
  bool blog_ok = false;

  if constexpr (S::SYM) S::term_hess_dirs(l);
  if constexpr (!S::SYM && !S::SYM_MHE && !S::COOP) S::build_dirs(l);
  const double nb_const = S::count_bounds(l);
  bool pts_ok = false;      // SYM policies: the iterate is the trial point the line search evaluated last (stage points, defects, f)
  double f_trial = 0.0;
  for (it = 0;; ++it) {
    if constexpr (S::SYM) {
      if (pts_ok) {
        OCP_FOR(e, N * NX) l.c[e] = l.ct[e];
        (void)S::eval_derivs_sym(lds_raw, wsb, true);
        fval = f_trial;
      } else fval = S::eval_derivs(lds_raw, wsb);
    } else fval = S::eval_derivs(lds_raw, wsb, S::COOP && pts_ok);
    OCP_TICK(PH_DERIV)
    const typename S::KktErr ke = S::kkt_pass(l, nb_const);
    DTICK0
    const double dual_s = ke.dual_s, prim = ke.prim, i_s_c = ke.i_s_c, th0 = ke.theta;
    const double c0 = uni(fmax(ke.pmax, -ke.pmin));
    if (it == 0) {
      theta_min = uni(pc.theta_min_fact * fmax(1.0, th0));
      theta_max = uni(pc.theta_max_fact * fmax(1.0, th0));
    }
    E0 = uni(nmax(nmax(dual_s, prim), c0 * i_s_c));
    if (E0 != E0) { st = HILO_STATUS_OTHER; break; }   // NaN in the iterate: IPOPT's 'Invalid_Number_Detected' -> -1
    if (E0 <= pc.tol) { st = HILO_STATUS_SOLVED; break; }
    if (E0 <= pc.acceptable_tol) {
      if (++acc_count >= pc.acceptable_iter) { st = HILO_STATUS_ACCEPTABLE; break; }
    } else acc_count = 0;
    if (it >= pc.max_iter) { st = HILO_STATUS_MAXITER; break; }
    // ---- barrier update (W&B eq. 7) ----
    for (int r = 0; r < 20; ++r) {
      const double Emu = uni(nmax(nmax(dual_s, prim), S::compl_of(ke, mu) * i_s_c));
      if (!(Emu <= pc.kappa_eps * mu && mu > mu_min * (1 + 1e-12))) break;
      mu = uni(fmax(mu_min, fmin(pc.kappa_mu * mu, pc.theta_mu == 1.5 ? mu * sqrt(mu) : pow(mu, pc.theta_mu))));
      tau = uni(fmax(pc.tau_min, 1.0 - mu));
      nfilt = 0;
    }
    DTICK(14)
    OCP_TICK(PH_ERR)
    // ---- search direction with inertia correction (W&B Alg. IC) ----
    S::finish_rhs(l, mu);
    double delta = 0.0;
    bool first_try = true, solved = false;
    for (;;) {
      tprof[PH_NRIC] += 1;
      if (uni(S::riccati(lds_raw, wsb, mu, delta))) { solved = true; break; }
      if (first_try) {
        delta = uni(delta_last == 0.0 ? pc.delta_w_0 : fmax(pc.delta_w_min, pc.kappa_w_minus * delta_last));
        first_try = false;
      } else {
        delta = uni(delta * (delta_last == 0.0 ? pc.kappa_w_plus_bar : pc.kappa_w_plus));
      }
      if (delta > pc.delta_w_max) break;
    }
    if (!solved) { st = HILO_STATUS_RESTORATION_FAILED; break; }
    if (delta > 0.0) delta_last = delta;
    OCP_TICK(PH_RICCATI)
    DTICKR
    // ---- bound-multiplier steps, fraction to the boundary (W&B eq. 8), directional derivative ----
    // step lengths as tau / max(ratio): the largest relative decrease  -d / slack  (primal) and  -dz / z  (bound multipliers)
    // over the slots, with the reciprocals of the slacks the multiplier steps need anyway - one division per step length
    // instead of one per slot and side
    double r_p = 0.0, r_z = 0.0, dphi = 0.0;
    {   // U slots per lane and trip, reads first, no branches (see kkt_pass)
      constexpr int U = S::BIG ? 3 : 2;
      for (int base = 0; base < SL; base += OCP_TPB * U) {
        double dv[U], lbv[U], ubv[U], zv[U], zlv[U], zuv[U], gv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e0 = base + u * OCP_TPB + (int)threadIdx.x;
          ok[u] = e0 < SL;
          const int e = ok[u] ? e0 : 0;
          dv[u] = l.D[e]; lbv[u] = l.lbA[e]; ubv[u] = l.ubA[e]; zv[u] = l.Z[e]; zlv[u] = l.zL[e]; zuv[u] = l.zU[e]; gv[u] = l.grad[e];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const double d = dv[u], zl = zlv[u], zu = zuv[u];
          const bool hl = ok[u] && lbv[u] > -INFINITY, hu = ok[u] && ubv[u] < INFINITY;
          const double isl = rcp_fast(hl ? zv[u] - lbv[u] : 1.0), isu = rcp_fast(hu ? ubv[u] - zv[u] : 1.0);
          const double izl = rcp_fast(hl ? zl : 1.0), izu = rcp_fast(hu ? zu : 1.0);
          // (the multiplier steps themselves are recomputed by the update: two slot vectors less in LDS)
          const double dl = mu * isl - zl - zl * isl * d, du = mu * isu - zu + zu * isu * d;
          r_p = hl ? fmax(r_p, -d * isl) : r_p;
          r_z = hl ? fmax(r_z, -dl * izl) : r_z;
          r_p = hu ? fmax(r_p, d * isu) : r_p;
          r_z = hu ? fmax(r_z, -du * izu) : r_z;
          double gphi = gv[u];
          gphi = hl ? gphi - mu * isl : gphi;
          gphi = hu ? gphi + mu * isu : gphi;
          dphi += ok[u] ? gphi * d : 0.0;   // D = 0 on slots that are not variables
        }
      }
    }
    if constexpr (NC > 0) {
      OCP_FOR(e, N * NC) {
        const int m = e % NC;
        double dl = 0.0, du = 0.0;
        if (S::row_on(pc, e / NC, m)) {
          const double d = l.cds[e];
          if (pc.dlb[m] > -INFINITY) {
            const double is = rcp_fast(l.cs[e] - pc.dlb[m]), vl = l.cvL[e];
            dl = mu * is - vl - vl * is * d;
            r_p = fmax(r_p, -d * is);
            r_z = fmax(r_z, -dl * rcp_fast(vl));
          }
          if (pc.dub[m] < INFINITY) {
            const double is = rcp_fast(pc.dub[m] - l.cs[e]), vu = l.cvU[e];
            du = mu * is - vu + vu * is * d;
            r_p = fmax(r_p, d * is);
            r_z = fmax(r_z, -du * rcp_fast(vu));
          }
          dphi += l.crb[e] * d;
        }
        l.cdvL[e] = dl;
        l.cdvU[e] = du;
      }
    }
    DTICK(15)
    if constexpr (OCP_TPB == 64) {
      double rv[3] = {r_p, r_z, dphi};
      WaveReduceN<R_MAX, R_MAX, R_SUM>::run(rv);
      r_p = rv[0]; r_z = rv[1]; dphi = rv[2];
    } else {
      r_p = block_reduce<OpMax>(r_p, l.red);
      r_z = block_reduce<OpMax>(r_z, l.red);
      dphi = block_reduce<OpSum>(dphi, l.red);
    }
    const double a_p = uni(r_p > tau ? tau / r_p : 1.0), a_z = uni(r_z > tau ? tau / r_z : 1.0);
    if (!blog_ok) blog = S::barrier_logs(l, l.Z, l.cs);
    const double phi0 = uni(fval + mu * blog);
    DTICK(16)
    OCP_TICK(PH_STEP)
    // ---- filter line search (W&B Alg. A) ----
    double alpha = a_p;
    bool accepted = false, armijo = false;
    double blog_t = 0.0, f_acc = 0.0;
    for (int ls = 0; ls < 60; ++ls) {
      DTICKR
      const double blog_part = S::form_trial_part(l, alpha);
      DTICK(17)
      tprof[PH_NLS] += 1;
      const FTheta trial = S::eval_values(lds_raw, wsb, l.Zt, l.ct, l.cst, l.cdt, blog_part);
      blog_t = trial.x;
      DTICK(18)
      const double ft = trial.f, tht = trial.theta;
      const double pht = uni(ft + mu * blog_t);
      bool ok = isfinite(pht) && isfinite(tht) && tht <= theta_max;
      if (ok) {
        for (int q = 0; q < nfilt; ++q) {
          const double tf = l.filt[2 * q], pf = l.filt[2 * q + 1];
          if (tht >= tf && pht - 10 * 2.220446049250313e-16 * fabs(pf) >= pf) { ok = false; break; }
        }
      }
      bool sw = false;
      if (ok) {
        sw = th0 <= theta_min && dphi < 0.0 && S::switching(pc, alpha, -dphi, th0);
        const double rnd = 10 * 2.220446049250313e-16 * fabs(phi0);
        if (sw) ok = pht - phi0 - rnd <= pc.eta_phi * alpha * dphi;
        else ok = tht <= (1 - pc.gamma_theta) * th0 || pht - phi0 - rnd <= -pc.gamma_phi * th0;
      }
      DTICK(19)
      if (ok) { accepted = true; armijo = sw; f_acc = ft; break; }
      // ---- second-order correction (W&B sec. 2.4): the full step was rejected and did not reduce the violation.  The same
      // system is solved with c_soc = alpha c(x_k) + c(x_k + alpha d); up to four corrections while theta drops by kappa_soc.
      if (ls == 0 && tht >= th0) {
        OCP_FOR(e, N * NX) l.c0[e] = l.c[e];
        if constexpr (NC > 0)
          OCP_FOR(e, N * NC) l.cd0[e] = l.cd[e];
        double a_prev = alpha, th_old = tht;
        bool soc_ok = false, soc_sw = false;
        for (int ps = 0; ps < 4; ++ps) {
          OCP_FOR(e, N * NX) l.c[e] = a_prev * l.c[e] + l.ct[e];
          if constexpr (NC > 0) {
            OCP_FOR(e, N * NC) {
              if (S::row_on(pc, e / NC, e % NC)) l.cd[e] = l.cs[e] + a_prev * (l.cd[e] - l.cs[e]) + (l.cdt[e] - l.cst[e]);
            }
          }
          __syncthreads();
          tprof[PH_NRIC] += 1;
          if (!uni(S::riccati(lds_raw, wsb, mu, delta))) break;
          double a = 1.0;
          OCP_FOR(e, SL) {
            const double d = l.D[e], lb = l.lbA[e], ub = l.ubA[e], z = l.Z[e];
            if (lb > -INFINITY && d < 0.0) a = fmin(a, -tau * (z - lb) / d);
            if (ub < INFINITY && d > 0.0) a = fmin(a, tau * (ub - z) / d);
          }
          if constexpr (NC > 0) {
            OCP_FOR(e, N * NC) {
              const int m = e % NC;
              if (!S::row_on(pc, e / NC, m)) continue;
              const double d = l.cds[e];
              if (pc.dlb[m] > -INFINITY && d < 0.0) a = fmin(a, -tau * (l.cs[e] - pc.dlb[m]) / d);
              if (pc.dub[m] < INFINITY && d > 0.0) a = fmin(a, tau * (pc.dub[m] - l.cs[e]) / d);
            }
          }
          const double a_s = block_reduce<OpMin>(a, l.red);
          const double blog_part2 = S::form_trial_part(l, a_s);
          tprof[PH_NLS] += 1;
          const FTheta t2 = S::eval_values(lds_raw, wsb, l.Zt, l.ct, l.cst, l.cdt, blog_part2);
          blog_t = t2.x;
          const double ths = t2.theta;
          const double phs = uni(t2.f + mu * blog_t);
          bool oks = isfinite(phs) && isfinite(ths) && ths <= theta_max;
          if (oks) {
            for (int q = 0; q < nfilt; ++q) {
              const double tf = l.filt[2 * q], pf = l.filt[2 * q + 1];
              if (ths >= tf && phs - 10 * 2.220446049250313e-16 * fabs(pf) >= pf) { oks = false; break; }
            }
          }
          if (oks) {
            const bool sw2 = th0 <= theta_min && dphi < 0.0 && S::switching(pc, alpha, -dphi, th0);
            const double rnd = 10 * 2.220446049250313e-16 * fabs(phi0);
            if (sw2) oks = phs - phi0 - rnd <= pc.eta_phi * alpha * dphi;
            else oks = ths <= (1 - pc.gamma_theta) * th0 || phs - phi0 - rnd <= -pc.gamma_phi * th0;
            if (oks) { soc_ok = true; soc_sw = sw2; f_acc = t2.f; break; }
          }
          if (!(ths <= 0.99 * th_old)) break;
          th_old = ths;
          a_prev = a_s;
        }
        if (soc_ok) {
          // The corrected step replaces the primal step and the equality multipliers; the bound-multiplier steps (which the
          // update recomputes from the direction in l.D) stay those of the UNCORRECTED direction, like a_z: rebuild it - the
          // corrected system's multipliers wait in the saved-defect slots meanwhile.  (One more factorisation in a rare branch.)
          OCP_FOR(e, N * NX) {
            const double cs0 = l.c0[e];
            l.c0[e] = l.lamn[e];
            l.c[e] = cs0;
          }
          if constexpr (NC > 0) {
            OCP_FOR(e, N * NC) {
              const double cd0 = l.cd0[e];
              l.cd0[e] = l.cnun[e];
              l.cd[e] = cd0;
            }
          }
          __syncthreads();
          tprof[PH_NRIC] += 1;
          (void)uni(S::riccati(lds_raw, wsb, mu, delta));
          OCP_FOR(e, N * NX) l.lamn[e] = l.c0[e];
          if constexpr (NC > 0)
            OCP_FOR(e, N * NC) l.cnun[e] = l.cd0[e];
          __syncthreads();
          accepted = true; armijo = soc_sw; break;
        }
        // no luck: back to the uncorrected direction (defects, row values, step, multipliers) and on with the backtracking
        OCP_FOR(e, N * NX) l.c[e] = l.c0[e];
        if constexpr (NC > 0)
          OCP_FOR(e, N * NC) l.cd[e] = l.cd0[e];
        __syncthreads();
        tprof[PH_NRIC] += 1;
        (void)uni(S::riccati(lds_raw, wsb, mu, delta));
      }
      alpha = uni(alpha * 0.5);
      // W&B eq. 23: below alpha_min the line search gives up and the restoration phase is called
      double amin = pc.gamma_theta;
      if (dphi < 0.0) {
        amin = fmin(amin, pc.gamma_phi * th0 / (-dphi));
        if (th0 <= theta_min) amin = fmin(amin, pc.delta_ls * pow(th0, pc.s_theta) / pow(-dphi, pc.s_phi));
      }
      if (alpha < 0.05 * amin) break;
    }
    OCP_TICK(PH_LS)
    DTICKR
    const bool do_resto = !accepted;
    if (!armijo || do_resto) {  // augment the filter (W&B eq. 22); also done before entering restoration
      if (nfilt == OCP_FILTER) {
        double keep = 0.0;
        if (t < 2 * (OCP_FILTER - 1)) keep = l.filt[t + 2];
        __syncthreads();
        if (t < 2 * (OCP_FILTER - 1)) l.filt[t] = keep;
        nfilt = OCP_FILTER - 1;
        __syncthreads();
      }
      if (t == 0) {
        l.filt[2 * nfilt] = (1 - pc.gamma_theta) * th0;
        l.filt[2 * nfilt + 1] = phi0 - pc.gamma_phi * th0;
      }
      ++nfilt;
      __syncthreads();
    }
    if (do_resto) {
      const int rr = uni(S::restore(lds_raw, wsb, mu, tau, nfilt, theta_max));
      if (rr != 0) { st = rr == 2 ? HILO_STATUS_INFEASIBLE : HILO_STATUS_RESTORATION_FAILED; break; }
      blog_ok = false;
      pts_ok = false;
      // IPOPT after restoration: equality multipliers reset (constr_mult_reset_threshold = 0), bound multipliers
      // reset to 1 when they exceed bound_mult_reset_threshold = 1000
      double zm = 0.0;
      OCP_FOR(e, SL) zm = fmax(zm, fmax(l.zL[e], l.zU[e]));
      if constexpr (NC > 0)
        OCP_FOR(e, N * NC) zm = fmax(zm, fmax(l.cvL[e], l.cvU[e]));
      const bool zreset = uni(block_reduce<OpMax>(zm, l.red) > 1e3);   // (wave-uniform: a scalar branch around the element loops)
      if (zreset) {
        OCP_FOR(e, SL) {
          l.zL[e] = l.lbA[e] > -INFINITY ? 1.0 : 0.0;
          l.zU[e] = l.ubA[e] < INFINITY ? 1.0 : 0.0;
        }
      }
      OCP_FOR(e, N * NX) l.lam[e] = 0.0;
      if constexpr (NC > 0) {
        OCP_FOR(e, N * NC) l.cnu[e] = 0.0;
        if (zreset) {
          OCP_FOR(e, N * NC) {
            const int m = e % NC;
            const bool on = S::row_on(pc, e / NC, m);
            const double vl = l.cvL[e], vu = l.cvU[e];
            l.cvL[e] = on ? (pc.dlb[m] > -INFINITY ? 1.0 : 0.0) : vl;
            l.cvU[e] = on ? (pc.dub[m] < INFINITY ? 1.0 : 0.0) : vu;
          }
        }
      }
      __syncthreads();
      continue;
    }
    // ---- accept: primal, equality multipliers, bound multipliers (+ W&B eq. 16 safeguard) ----
    blog = blog_t;
    blog_ok = true;
    pts_ok = true;
    f_trial = f_acc;
    const double ks_lo = uni(mu / pc.kappa_sigma), ks_hi = uni(pc.kappa_sigma * mu);
    {   // U slots per lane and trip, reads first, no branches (see kkt_pass)
      constexpr int U = S::BIG ? 3 : 2;
      for (int base = 0; base < SL; base += OCP_TPB * U) {
        double zn[U], zo[U], dv[U], lbv[U], ubv[U], zlv[U], zuv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e0 = base + u * OCP_TPB + (int)threadIdx.x;
          ok[u] = e0 < SL;
          const int e = ok[u] ? e0 : 0;
          zn[u] = l.Zt[e]; zo[u] = l.Z[e]; dv[u] = l.D[e]; lbv[u] = l.lbA[e]; ubv[u] = l.ubA[e]; zlv[u] = l.zL[e]; zuv[u] = l.zU[e];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = base + u * OCP_TPB + (int)threadIdx.x;
          const double d = dv[u], zl = zlv[u], zu = zuv[u];
          const bool hl = ok[u] && lbv[u] > -INFINITY, hu = ok[u] && ubv[u] < INFINITY;
          // dzL = mu / s - zL - zL d / s at the old point (the step phase's formula)
          const double iol = rcp_fast(hl ? zo[u] - lbv[u] : 1.0), isl = rcp_fast(hl ? zn[u] - lbv[u] : 1.0);
          const double iou = rcp_fast(hu ? ubv[u] - zo[u] : 1.0), isu = rcp_fast(hu ? ubv[u] - zn[u] : 1.0);
          const double dl = mu * iol - zl - zl * iol * d, du = mu * iou - zu + zu * iou * d;
          const double zln = fmin(fmax(zl + a_z * dl, ks_lo * isl), ks_hi * isl), zun = fmin(fmax(zu + a_z * du, ks_lo * isu), ks_hi * isu);
          if (ok[u]) l.Z[e] = zn[u];
          if (hl) l.zL[e] = zln;
          if (hu) l.zU[e] = zun;
        }
      }
      const int NV = N * NX;
      for (int base = 0; base < NV; base += OCP_TPB * U) {
        double la[U], ln[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e0 = base + u * OCP_TPB + (int)threadIdx.x;
          ok[u] = e0 < NV;
          la[u] = l.lam[ok[u] ? e0 : 0]; ln[u] = l.lamn[ok[u] ? e0 : 0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (ok[u]) l.lam[base + u * OCP_TPB + (int)threadIdx.x] = la[u] + alpha * (ln[u] - la[u]);
      }
    }
    if constexpr (NC > 0) {
      OCP_FOR(e, N * NC) {
        const int m = e % NC;
        if (!S::row_on(pc, e / NC, m)) continue;
        const double snew = l.cst[e];
        l.cs[e] = snew;
        l.cnu[e] += alpha * (l.cnun[e] - l.cnu[e]);
        if (pc.dlb[m] > -INFINITY) {
          const double is = rcp_fast(snew - pc.dlb[m]);
          l.cvL[e] = fmin(fmax(l.cvL[e] + a_z * l.cdvL[e], ks_lo * is), ks_hi * is);
        }
        if (pc.dub[m] < INFINITY) {
          const double is = rcp_fast(pc.dub[m] - snew);
          l.cvU[e] = fmin(fmax(l.cvU[e] + a_z * l.cdvU[e], ks_lo * is), ks_hi * is);
        }
      }
    }
    __syncthreads();
    DTICK(20)
    OCP_TICK(PH_UPDATE)
  }

  // ---- write back ([x-block | u-block] after the prefix) ----
  const int Ncw = NH > 0 ? pc.Nc : N;
  double* vo = v_opt + b * (int64_t)(v_prefix + (N + 1) * NXV + Ncw * NU + NTAIL) + v_prefix;
  double* const vo2 = ex.v_copy ? ex.v_copy + (vo - v_opt) : nullptr;
  OCP_FOR(e, SL) {
    const int k = e / NZ, i = e - k * NZ;
    int dst = -1;
    if (i < NXV) dst = k * NXV + i;
    else if (i < NXV + NTAIL) { if (k == 0) dst = (N + 1) * NXV + Ncw * NU + (i - NXV); }
    else if (i < NX) {}   // held inputs: copies of u_{Nc-1}, not part of the reference's decision vector
    else if (k < Ncw) dst = (N + 1) * NXV + k * NU + (i - NX);
    if (dst >= 0) {
      const double zv = l.Z[e];
      vo[dst] = zv;
      if (vo2) vo2[dst] = zv;
      if (ex.lam_x) (ex.lam_x + (vo - v_opt))[dst] = S::is_free(pc, k, i) ? l.zU[e] - l.zL[e] : 0.0;
    }
  }
  if (lam_g) {
    // the reference's g: per stage [shooting defect (NXV rows) | constraint rows (n_con_ref)] (mpc.py:1667, :1707-1725)
    // the last stage carries the terminal rows between its defect and its stage rows (mpc.py:1693-1700 before :1707)
    const int ncr = NC > 0 ? pc.n_con_ref : 0, ntr = NC > 0 ? pc.n_tcon_ref : 0, rows = NXV + ncr;
    double* lg = lam_g + b * (int64_t)(N * rows + ntr);
    double* const gg = ex.g ? ex.g + b * (int64_t)(N * rows + ntr) : nullptr;
    if (gg) {   // constraint values at the returned point: defects x_{k+1} - F_k, then the rows (dropped rows: 0)
      OCP_FOR(e, N * rows + ntr) gg[e] = 0.0;
      __syncthreads();
      OCP_FOR(e, N * NXV) gg[(e / NXV) * rows + e % NXV] = l.c[(e / NXV) * NX + e % NXV];
      if constexpr (NC > 0) {
        OCP_FOR(e, N * NC) {
          const int k = e / NC, m = e - k * NC;
          if (m < pc.nc) gg[k * rows + NXV + (k == N - 1 ? ntr : 0) + pc.row_ref[m]] = l.cd[e];
          else if (k == N - 1 && m < pc.nc + pc.nc_term) gg[k * rows + NXV + pc.trow_ref[m - pc.nc]] = l.cd[e];
        }
      }
    }
    OCP_FOR(e, N * NXV) {
      const int k = e / NXV, i = e - k * NXV;
      double v = l.lam[k * NX + i];
      // terminal cost on F_{N-1} in the reference (mpc.py:1682) vs on x_N here: multipliers of the last defect
      // differ by grad V(x_N) (flag bit 0)
      if ((pc.flags & 1) && k == N - 1) v += l.grad[N * NZ + i];
      if constexpr (pb_lam_fix<PB>::value && NC > 0) {
        if (k == N - 1) {
          double xN[NX], nuN[NC];
#pragma unroll
          for (int j = 0; j < NX; ++j) xN[j] = l.Z[N * NZ + j];
#pragma unroll
          for (int m = 0; m < NC; ++m) nuN[m] = l.cnu[(N - 1) * NC + m];
          v += PB::lam_fix(pc, (const double*)l.par, S::sd_of(l, N), i, xN, nuN);
        }
      }
      lg[k * rows + i] = v;
    }
    if constexpr (NC > 0) {
      OCP_FOR(e, N * ncr) {  // dropped (unbounded) rows
        const int k = e / ncr;
        lg[k * rows + NXV + (k == N - 1 ? ntr : 0) + e % ncr] = 0.0;
      }
      OCP_FOR(e, ntr) lg[(N - 1) * rows + NXV + e] = 0.0;
      __syncthreads();
      OCP_FOR(e, N * NC) {
        const int k = e / NC, m = e - k * NC;
        if (m < pc.nc) lg[k * rows + NXV + (k == N - 1 ? ntr : 0) + pc.row_ref[m]] = l.cnu[e];
        else if (k == N - 1 && m < pc.nc + pc.nc_term) lg[k * rows + NXV + pc.trow_ref[m - pc.nc]] = l.cnu[e];
      }
    }
  }
  if (first) {
    if (first_kind == 0) { OCP_FOR(a, S::NU0) first[b * S::NU0 + a] = l.Z[NX + a] * pc.sz[NX + a]; }
    else { OCP_FOR(a, NX) first[b * NX + a] = l.Z[N * NZ + a] * pc.sz[a]; }
  }
  if (ex.gather && first_kind == 0) {
    double* gr = ex.gather + b * (int64_t)ex.gather_stride;
    OCP_FOR(a, S::NU0) gr[a] = l.Z[NX + a] * pc.sz[NX + a];
    if (t == 0) { gr[S::NU0] = (double)st; gr[S::NU0 + 1] = (double)it; }
  }
  if constexpr (pb_plant<PB>::value) {
    if (ex.x_next && t == 0) {   // (one lane: four right-hand sides, under a microsecond - a launch and its gap cost ten)
      double xs[S::NX0], us[S::NU0 > 0 ? S::NU0 : 1];
#pragma unroll
      for (int i = 0; i < S::NX0; ++i) xs[i] = x0[b * S::NX0 + i];
#pragma unroll
      for (int a = 0; a < S::NU0; ++a) us[a] = l.Z[NX + a] * pc.sz[NX + a];
      PB::plant(pc, (const double*)l.par, xs, us, ex.x_next + b * S::NX0);
    }
  }
  if (t == 0) {
    f_opt[b] = fval;
    status[b] = st;
    iters[b] = it;
    if (kkt) kkt[b] = E0;
    if (prof && b == 0)
      for (int q = 0; q < PH_COUNT; ++q) prof[q] = tprof[q];
  }
#undef OCP_TICK
}

template <class PB, int TPB>
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(HILO_OCP_MINW, HILO_OCP_MINW))) void ocp_solve_kernel(const OcpConst* __restrict__ pcg, int64_t batch,
                                                       const double* __restrict__ x0, const double* __restrict__ par,
                                                       int64_t par_stride, const double* __restrict__ sdata,
                                                       int64_t sd_stride, const double* __restrict__ v0,
                                                       int64_t v0_stride, int v0_prefix, int v_prefix, double* __restrict__ v_opt,
                                                       double* __restrict__ f_opt, double* __restrict__ lam_g,
                                                       double* __restrict__ first, int first_kind,
                                                       int32_t* __restrict__ status, int32_t* __restrict__ iters,
                                                       double* __restrict__ kkt, long long* __restrict__ prof,
                                                       double* __restrict__ ws = nullptr, const OcpExtra ex = OcpExtra()) {
  extern __shared__ double lds_raw_generic[];
  if constexpr (PB::BIG) {
    // workspace mode: a workgroup walks over instances blockIdx.x, + gridDim.x, ... and keeps ONE workspace slot (grid = batch
    // by default: one instance per workgroup; a smaller grid keeps the slots cache-resident, see big_grid_slots())
    for (int64_t b = blockIdx.x; b < batch; b += gridDim.x) {
      ocp_solve_body<PB, TPB>((lds_double*)lds_raw_generic, pcg, batch, x0, par, par_stride, sdata, sd_stride, v0, v0_stride,
                              v0_prefix, v_prefix, v_opt, f_opt, lam_g, first, first_kind, status, iters, kkt, prof, ws, ex, b,
                              (int64_t)blockIdx.x);
      __syncthreads();
    }
  } else {
    ocp_solve_body<PB, TPB>((lds_double*)lds_raw_generic, pcg, batch, x0, par, par_stride, sdata, sd_stride, v0, v0_stride,
                            v0_prefix, v_prefix, v_opt, f_opt, lam_g, first, first_kind, status, iters, kkt, prof, ws, ex);
  }
}

}  // namespace hilo
