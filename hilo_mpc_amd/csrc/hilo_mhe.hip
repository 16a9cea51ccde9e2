// Batched moving-horizon estimation on the stage-structured interior-point engine (hilo_ocp.h) + C ABI.
//
// Replaces `ca.nlpsol("solver",'ipopt',{'f','x','p','g'})` built by `MovingHorizonEstimator.setup`
// (hilo_mpc/modules/estimator/mhe.py:782-790) and called from `estimate` (mhe.py:375) for a pre-discretised model,
// `integration_method='discrete'`, state noise, pinned model parameters.  Transcription restated from
// mhe.py:596-760 (quirks Q7 kept):
//   v = [p | x_0..x_N | w_0..w_{N-1}]                                         mhe.py:614-655
//   g_k = x_{k+1} - (Phi_s(x_k, u_meas_k, p) + w_k) = 0                       mhe.py:733-740
//   J   = arrival(x_0) at k = 0; (h(x_k)-y_k)^T Wy (.) + w_k^T Ww w_k, k >= 1  mhe.py:742-748 (no stage cost at k = 0)
//   costs act on un-scaled quantities (hilo_mpc/util/modeling.py:665-672)
// In engine terms: controls := the noise w_k (B_k = I), x_0 free, per-stage data = (u_meas_k, y_meas_k).
// Round 3: models written as expressions and the collocation transcription (the reference's default, mhe.py:512-561) go
// through the run-time compiled policy (hilo_jit.hip, JIT_MHE: desc.user_source) - the zoo models too when collocation is asked
// for; the policy itself lives in hilo_mhe_policy.h.
#include <stdlib.h>
#include <string.h>

#include "hilo_jit.h"
#include "hilo_mhe_est.h"
#include "hilo_mhe_policy.h"

namespace hilo {

// par[b] = [p_b | x_arrival_b];  sd[b][k] = [u_meas[b][k] | y_meas[b][k]] for k < N, zeros for k = N
__global__ void mhe_pack_kernel(int64_t batch, int N, int np, int nx, int nu, int ny, const double* __restrict__ p,
                                int64_t p_stride, const double* __restrict__ xa, const double* __restrict__ um,
                                const double* __restrict__ ym, double* __restrict__ par, double* __restrict__ sd) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int wp = np + nx, ws = nu + ny;
  const int64_t npar = batch * wp, nsd = batch * (N + 1) * ws;
  if (e < npar) {
    const int64_t b = e / wp;
    const int i = (int)(e - b * wp);
    par[e] = i < np ? p[b * p_stride + i] : xa[b * nx + (i - np)];
  } else if (e < npar + nsd) {
    const int64_t q = e - npar;
    const int64_t b = q / ((N + 1) * ws);
    const int r = (int)(q - b * (N + 1) * ws), k = r / ws, i = r - k * ws;
    double v = 0.0;
    if (k < N) v = i < nu ? um[(b * N + k) * nu + i] : ym[(b * N + k) * ny + (i - nu)];
    sd[q] = v;
  }
}

}  // namespace hilo

using namespace hilo;

struct hilo_mhe {
  int device, model_id, nx, nu, np, ny, N, n_v, n_g;
  OcpConst host;
  OcpConst* dev;
  double *v_guess, *v_warm, *par_buf, *sd_buf;
  int64_t warm_batch, buf_batch, vc_batch;
  int warm_valid;
  size_t lds_bytes;
  JitKernels jit;                    // run-time compiled policy (desc.user_source) or empty
  int use_jit, coll_d, n_vc;         // n_vc: row length of the engine's result [p | x | w] (= n_v without the collocation block)
  double *vc, *lamc;                 // collocation: the engine's result before the output pass
  const MheEstVariant* est;          // parameter-estimating variant or NULL
  int nc;                            // inequality rows per interval of the engine (stage constraint)
  int gen, noise;                    // run-time compiled GENERAL policy (MheGen: parameters as states, optional state noise) / w present
  double *x0e, *v0e, *ve, *lame, *v_guess_e;   // engine-layout buffers of the estimating variant
  unsigned est_mask;                 // bit j: parameter j is a variable
};

#define HILO_MHE_MODELS(X)             \
  X(HILO_MODEL_CHEMOSTAT4, Chemostat4) \
  X(HILO_MODEL_BIOREACTOR3, Bioreactor3)

// `taylor`: the launch will take the policy without symbolic derivatives (sub-stepped integration, mhe_launch) - its iterate holds the
// direction table of the Taylor sweeps and is larger
static int mhe_model_dims(int id, int* nx, int* nu, int* np, int* ny, size_t* lds, int N, bool taylor) {
  switch (id) {
#define X(ID, T) case ID: *nx = T::NX; *nu = T::NU; *np = T::NP; *ny = T::NY; \
    *lds = (taylor ? Ocp<MheNoise<T, false>>::lds_doubles(N) : Ocp<MheNoise<T>>::lds_doubles(N)) * sizeof(double); return HILO_OK;
    HILO_MHE_MODELS(X)
#undef X
  }
  return fail(HILO_ENOTSUP, "model id %d has no MHE instantiation in this build", id);
}

extern "C" void hilo_mhe_destroy(hilo_mhe* h) {
  if (!h) return;
  double* ptrs[] = {(double*)h->dev, h->v_guess, h->v_warm, h->par_buf, h->sd_buf, h->x0e, h->v0e, h->ve, h->lame,
                    h->v_guess_e, h->vc, h->lamc};
  jit_unload(&h->jit);
  for (double* p : ptrs)
    if (p) (void)hipFree(p);
  delete h;
}

extern "C" int hilo_mhe_create(const hilo_mhe_desc* d, int device, hilo_mhe** out) {
  HILO_REQUIRE(d && out, "hilo_mhe_create: NULL argument");
  HILO_REQUIRE(d->N >= 2 && d->N <= 512, "hilo_mhe_create: horizon %d out of range [2, 512]", d->N);
  HILO_REQUIRE(d->dt > 0.0, "hilo_mhe_create: dt must be positive");
  int nx, nu, np, ny;
  size_t lds = 0;
  int rc = HILO_OK;
  const bool jit = d->user_source != nullptr;
  const int D = d->collocation_degree;
  HILO_REQUIRE(D >= 0 && D <= COLL_MAXD, "hilo_mhe_create: collocation degree %d out of range [0, %d]", D, COLL_MAXD);
  HILO_REQUIRE(!D || (d->coll_A && d->coll_D), "hilo_mhe_create: collocation needs the basis (coll_A, coll_D)");
  if (jit) {
    // dimensions: the description's (model written as expressions) or the zoo's (desc.user_source = alias of the zoo functor)
    if (d->model_id == HILO_MODEL_USER) {
      nx = d->user_nx; nu = d->user_nu; np = d->user_np; ny = d->user_ny;
      HILO_REQUIRE(nx >= 1 && nu >= 0 && np >= 0 && ny >= 1, "hilo_mhe_create: bad user model dimensions");
      HILO_REQUIRE(2 * nx <= OCP_MAXNZ && nx <= OCP_MAXNU, "hilo_mhe_create: %d states exceed this build's estimator (%d)", nx, OCP_MAXNU);
    } else {
      int disc = 0;
      rc = hilo_model_dims(d->model_id, &nx, &nu, &np, &ny, &disc);
      if (rc) return rc;
    }
  } else {
    HILO_REQUIRE(d->model_id != HILO_MODEL_USER, "hilo_mhe_create: HILO_MODEL_USER needs desc.user_source");
    if (D) return fail(HILO_ENOTSUP, "the collocation transcription runs on the run-time compiled policy: pass desc.user_source");
    rc = mhe_model_dims(d->model_id, &nx, &nu, &np, &ny, &lds, d->N, d->n_sub > 1 || getenv("HILO_NMPC_TAYLOR") != nullptr);
    if (rc) return rc;
  }
  // desc.Ww == NULL: an estimator WITHOUT state noise (mhe.py:599: no w block in v; the collocation / discrete branches run without
  // it, :726-736) - on the general run-time compiled policy, like parameter estimation for models given as source
  const bool has_noise = d->Ww != nullptr;
  HILO_REQUIRE(d->n_con >= 0 && d->n_con <= OCP_MAXNC && (d->n_con == 0 || (jit && d->con_lb && d->con_ub)),
               "hilo_mhe_create: a stage constraint needs user_source with UserFun and its bounds");
  const bool gen = jit && ((d->estimate_parameters && np > 0) || !has_noise || d->n_con > 0);
  // rows of the stage constraint: one per expression with a finite bound; under collocation the node's rows are followed by those of
  // the D collocation points (mhe.py:536-553, :749-757)
  int nrow_pt = 0, row_expr[OCP_MAXNC];
  double row_lb[OCP_MAXNC], row_ub[OCP_MAXNC];
  for (int j = 0; j < d->n_con; ++j) {
    HILO_REQUIRE(d->con_lb[j] <= d->con_ub[j], "hilo_mhe_create: constraint %d has lb > ub", j);
    if (d->con_lb[j] > -INFINITY || d->con_ub[j] < INFINITY) { row_expr[nrow_pt] = j; row_lb[nrow_pt] = d->con_lb[j]; row_ub[nrow_pt++] = d->con_ub[j]; }
  }
  const int nrow_all = nrow_pt * (D > 0 ? D + 1 : 1);
  HILO_REQUIRE(nrow_all <= OCP_MAXNC, "hilo_mhe_create: %d constraint rows per interval exceed %d", nrow_all, OCP_MAXNC);
  if (!jit && !has_noise) return fail(HILO_ENOTSUP, "an estimator without state noise runs on the run-time compiled policy: pass desc.user_source");
  if (gen)
    HILO_REQUIRE(nx + np <= OCP_MAXNX && (nx + np) + (has_noise ? nx : 0) <= OCP_MAXNZ,
                 "hilo_mhe_create: %d states + %d parameters exceed this build's general estimator (%d engine states)", nx, np, OCP_MAXNX);
  const MheEstVariant* ev = nullptr;
  if (!jit && d->estimate_parameters && np > 0) {
    ev = mhe_est_find(d->model_id);
    if (!ev) return fail(HILO_ENOTSUP, "model id %d has no parameter-estimating MHE instantiation in this build", d->model_id);
    lds = ev->lds_bytes(d->N);
  }
  if (lds > 160 * 1024) return fail(HILO_ENOTSUP, "horizon %d needs %zu B of LDS per instance (limit 163840)", d->N, lds);
  hilo_mhe* h = new hilo_mhe();
  memset(h, 0, sizeof(*h));
  h->est = ev;
  h->device = device; h->model_id = d->model_id; h->nx = nx; h->nu = nu; h->np = np; h->ny = ny; h->N = d->N;
  h->gen = gen ? 1 : 0;
  h->noise = has_noise ? 1 : 0;
  h->n_vc = np + (d->N + 1) * nx + (has_noise ? d->N * nx : 0);   // mhe.py:596-599
  h->n_v = h->n_vc + d->N * D * nx;                          // + collocation states (mhe.py:600-601)
  h->n_g = d->N * (nx + D * nx + (D + 1) * d->n_con);        // per stage [rows at the collocation points | collocation rows |
                                                             // continuity | rows at the node] (mhe.py:536-553, :728, :740, :749-757)
  h->nc = nrow_all;
  h->lds_bytes = lds;
  h->use_jit = jit ? 1 : 0;
  h->coll_d = D;
  OcpConst& c = h->host;
  memset(&c, 0, sizeof(c));
  ocp_default_options(c);
  c.N = d->N; c.Nc = d->N; c.order = d->erk_order >= 1 ? d->erk_order : 4; c.nsub = d->n_sub >= 1 ? d->n_sub : 1; c.dt = d->dt;
  if (D) {
    c.coll.d = D;
    for (int i = 0; i < D * D; ++i) c.coll.A[i] = d->coll_A[i];
    for (int i = 0; i <= D; ++i) { c.coll.Dc[i] = d->coll_D[i]; c.coll.Bq[i] = 0.0; }
  }
  if (d->max_iter > 0) c.max_iter = d->max_iter;
  if (d->acceptable_iter > 0) c.acceptable_iter = d->acceptable_iter;
  if (d->tol > 0) c.tol = d->tol;
  if (d->acceptable_tol > 0) c.acceptable_tol = d->acceptable_tol;
  if (d->mu_init > 0) c.mu_init = d->mu_init;
  for (int i = 0; i < nx; ++i) {
    c.sz[i] = d->x_scaling ? d->x_scaling[i] : 1.0;
    c.sz[nx + i] = d->w_scaling ? d->w_scaling[i] : 1.0;
  }
  {
    double* q = c.cost;  // [Wx | Wy | Ww | su]
    for (int i = 0; i < nx * nx; ++i) *q++ = d->Wx ? d->Wx[i] : 0.0;
    for (int i = 0; i < ny * ny; ++i) *q++ = d->Wy ? d->Wy[i] : 0.0;
    for (int i = 0; i < nx * nx; ++i) *q++ = d->Ww ? d->Ww[i] : 0.0;
    for (int i = 0; i < nu; ++i) *q++ = d->u_scaling ? d->u_scaling[i] : 1.0;
  }
  const double relax = d->bound_relax_factor >= 0.0 ? d->bound_relax_factor : 1e-8;
  c.bound_relax = relax;
  for (int i = 0; i < 2 * nx; ++i) {
    const double* lbs = i < nx ? d->x_lb : d->w_lb;
    const double* ubs = i < nx ? d->x_ub : d->w_ub;
    const int j = i < nx ? i : i - nx;
    double lb = lbs ? lbs[j] / c.sz[i] : -INFINITY, ub = ubs ? ubs[j] / c.sz[i] : INFINITY;
    if (lb > -INFINITY) lb -= relax * fmax(1.0, fabs(lb));
    if (ub < INFINITY) ub += relax * fmax(1.0, fabs(ub));
    HILO_REQUIRE(lb < ub, "hilo_mhe_create: empty box for variable %d", i);
    c.lbz[i] = lb; c.ubz[i] = ub;
  }
  if (ev || gen) {
    // engine state = [x | p], engine input = w: re-lay the per-slot data of the plain variant
    const int nxa = nx + np;
    int o[5];
    if (ev) ev->offsets(o);
    else { o[0] = 0; o[1] = nx * nx; o[2] = o[1] + np * np; o[3] = o[2] + ny * ny; o[4] = o[3] + nx * nx; }   // MheGen::O_*
    const int o_rowx = o[4] + nu, o_nrow = o_rowx + OCP_MAXNC, o_ncr = o_nrow + 1, o_rref = o_ncr + 1;
    double sz[OCP_MAXNZ], lb[OCP_MAXNZ], ub[OCP_MAXNZ];
    for (int i = 0; i < 2 * nx; ++i) { sz[i] = c.sz[i]; lb[i] = c.lbz[i]; ub[i] = c.ubz[i]; }
    HILO_REQUIRE(nxa + (has_noise ? nx : 0) <= OCP_MAXNZ && nxa <= OCP_MAXNX + OCP_MAXNU, "model too large for parameter estimation in this build");
    memset(c.cost, 0, sizeof(c.cost));
    for (int i = 0; i < nx * nx; ++i) c.cost[o[0] + i] = d->Wx ? d->Wx[i] : 0.0;
    for (int i = 0; i < np * np; ++i) c.cost[o[1] + i] = d->Wp ? d->Wp[i] : 0.0;
    for (int i = 0; i < ny * ny; ++i) c.cost[o[2] + i] = d->Wy ? d->Wy[i] : 0.0;
    for (int i = 0; i < nx * nx; ++i) c.cost[o[3] + i] = d->Ww ? d->Ww[i] : 0.0;
    for (int i = 0; i < nu; ++i) c.cost[o[4] + i] = d->u_scaling ? d->u_scaling[i] : 1.0;
    if (gen && nrow_all > 0) {
      c.cost[o_nrow] = nrow_pt; c.cost[o_ncr] = d->n_con;
      for (int m = 0; m < OCP_MAXNC; ++m) { c.dlb[m] = -INFINITY; c.dub[m] = INFINITY; }
      for (int r = 0; r < nrow_pt; ++r) { c.cost[o_rowx + r] = row_expr[r]; c.cost[o_rref + r] = row_expr[r]; }
      for (int m = 0; m < nrow_all; ++m) {
        const int r = m % nrow_pt;
        c.dlb[m] = row_lb[r] > -INFINITY ? row_lb[r] - relax * fmax(1.0, fabs(row_lb[r])) : row_lb[r];
        c.dub[m] = row_ub[r] < INFINITY ? row_ub[r] + relax * fmax(1.0, fabs(row_ub[r])) : row_ub[r];
        c.row_ref[m] = (short)m;
      }
      c.nc = nrow_all; c.nc_term = 0; c.n_con_ref = nrow_all; c.n_tcon_ref = 0;   // compact multipliers; the output pass orders them
    }
    for (int i = 0; i < nx; ++i) {
      c.sz[i] = sz[i]; c.lbz[i] = lb[i]; c.ubz[i] = ub[i];
      if (has_noise) { c.sz[nxa + i] = sz[nx + i]; c.lbz[nxa + i] = lb[nx + i]; c.ubz[nxa + i] = ub[nx + i]; }
      c.x0_free_mask |= 1u << i;
    }
    for (int j = 0; j < np; ++j) {
      const double sp = d->p_scaling ? d->p_scaling[j] : 1.0;
      double pl = d->p_lb ? d->p_lb[j] / sp : -INFINITY, pu = d->p_ub ? d->p_ub[j] / sp : INFINITY;
      if (!(pl <= pu)) { delete h; return fail(HILO_EINVAL, "hilo_mhe_create: p_lb > p_ub for parameter %d", j); }
      const bool estimated = d->estimate_parameters && pl < pu;
      c.sz[nx + j] = sp;
      c.k0_only_mask |= 1u << (nx + j);                         // one box (and one barrier term) per parameter
      if (estimated) {
        c.x0_free_mask |= 1u << (nx + j);
        h->est_mask |= 1u << j;
        if (pl > -INFINITY) pl -= relax * fmax(1.0, fabs(pl));
        if (pu < INFINITY) pu += relax * fmax(1.0, fabs(pu));
        c.lbz[nx + j] = pl; c.ubz[nx + j] = pu;
      } else {
        c.lbz[nx + j] = -INFINITY; c.ubz[nx + j] = INFINITY;   // pinned through x_0 (IPOPT: fixed variable removed)
      }
    }
  }
  if (jit) {
    JitRequest rq;
    rq.user_source = d->user_source;
    rq.policy = JIT_MHE;
    rq.coll_d = D; rq.N = d->N;
    rq.sym = c.nsub == 1 && !getenv("HILO_NMPC_TAYLOR");
    rq.mhe_gen = gen; rq.mhe_noise = has_noise;
    rq.has_fun = d->n_con > 0; rq.nc = nrow_all;
    rc = jit_nmpc_kernels(rq, device, &h->jit);
    if (!rc && getenv("HILO_JIT_COMPILE_ONLY")) { delete h; return HILO_COMPILED_ONLY; }   // cache warmed, no handle
    if (!rc && (h->jit.dims[0] != nx || h->jit.dims[1] != nu || h->jit.dims[2] != np || h->jit.dims[3] != ny))
      rc = fail(HILO_EINVAL, "hilo_mhe_create: the compiled model has (nx, nu, np, ny) = (%d, %d, %d, %d); the description says "
                             "(%d, %d, %d, %d)", h->jit.dims[0], h->jit.dims[1], h->jit.dims[2], h->jit.dims[3], nx, nu, np, ny);
    if (!rc && gen && h->jit.dims[6] != nx + np)
      rc = fail(HILO_EINVAL, "hilo_mhe_create: the compiled general estimator has %d engine states, expected %d", h->jit.dims[6], nx + np);
    if (rc) { delete h; return rc; }
    h->lds_bytes = (size_t)h->jit.dims[5];
  }
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc((void**)&h->dev, sizeof(OcpConst));
  if (e == hipSuccess) e = hipMemcpy(h->dev, &c, sizeof(OcpConst), hipMemcpyHostToDevice);
  if ((ev || gen) && e == hipSuccess) {
    // tiled guess in engine layout: [x_guess | p_guess] per stage, w_guess (mhe.py:620, :633-649)
    const int nxa = nx + np, nve = (d->N + 1) * nxa + (has_noise ? d->N * nx : 0);
    double* g = new double[nve];
    for (int k = 0; k <= d->N; ++k) {
      for (int i = 0; i < nx; ++i) g[k * nxa + i] = (d->x_guess ? d->x_guess[i] : 0.0) / c.sz[i];
      for (int j = 0; j < np; ++j) g[k * nxa + nx + j] = (d->p_guess ? d->p_guess[j] : 0.0) / c.sz[nx + j];
    }
    for (int k = 0; k < d->N && has_noise; ++k)
      for (int i = 0; i < nx; ++i) g[(d->N + 1) * nxa + k * nx + i] = (d->w_guess ? d->w_guess[i] : 0.0) / c.sz[nxa + i];
    e = hipMalloc((void**)&h->v_guess_e, sizeof(double) * nve);
    if (e == hipSuccess) e = hipMemcpy(h->v_guess_e, g, sizeof(double) * nve, hipMemcpyHostToDevice);
    delete[] g;
  }
  // tiled guess row; the run-time compiled kernel reads every start row behind a parameter prefix ([p | x | w]), the
  // zoo kernels take the guess without one
  const int gpre = jit ? np : 0;
  const int nvf = gpre + (d->N + 1) * nx + d->N * nx;      // (read by the plain policies only; they always have the noise block)
  if (e == hipSuccess) e = hipMalloc((void**)&h->v_guess, sizeof(double) * nvf);
  if (e == hipSuccess) {
    double* g = new double[nvf];  // mhe.py:633-649: tiled guesses, scaled (mhe.py:229-236)
    for (int i = 0; i < gpre; ++i) g[i] = 0.0;
    for (int k = 0; k <= d->N; ++k)
      for (int i = 0; i < nx; ++i) g[gpre + k * nx + i] = (d->x_guess ? d->x_guess[i] : 0.0) / c.sz[i];
    for (int k = 0; k < d->N; ++k)
      for (int i = 0; i < nx; ++i) g[gpre + (d->N + 1) * nx + k * nx + i] = (d->w_guess ? d->w_guess[i] : 0.0) / c.sz[nx + i];
    e = hipMemcpy(h->v_guess, g, sizeof(double) * nvf, hipMemcpyHostToDevice);
    delete[] g;
  }
  if (e != hipSuccess) {
    hilo_mhe_destroy(h);
    return fail(HILO_EHIP, "hilo_mhe_create: %s", hipGetErrorString(e));
  }
  *out = h;
  return HILO_OK;
}

extern "C" int hilo_mhe_dims(const hilo_mhe* h, int* n_v, int* n_g, int* nx, int* nu, int* np, int* ny) {
  HILO_REQUIRE(h, "hilo_mhe_dims: NULL handle");
  if (n_v) *n_v = h->n_v;
  if (n_g) *n_g = h->n_g;
  if (nx) *nx = h->nx;
  if (nu) *nu = h->nu;
  if (np) *np = h->np;
  if (ny) *ny = h->ny;
  return HILO_OK;
}

extern "C" int hilo_mhe_reset_warm_start(hilo_mhe* h) {
  HILO_REQUIRE(h, "hilo_mhe_reset_warm_start: NULL handle");
  h->warm_valid = 0;
  return HILO_OK;
}

template <class PB>
static int mhe_launch_pb(hilo_mhe* h, int64_t batch, const double* v0, int64_t v0s, int prefix_in_v0, double* v_opt,
                         double* f_opt, double* lam_g, double* x_opt, int32_t* status, int32_t* iters, double* kkt,
                         hipStream_t s) {
  if (h->lds_bytes > 64 * 1024)
    HILO_HIP_CHECK(hipFuncSetAttribute((const void*)ocp_solve_kernel<PB, OCP_TPB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)h->lds_bytes));
  // v0 rows either carry the parameter prefix (user v0 / previous solution) or not (tiled guess, stride 0)
  const double* v0p = v0;
  hipLaunchKernelGGL((ocp_solve_kernel<PB, OCP_TPB>), dim3((unsigned)batch), dim3(OCP_TPB), h->lds_bytes, s, h->dev, batch,
                     (const double*)nullptr, h->par_buf, (int64_t)(h->np + h->nx), h->sd_buf,
                     (int64_t)((h->N + 1) * (h->nu + h->ny)), v0p, v0s, prefix_in_v0 ? h->np : 0, h->np, v_opt, f_opt, lam_g,
                     x_opt, 1, status, iters, kkt, (long long*)nullptr);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

template <class M>
static int mhe_launch(hilo_mhe* h, int64_t batch, const double* v0, int64_t v0s, int prefix_in_v0, double* v_opt,
                      double* f_opt, double* lam_g, double* x_opt, int32_t* status, int32_t* iters, double* kkt,
                      hipStream_t s) {
  if constexpr (MheNoise<M>::SYM_MHE) {
    if (h->host.nsub != 1 || getenv("HILO_NMPC_TAYLOR"))
      return mhe_launch_pb<MheNoise<M, false>>(h, batch, v0, v0s, prefix_in_v0, v_opt, f_opt, lam_g, x_opt, status, iters, kkt, s);
  }
  return mhe_launch_pb<MheNoise<M>>(h, batch, v0, v0s, prefix_in_v0, v_opt, f_opt, lam_g, x_opt, status, iters, kkt, s);
}

extern "C" int hilo_mhe_estimate(hilo_mhe* h, int64_t batch, const double* x_arrival, const double* p, int64_t p_stride,
                                 const double* u_meas, const double* y_meas, const double* v0, double* v_opt,
                                 double* f_opt, double* lam_g, double* x_opt, int32_t* status, int32_t* iters, double* kkt,
                                 void* stream) {
  HILO_REQUIRE(h, "hilo_mhe_estimate: NULL handle");
  HILO_REQUIRE(batch >= 0, "hilo_mhe_estimate: negative batch");
  if (batch == 0) return HILO_OK;
  HILO_REQUIRE(x_arrival && y_meas && v_opt && f_opt && x_opt && status && iters, "hilo_mhe_estimate: NULL argument");
  HILO_REQUIRE(h->nu == 0 || u_meas, "hilo_mhe_estimate: the model has %d inputs but u_meas is NULL", h->nu);
  HILO_REQUIRE(h->np == 0 || p, "hilo_mhe_estimate: the model has %d parameters but p is NULL", h->np);
  HILO_REQUIRE(p_stride == 0 || p_stride >= h->np, "hilo_mhe_estimate: p_stride %lld < np", (long long)p_stride);
  HILO_HIP_CHECK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const int wp = h->np + h->nx, ws = h->nu + h->ny;
  if (h->buf_batch != batch) {
    if (h->par_buf) HILO_HIP_CHECK(hipFree(h->par_buf));
    if (h->sd_buf) HILO_HIP_CHECK(hipFree(h->sd_buf));
    h->par_buf = h->sd_buf = nullptr;
    hipError_t e = hipMalloc((void**)&h->par_buf, sizeof(double) * (size_t)wp * batch);
    if (e == hipSuccess) e = hipMalloc((void**)&h->sd_buf, sizeof(double) * (size_t)(h->N + 1) * ws * batch);
    if (e != hipSuccess) return fail(HILO_ENOMEM, "MHE buffers: %s", hipGetErrorString(e));
    h->buf_batch = batch;
  }
  {
    const int64_t tot = batch * wp + batch * (int64_t)(h->N + 1) * ws;
    hipLaunchKernelGGL(mhe_pack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, batch, h->N, h->np, h->nx,
                       h->nu, h->ny, p, p_stride, x_arrival, u_meas, y_meas, h->par_buf, h->sd_buf);
    HILO_HIP_CHECK(hipGetLastError());
  }
  if (h->est || h->gen) {
    const int nxa = h->nx + h->np, nve = (h->N + 1) * nxa + (h->noise ? h->N * h->nx : 0);
    if (!h->ve || h->warm_batch != batch) {
      double** bufs[] = {&h->x0e, &h->v0e, &h->ve, &h->lame, &h->v_warm};
      const size_t sizes[] = {(size_t)nxa, (size_t)nve, (size_t)nve, (size_t)h->N * (nxa + h->nc), (size_t)h->n_v};
      for (int q = 0; q < 5; ++q) {
        if (*bufs[q]) HILO_HIP_CHECK(hipFree(*bufs[q]));
        *bufs[q] = nullptr;
        hipError_t e = hipMalloc((void**)bufs[q], sizeof(double) * sizes[q] * batch);
        if (e != hipSuccess) return fail(HILO_ENOMEM, "MHE buffers: %s", hipGetErrorString(e));
      }
      h->warm_batch = batch;
      h->warm_valid = 0;
    }
    // x0e = [unused | p]: the pinned parameter slots take their value from here; par = [x_arrival | p (arrival)]
    {
      int rc2 = mhe_est_pack(batch, h->nx, h->np, p, p_stride, x_arrival, h->x0e, h->par_buf, s);
      if (rc2) return rc2;
    }
    const double* start = h->v_guess_e;
    int64_t stride = 0;
    const double* vref = v0 ? v0 : (h->warm_valid ? h->v_warm : nullptr);
    if (vref) {
      int rc2 = mhe_est_convert_in(batch, h->N, h->nx, h->np, vref, h->n_v, h->v0e, s, h->noise);
      if (rc2) return rc2;
      start = h->v0e;
      stride = nve;
    }
    MheEstArgs a{h->dev, batch, h->x0e, h->par_buf, h->sd_buf, (int64_t)((h->N + 1) * ws), start, stride, h->ve, f_opt, h->lame,
                 status, iters, kkt, h->lds_bytes, s};
    int rc2;
    if (h->gen) {
      // the general run-time compiled policy: same engine layout; its output pass (the module's `coll_out` kernel, hilo_mhe_policy.h::
      // mhe_gen_output) writes the reference's v / lam_g with the collocation block and x_opt (handed over in the `par` argument)
      rc2 = jit_launch_solve(h->jit.solve, h->dev, batch, h->x0e, h->par_buf, (int64_t)nxa, h->sd_buf, (int64_t)((h->N + 1) * ws), start,
                             stride, h->ve, f_opt, h->lame, nullptr, status, iters, kkt, nullptr, nullptr, s, OcpExtra());
      if (!rc2)
        rc2 = jit_launch_coll_out(h->jit.coll_out, h->dev, batch, h->N, h->ve, h->lame, x_opt, 0, h->sd_buf,
                                  (int64_t)((h->N + 1) * ws), v_opt, lam_g, s);
    } else {
      rc2 = h->est->launch(a);
      if (!rc2) rc2 = mhe_est_convert_out(h->dev, batch, h->N, h->nx, h->np, h->ve, h->lame, v_opt, lam_g, x_opt, s);
    }
    if (rc2) return rc2;
    HILO_HIP_CHECK(hipMemcpyAsync(h->v_warm, v_opt, sizeof(double) * h->n_v * batch, hipMemcpyDeviceToDevice, s));
    h->warm_valid = 1;
    return HILO_OK;
  }
  const double* vstart = v0;
  int64_t vstride = h->n_v;
  int prefix = 1;
  if (!vstart) {
    if (h->warm_valid && h->warm_batch == batch) vstart = h->v_warm;  // mhe.py:385
    else { vstart = h->v_guess; vstride = 0; prefix = 0; }
  }
  int rc = HILO_ENOTSUP;
  if (h->use_jit) {
    double *vdst = v_opt, *ldst = lam_g;
    if (h->coll_d) {   // the engine's result goes to buffers of the handle; the output pass writes the reference's layout
      if (!h->vc || h->vc_batch != batch || !h->lamc) {
        if (h->vc) HILO_HIP_CHECK(hipFree(h->vc));
        if (h->lamc) HILO_HIP_CHECK(hipFree(h->lamc));
        h->vc = h->lamc = nullptr;
        hipError_t e = hipMalloc((void**)&h->vc, sizeof(double) * (size_t)h->n_vc * batch);
        if (e == hipSuccess) e = hipMalloc((void**)&h->lamc, sizeof(double) * (size_t)h->N * h->nx * batch);
        if (e != hipSuccess) return fail(HILO_ENOMEM, "MHE buffers: %s", hipGetErrorString(e));
        h->vc_batch = batch;
      }
      vdst = h->vc; ldst = h->lamc;
    }
    // every start row carries the parameter prefix (the tiled guess has a dummy one): v0_prefix = v_prefix = np in the kernel
    rc = jit_launch_solve(h->jit.solve, h->dev, batch, nullptr, h->par_buf, (int64_t)wp, h->sd_buf,
                          (int64_t)((h->N + 1) * ws), vstart, vstride, vdst, f_opt, ldst, x_opt, status, iters, kkt,
                          nullptr, nullptr, s, OcpExtra());
    if (!rc && h->coll_d)
      rc = jit_launch_coll_out(h->jit.coll_out, h->dev, batch, h->N, h->vc, h->lamc, h->par_buf, (int64_t)wp, h->sd_buf,
                               (int64_t)((h->N + 1) * ws), v_opt, lam_g, s);
  } else
  switch (h->model_id) {
#define X(ID, T) case ID: rc = mhe_launch<T>(h, batch, vstart, vstride, prefix, v_opt, f_opt, lam_g, x_opt, status, iters, kkt, s); break;
    HILO_MHE_MODELS(X)
#undef X
  }
  if (rc) return rc;
  // the parameter prefix of v_opt: the pinned values (mhe.py:614-623 with p_lb = p_ub)
  if (h->np > 0)
    HILO_HIP_CHECK(hipMemcpy2DAsync(v_opt, sizeof(double) * h->n_v, h->par_buf, sizeof(double) * wp,
                                    sizeof(double) * h->np, batch, hipMemcpyDeviceToDevice, s));
  if (h->warm_batch != batch) {
    if (h->v_warm) HILO_HIP_CHECK(hipFree(h->v_warm));
    h->v_warm = nullptr;
    hipError_t e = hipMalloc((void**)&h->v_warm, sizeof(double) * h->n_v * batch);
    if (e != hipSuccess) return fail(HILO_ENOMEM, "warm-start buffer: %s", hipGetErrorString(e));
    h->warm_batch = batch;
  }
  HILO_HIP_CHECK(hipMemcpyAsync(h->v_warm, v_opt, sizeof(double) * h->n_v * batch, hipMemcpyDeviceToDevice, s));
  h->warm_valid = 1;
  return HILO_OK;
}
