// Batched direct-multiple-shooting NMPC on the generic stage-structured interior-point engine (hilo_ocp.h) + C ABI.
//
// Replaces, for a pre-discretised model with `integration_method='discrete'`, the solver object the
// reference builds with `ca.nlpsol('solver','ipopt',{'f','x','p','g'})` (hilo_mpc/modules/controller/mpc.py:1778-1787)
// and calls once per step from `NMPC._optimize` (mpc.py:722).  Transcription restated from mpc.py:1455-1787:
//   v = [x_0..x_N | u_0..u_{N-1}] (scaled),  g_k = x_{k+1} - Phi(x_k,u_k) = 0,  J = sum_k l(x_k,u_k) + V(x_N),
//   x_0 pinned (mpc.py:797-802; removed from the variables like IPOPT's make_parameter), box bounds on x, u,
//   quadratic costs of QuadraticCost (hilo_mpc/util/modeling.py:243-283) on the scaled variables, the input-change
//   term only in interval 0 (mpc.py:1631-1635), the model scaled as in hilo_mpc/modules/base.py:1562-1591.
#include <stdlib.h>
#include <string.h>

#include "hilo_nmpc_gen.h"
#include "hilo_nmpc_track.h"

namespace hilo {

// ---- closed-loop helper for benchmarks / tests: x+ = Phi(x, u) with the controller's own shooting map ------------
template <class M>
__global__ void plant_step_kernel(const OcpConst* __restrict__ pcg, int64_t batch, const double* __restrict__ x,
                                  const double* __restrict__ u, const double* __restrict__ par, int64_t par_stride,
                                  double* __restrict__ xn) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  constexpr int NX = M::NX, NU = M::NU, NP = M::NP;
  double xv[NX], uv[NU > 0 ? NU : 1], pv[NP > 0 ? NP : 1], xo[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) xv[i] = x[b * NX + i];
#pragma unroll
  for (int i = 0; i < NU; ++i) uv[i] = u[b * NU + i];
#pragma unroll
  for (int i = 0; i < NP; ++i) pv[i] = par[b * par_stride + i];
  if constexpr (model_has_ext<M>::value) {
    const GpExt ext{pcg->ext, nullptr, 1, 0, 0, false};  // every lane sums its own kernel row
    model_step<M>(pcg->order, pcg->nsub, xv, uv, pv, pcg->dt, xo, ext);
  } else {
    model_step<M>(pcg->order, pcg->nsub, xv, uv, pv, pcg->dt, xo);
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) xn[b * NX + i] = xo[i];
}

}  // namespace hilo

using namespace hilo;

#include "hilo_nmpc_handle.h"

#define HILO_NMPC_MODELS(X)              \
  X(HILO_MODEL_CHEMOSTAT4, Chemostat4)   \
  X(HILO_MODEL_PENDULUM4, Pendulum4)     \
  X(HILO_MODEL_BIOREACTOR3, Bioreactor3) \
  X(HILO_MODEL_CHEMOSTAT4_GP, Chemostat4Gp) \
  X(HILO_MODEL_ROBOT6, Robot6)           \
  X(HILO_MODEL_CSTR3, Cstr3)

static int nmpc_model_dims(int id, int* nx, int* nu, int* np, size_t* lds, int N) {
  *lds = 0;
  switch (id) {
#define X(ID, T) case ID: *nx = T::NX; *nu = T::NU; *np = T::NP; *lds = Ocp<NmpcTrack<T>>::lds_doubles(N) * sizeof(double); return HILO_OK;
    HILO_NMPC_MODELS(X)
#undef X
  }
  return fail(HILO_ENOTSUP, "model id %d has no NMPC instantiation in this build", id);
}

extern "C" void hilo_nmpc_destroy(hilo_nmpc* h) {
  if (!h) return;
  if (h->dev) (void)hipFree(h->dev);
  if (h->v_guess) (void)hipFree(h->v_guess);
  if (h->v_warm) (void)hipFree(h->v_warm);
  if (h->par_buf) (void)hipFree(h->par_buf);
  if (h->prof) (void)hipFree(h->prof);
  if (h->ext_pack) (void)hipFree(h->ext_pack);
  for (double* g : h->user_gp_pack)
    if (g) (void)hipFree(g);
  jit_unload(&h->jit);
  if (h->ws) (void)hipFree(h->ws);
  if (h->vc) (void)hipFree(h->vc);
  if (h->lamc) (void)hipFree(h->lamc);
  delete h;
}

static void copy_or(double* dst, const double* src, int n, double dflt) {
  for (int i = 0; i < n; ++i) dst[i] = src ? src[i] : dflt;
}

extern "C" int hilo_nmpc_create(const hilo_nmpc_desc* d, int device, hilo_nmpc** out) {
  HILO_REQUIRE(d && out, "hilo_nmpc_create: NULL argument");
  HILO_REQUIRE(d->N >= 1 && d->N <= 512, "hilo_nmpc_create: horizon %d out of range [1, 512]", d->N);
  HILO_REQUIRE(d->dt > 0.0, "hilo_nmpc_create: dt must be positive");
  // ---- run-time compiled problems (hilo_jit.hip) ----
  const bool jit = d->user_source != nullptr;
  if (jit && d->user_policy == JIT_USER) return nmpc_user_create(d, device, out);
  HILO_REQUIRE(!jit || d->user_policy == JIT_TRACK, "hilo_nmpc_create: user_policy must be 0 (tracking) or 2 (general)");
  HILO_REQUIRE(jit || d->model_id != HILO_MODEL_USER, "hilo_nmpc_create: HILO_MODEL_USER needs desc.user_source");
  if (d->Nc != 0 && d->Nc != d->N)
    return fail(HILO_ENOTSUP, "control horizon (%d) != prediction horizon (%d) needs the run-time compiled general policy "
                              "(desc.user_source with user_policy 2)", d->Nc, d->N);
  int nx, nu, np;
  size_t lds;
  int rc = HILO_OK;
  if (d->model_id == HILO_MODEL_USER) {
    nx = d->user_nx; nu = d->user_nu; np = d->user_np;
    HILO_REQUIRE(nx >= 1 && nu >= 0 && np >= 0, "hilo_nmpc_create: bad user model dimensions");
  } else {
    rc = nmpc_model_dims(d->model_id, &nx, &nu, &np, &lds, d->N);
    if (rc) return rc;
  }
  HILO_REQUIRE(nx <= OCP_MAXNX && nu <= OCP_MAXNU, "model too large for this build");
  bool jit_big = false;
  if (jit) {   // the tracking policy compiled for the user's functor: footprint from the dimensions
    if (d->n_path_var > 0 || d->n_con > 0 || d->n_tcon > 0 || d->collocation_degree > 0 || d->time_varying || d->learned)   // (learned terms of a user model: desc.user_gp)
      return fail(HILO_ENOTSUP, "a run-time compiled model with path following, constraints, collocation, per-stage data or a "
                                "learned term needs user_policy 2");
    int ncost = (nx + nu) * (nx + nu) + (nx + nu) + nx * nx + nx + nu * nu + 1;   // NmpcTrack<M>::NCOST ...
    if (ncost + (nx + nu) * (nx + nu) + nu * nu <= OCP_NCOST) ncost += (nx + nu) * (nx + nu) + nu * nu;   // ... with the symmetrised tables (SYMTAB)
    const int nconst = (int)((__builtin_offsetof(OcpConst, cost) + sizeof(double) * ncost + 7) / 8);
    const size_t fixed = ocp_fixed_doubles(nx, nu, nconst, np + nu, 0, 0, d->N) * sizeof(double);
    lds = fixed + ocp_iter_doubles(nx, nu, 0, d->N) * sizeof(double);
    if (lds > 160 * 1024) { jit_big = true; lds = fixed; }
  }
  // ---- general problems: path variable and / or nonlinear stage constraint ----
  HILO_REQUIRE(d->n_path_var >= 0 && d->n_path_var <= 1, "hilo_nmpc_create: at most one path variable is supported (got %d)",
               d->n_path_var);
  HILO_REQUIRE(d->n_con >= 0 && d->n_tcon >= 0, "hilo_nmpc_create: negative constraint count");
  if (d->n_con > GEN_NEXPR || d->n_tcon > GEN_NEXPR)   // capacity of the precompiled variants, not of the method
    return fail(HILO_ENOTSUP, "the precompiled variants hold at most %d stage / terminal constraint expressions (got %d / %d); "
                              "compile the problem at run time (desc.user_source, user_policy 2)", GEN_NEXPR, d->n_con, d->n_tcon);
  const bool general = d->n_path_var > 0 || d->n_con > 0 || d->n_tcon > 0;
  const GenVariant* gv = nullptr;
  int nth = 0, ne = 0, nrow = 0, n_con_ref = 0;
  int row_expr[OCP_MAXNC], row_sign[OCP_MAXNC], row_e[OCP_MAXNC], row_ref[OCP_MAXNC];
  int ntrow = 0, n_tcon_ref = 0, ne_stage = 0;   // terminal rows of the engine / of the reference's g; slacks of the stage constraint
  int trow_expr[OCP_MAXNC], trow_sign[OCP_MAXNC], trow_e[OCP_MAXNC], trow_ref[OCP_MAXNC];
  double trow_lb[OCP_MAXNC], trow_ub[OCP_MAXNC];
  double row_lb[OCP_MAXNC], row_ub[OCP_MAXNC];
  if (general) {
    if (d->learned) return fail(HILO_ENOTSUP, "a learned term together with path following / stage constraints is not built");
    nth = d->n_path_var;
    if (d->n_con > 0) {
      HILO_REQUIRE(d->con_prog && d->con_prog_len > 0, "hilo_nmpc_create: n_con > 0 but no constraint program");
      ne = d->con_soft ? d->n_con : 0;
      n_con_ref = d->con_soft ? 2 * d->n_con : d->n_con;   // rows per stage in the reference's g (mpc.py:1711-1712)
      for (int j = 0; j < d->n_con; ++j) {
        const double lb = d->con_lb ? d->con_lb[j] : -INFINITY, ub = d->con_ub ? d->con_ub[j] : INFINITY;
        HILO_REQUIRE(lb <= ub, "hilo_nmpc_create: constraint %d has lb > ub", j);
        if (d->con_soft) {   // c - e <= ub | -c - e <= -lb; a row without a finite bound constrains nothing and is dropped
          if (ub < INFINITY) {
            HILO_REQUIRE(nrow < OCP_MAXNC, "too many constraint rows");
            row_expr[nrow] = j; row_sign[nrow] = 1; row_e[nrow] = j; row_lb[nrow] = -INFINITY; row_ub[nrow] = ub;
            row_ref[nrow++] = j;
          }
          if (lb > -INFINITY) {
            HILO_REQUIRE(nrow < OCP_MAXNC, "too many constraint rows");
            row_expr[nrow] = j; row_sign[nrow] = -1; row_e[nrow] = j; row_lb[nrow] = -INFINITY; row_ub[nrow] = -lb;
            row_ref[nrow++] = d->n_con + j;
          }
        } else if (lb > -INFINITY || ub < INFINITY) {
          HILO_REQUIRE(nrow < OCP_MAXNC, "too many constraint rows");
          row_expr[nrow] = j; row_sign[nrow] = 1; row_e[nrow] = -1; row_lb[nrow] = lb; row_ub[nrow] = ub;
          row_ref[nrow++] = j;
        }
      }
    }
    ne_stage = ne;
    if (d->n_tcon > 0) {
      HILO_REQUIRE(d->tcon_prog && d->tcon_prog_len > 0, "hilo_nmpc_create: n_tcon > 0 but no terminal constraint program");
      if (d->tcon_soft) {
        if (ne_stage > 0 || d->n_tcon != 1)
          return fail(HILO_ENOTSUP, "one shared slack in this build: a soft terminal constraint needs n_tcon = 1 and a hard (or "
                                    "no) stage constraint");
        ne += d->n_tcon;
      }
      n_tcon_ref = d->tcon_soft ? 2 * d->n_tcon : d->n_tcon;   // mpc.py:1687-1690 / :1696-1698
      for (int j = 0; j < d->n_tcon; ++j) {
        const double lb = d->tcon_lb ? d->tcon_lb[j] : -INFINITY, ub = d->tcon_ub ? d->tcon_ub[j] : INFINITY;
        HILO_REQUIRE(lb <= ub, "hilo_nmpc_create: terminal constraint %d has lb > ub", j);
        auto add = [&](int sign, int e, double rlb, double rub, int ref) {
          trow_expr[ntrow] = j; trow_sign[ntrow] = sign; trow_e[ntrow] = e; trow_lb[ntrow] = rlb; trow_ub[ntrow] = rub;
          trow_ref[ntrow++] = ref;
        };
        HILO_REQUIRE(nrow + ntrow + (d->tcon_soft ? (ub < INFINITY) + (lb > -INFINITY) : 1) <= OCP_MAXNC, "too many constraint rows");
        if (d->tcon_soft) {   // same row pair as the soft stage constraint; the slack index counts after the stage slacks
          if (ub < INFINITY) add(1, ne_stage + j, -INFINITY, ub, j);
          if (lb > -INFINITY) add(-1, ne_stage + j, -INFINITY, -lb, d->n_tcon + j);
        } else {
          add(1, -1, lb, ub, j);
        }
      }
    }
    gv = nmpc_gen_find(d->model_id, nth, ne, nrow + ntrow, d->N);
    if (!gv)
      return fail(HILO_ENOTSUP, "no device instantiation for model %d with %d path variable(s), %d shared slack(s) and %d "
                                "inequality row(s) per stage at horizon %d in this build", d->model_id, nth, ne, nrow + ntrow, d->N);
    lds = gv->lds_bytes(d->N);
  }
  const CollVariant* cv = nullptr;
  if (d->collocation_degree > 0) {
    if (general || d->learned)
      return fail(HILO_ENOTSUP, "collocation together with path following / stage constraints / learned terms is not built");
    HILO_REQUIRE(d->coll_A && d->coll_D, "hilo_nmpc_create: collocation needs the basis (coll_A, coll_D)");
    cv = nmpc_coll_find(d->model_id, d->collocation_degree);
    if (!cv)
      return fail(HILO_ENOTSUP, "no collocation instantiation for model %d with degree %d in this build (degree 3 for the "
                                "continuous zoo models, 1-4 for chemostat4)", d->model_id, d->collocation_degree);
    lds = cv->lds_bytes(d->N);
  }
  const TvVariant* tvv = nullptr;
  if (d->time_varying) {
    if (general || d->learned || cv)
      return fail(HILO_ENOTSUP, "per-stage references / parameters together with path following, stage constraints, learned "
                                "terms or collocation are not built");
    tvv = nmpc_tv_find(d->model_id);
    if (!tvv) return fail(HILO_ENOTSUP, "no per-stage-data instantiation for model %d in this build", d->model_id);
    lds = tvv->lds_bytes(d->N);
  }
  const TrackBigVariant* bigv = nullptr;
  if (lds > 160 * 1024 && !general && !cv && !tvv && !d->learned && !jit) {
    bigv = nmpc_track_big_find(d->model_id);   // long horizon: iterate in a global-memory workspace
    if (bigv) lds = bigv->lds_bytes(d->N);
  }
  if (lds > 160 * 1024)
    return fail(HILO_ENOTSUP, "horizon %d needs %zu B of LDS per instance (limit 163840)", d->N, lds);
  // engine dimensions: [model x | theta | e], [model u | u_theta]
  const int nxe = gv ? gv->nx : nx, nue = gv ? gv->nu : nu, nxv = gv ? gv->nxv : nx;
  const int nz = nxe + nue;
  hilo_nmpc* h = new hilo_nmpc();
  memset(h, 0, sizeof(*h));
  h->device = device; h->model_id = d->model_id; h->nx = nx; h->nu = nu; h->np = np; h->N = d->N;
  h->gen = gv; h->nu_out = nu;
  h->jit_policy = -1;
  h->nxe = nxe; h->nue = nue; h->nxv = nxv; h->ntail = ne; h->Nc = d->N;
  h->coll = cv;
  h->tv = tvv;
  h->big = bigv;
  h->n_vc = (d->N + 1) * nx + d->N * nu;
  const int dn = cv ? cv->degree * nx : 0;
  h->n_v = (d->N + 1) * nxv + d->N * nue + ne + d->N * dn;   // mpc.py:1440-1443 (+ the soft-constraint slack, :1529-1537)
  h->n_g = d->N * (nxv + n_con_ref + dn) + (general ? n_tcon_ref : 0);   // mpc.py:1657-1669, :1684-1725
  h->lds_bytes = lds;
  OcpConst& c = h->host;
  memset(&c, 0, sizeof(c));
  ocp_default_options(c);
  c.N = d->N; c.Nc = d->N; c.order = d->erk_order >= 1 ? d->erk_order : 4; c.nsub = d->n_sub >= 1 ? d->n_sub : 1;
  c.dt = d->dt;
  c.flags = 1;  // lam_g in the reference's convention (terminal cost on Phi_{N-1}, mpc.py:1682)
  if (cv) {
    c.coll.d = cv->degree;
    for (int i = 0; i < cv->degree * cv->degree; ++i) c.coll.A[i] = d->coll_A[i];
    for (int i = 0; i <= cv->degree; ++i) c.coll.Dc[i] = d->coll_D[i];
  }
  if (d->max_iter > 0) c.max_iter = d->max_iter;
  if (d->acceptable_iter > 0) c.acceptable_iter = d->acceptable_iter;
  double sx[OCP_MAXNX], su[OCP_MAXNU];
  copy_or(sx, d->x_scaling, nx, 1.0);
  copy_or(su, d->u_scaling, nu, 1.0);
  for (int i = 0; i < nz; ++i) c.sz[i] = 1.0;   // path variable, shared slack, virtual input: unit scaling (mpc.py:1200-1201)
  for (int i = 0; i < nx; ++i) c.sz[i] = sx[i];
  for (int i = 0; i < nu; ++i) c.sz[nxe + i] = su[i];
  const double relax = d->bound_relax_factor >= 0.0 ? d->bound_relax_factor : 1e-8;
  c.bound_relax = relax;
  auto relaxed_lb = [&](double lb) { return lb > -INFINITY ? lb - relax * fmax(1.0, fabs(lb)) : lb; };
  auto relaxed_ub = [&](double ub) { return ub < INFINITY ? ub + relax * fmax(1.0, fabs(ub)) : ub; };
  if (!gv) {
    // cost block layout of NmpcTrack<M>: [Wz | zref | WN | xrefN | Wdu | has_du]
    double* q = c.cost;
    for (int i = 0; i < nz * nz; ++i) *q++ = d->Wz ? d->Wz[i] : 0.0;
    for (int i = 0; i < nz; ++i) *q++ = d->zref ? d->zref[i] : 0.0;
    for (int i = 0; i < nx * nx; ++i) *q++ = d->WN ? d->WN[i] : 0.0;
    for (int i = 0; i < nx; ++i) *q++ = d->xrefN ? d->xrefN[i] : 0.0;
    for (int i = 0; i < nu * nu; ++i) *q++ = d->Wdu ? d->Wdu[i] : 0.0;
    *q++ = d->Wdu ? 1.0 : 0.0;
    if ((q - c.cost) + nz * nz + nu * nu <= OCP_NCOST) {   // NmpcTrack<M>::SYMTAB: Sz = Wz + Wz^T, Sdu = Wdu + Wdu^T
      for (int i = 0; i < nz; ++i)
        for (int j = 0; j < nz; ++j) *q++ = d->Wz ? d->Wz[i * nz + j] + d->Wz[j * nz + i] : 0.0;
      for (int i = 0; i < nu; ++i)
        for (int j = 0; j < nu; ++j) *q++ = d->Wdu ? d->Wdu[i * nu + j] + d->Wdu[j * nu + i] : 0.0;
    }
  } else {
    // cost block layout of NmpcGen (hilo_nmpc_gen.h); model z index -> engine z index
    auto ez = [&](int i) { return i < nx ? i : nxe + (i - nx); };
    const int mz = nx + nu;
    for (int i = 0; i < mz; ++i) {
      for (int j = 0; j < mz; ++j) c.cost[gv->o_wz + ez(i) * nz + ez(j)] = d->Wz ? d->Wz[i * mz + j] : 0.0;
      c.cost[gv->o_zref + ez(i)] = d->zref ? d->zref[i] : 0.0;
    }
    for (int i = 0; i < nx; ++i) {
      for (int j = 0; j < nx; ++j) c.cost[gv->o_wn + i * nxe + j] = d->WN ? d->WN[i * nx + j] : 0.0;
      c.cost[gv->o_xrefn + i] = d->xrefN ? d->xrefN[i] : 0.0;
    }
    for (int i = 0; i < nu * nu; ++i) c.cost[gv->o_wdu + i] = d->Wdu ? d->Wdu[i] : 0.0;
    c.cost[gv->o_hasdu] = d->Wdu ? 1.0 : 0.0;
    if (nth && d->has_u_pf_ref) {   // mpc.py:1202-1204
      const int iu = nxe + nu;
      c.cost[gv->o_wz + iu * nz + iu] = d->u_pf_weight;
      c.cost[gv->o_zref + iu] = d->u_pf_ref;
    }
    for (int a = 0; a < ne_stage; ++a)    // e^T W e once per stage (mpc.py:1708), W = 1e4 I by default (modeling.py:875)
      for (int b = 0; b < ne_stage; ++b)
        c.cost[gv->o_wz + (nx + nth + a) * nz + (nx + nth + b)] = d->con_weight ? d->con_weight[a * ne_stage + b] : (a == b ? 1e4 : 0.0);
    for (int a = ne_stage; a < ne; ++a)   // e_T^T W e_T once (mpc.py:1686): a terminal weight on the constant state e_T
      for (int b = ne_stage; b < ne; ++b)
        c.cost[gv->o_wn + (nx + nth + a) * nxe + (nx + nth + b)] =
            d->tcon_weight ? d->tcon_weight[(a - ne_stage) * d->n_tcon + (b - ne_stage)] : (a == b ? 1e4 : 0.0);
    int rcode = HILO_OK;
    const char* why = "";
    int plen = 0;
    if (nth) {
      if (d->n_path_stage < 0 || d->n_path_stage > GEN_NPT || d->n_path_term < 0 || d->n_path_term > GEN_NPT)
        rcode = fail(HILO_EINVAL, "hilo_nmpc_create: at most %d path terms per cost", GEN_NPT);
      else if ((d->n_path_stage + d->n_path_term > 0) && (!d->path_prog || d->path_prog_len <= 0))
        rcode = fail(HILO_EINVAL, "hilo_nmpc_create: path terms without reference programs");
      else if (d->n_path_stage + d->n_path_term > 0 &&
               expr_check(d->path_prog, d->path_prog_len, d->n_path_stage + d->n_path_term, nxv, 1, np, &why))
        rcode = fail(HILO_EINVAL, "hilo_nmpc_create: path reference program: %s", why);
      if (!rcode) {
        c.cost[gv->o_nps] = d->n_path_stage; c.cost[gv->o_npt] = d->n_path_term;
        for (int a = 0; a < d->n_path_stage; ++a) {
          c.cost[gv->o_idxs + a] = d->path_stage_idx[a];
          if (d->path_stage_idx[a] < 0 || d->path_stage_idx[a] >= nx) rcode = fail(HILO_EINVAL, "path term on state %d", d->path_stage_idx[a]);
          for (int b = 0; b < d->n_path_stage; ++b) c.cost[gv->o_ws + a * GEN_NPT + b] = d->path_stage_W[a * d->n_path_stage + b];
        }
        for (int a = 0; a < d->n_path_term; ++a) {
          c.cost[gv->o_idxt + a] = d->path_term_idx[a];
          if (d->path_term_idx[a] < 0 || d->path_term_idx[a] >= nx) rcode = fail(HILO_EINVAL, "path term on state %d", d->path_term_idx[a]);
          for (int b = 0; b < d->n_path_term; ++b) c.cost[gv->o_wt + a * GEN_NPT + b] = d->path_term_W[a * d->n_path_term + b];
        }
        plen = d->n_path_stage + d->n_path_term > 0 ? d->path_prog_len : 0;
      }
    }
    if (!rcode && d->n_con > 0 && expr_check(d->con_prog, d->con_prog_len, d->n_con, nx, nu, np, &why))
      rcode = fail(HILO_EINVAL, "hilo_nmpc_create: constraint program: %s", why);
    if (!rcode && d->n_tcon > 0 && expr_check(d->tcon_prog, d->tcon_prog_len, d->n_tcon, nx, nu, np, &why))
      rcode = fail(HILO_EINVAL, "hilo_nmpc_create: terminal constraint program: %s", why);
    if (!rcode && gv->o_prog + plen + (d->n_con > 0 ? d->con_prog_len : 0) + (d->n_tcon > 0 ? d->tcon_prog_len : 0) > OCP_NCOST)
      rcode = fail(HILO_ENOTSUP, "expression programs too long (%d doubles available)", OCP_NCOST - gv->o_prog);
    if (rcode) { delete h; return rcode; }
    for (int i = 0; i < plen; ++i) c.cost[gv->o_prog + i] = d->path_prog[i];
    if (d->n_con > 0)
      for (int i = 0; i < d->con_prog_len; ++i) c.cost[gv->o_prog + plen + i] = d->con_prog[i];
    {
      const int off = gv->o_prog + plen + (d->n_con > 0 ? d->con_prog_len : 0);
      for (int i = 0; i < (d->n_tcon > 0 ? d->tcon_prog_len : 0); ++i) c.cost[off + i] = d->tcon_prog[i];
    }
    c.cost[gv->o_nexpr] = d->n_con;
    c.cost[gv->o_ntexpr] = d->n_tcon;
    c.nc = nrow;
    c.nc_term = ntrow;
    c.n_con_ref = n_con_ref;
    c.n_tcon_ref = n_tcon_ref;
    c.cost[gv->o_tsoft] = d->tcon_soft ? 1.0 : 0.0;
    for (int m = 0; m < OCP_MAXNC; ++m) { c.dlb[m] = -INFINITY; c.dub[m] = INFINITY; }
    for (int m = 0; m < nrow; ++m) {
      c.cost[gv->o_rowx + m] = row_expr[m]; c.cost[gv->o_rows + m] = row_sign[m]; c.cost[gv->o_rowe + m] = row_e[m];
      c.dlb[m] = relaxed_lb(row_lb[m]); c.dub[m] = relaxed_ub(row_ub[m]);   // IPOPT relaxes constraint bounds alike
      c.row_ref[m] = (short)row_ref[m];
    }
    for (int r = 0; r < ntrow; ++r) {
      c.cost[gv->o_trowx + r] = trow_expr[r]; c.cost[gv->o_trows + r] = trow_sign[r]; c.cost[gv->o_trowe + r] = trow_e[r];
      c.dlb[nrow + r] = relaxed_lb(trow_lb[r]); c.dub[nrow + r] = relaxed_ub(trow_ub[r]);
      c.trow_ref[r] = (short)trow_ref[r];
    }
    for (int i = nx; i < nxe; ++i) c.x0_free_mask |= 1u << i;               // theta_0 and e are variables (mpc.py:785-789)
    for (int a = 0; a < ne; ++a) c.k0_only_mask |= 1u << (nx + nth + a);    // one box on the shared slack
  }
  for (int i = 0; i < nz; ++i) {
    // bounds arrive in original units; scaled like mpc.py:253-259, then relaxed like IPOPT's bound_relax_factor
    double lb = -INFINITY, ub = INFINITY;
    if (i < nx) { if (d->x_lb) lb = d->x_lb[i] / c.sz[i]; if (d->x_ub) ub = d->x_ub[i] / c.sz[i]; }
    else if (i < nx + nth) { lb = d->theta_lb; ub = d->theta_ub; }                                  // mpc.py:1198-1199
    else if (i < nxe) {                                                                             // :1533-1534, :1544-1545
      const int a = i - nx - nth;
      lb = 0.0;
      ub = a < ne_stage ? (d->con_max_violation ? d->con_max_violation[a] : INFINITY)
                        : (d->tcon_max_violation ? d->tcon_max_violation[a - ne_stage] : INFINITY);
    }
    else if (i < nxe + nu) { const int j = i - nxe; if (d->u_lb) lb = d->u_lb[j] / c.sz[i]; if (d->u_ub) ub = d->u_ub[j] / c.sz[i]; }
    else { lb = d->u_pf_lb; ub = d->u_pf_ub; }                                                      // mpc.py:1196-1197
    lb = relaxed_lb(lb); ub = relaxed_ub(ub);
    if (!(lb < ub)) { delete h; return fail(HILO_EINVAL, "hilo_nmpc_create: empty box for variable %d", i); }
    c.lbz[i] = lb; c.ubz[i] = ub;
  }
  if (d->tol > 0) c.tol = d->tol;
  if (d->acceptable_tol > 0) c.acceptable_tol = d->acceptable_tol;
  if (d->mu_init > 0) c.mu_init = d->mu_init;
  if (d->max_hessian_perturbation > 0) c.delta_w_max = d->max_hessian_perturbation;
  hipError_t e = hipSetDevice(device);
  if (d->model_id == HILO_MODEL_CHEMOSTAT4_GP) {
    // dynamic_model.py:3040-3125: the label `mu` is replaced by the posterior mean over the features (S, I)
    if (!d->learned) {
      hilo_nmpc_destroy(h);
      return fail(HILO_EINVAL, "hilo_nmpc_create: model 'chemostat4_gp' needs desc.learned (a hilo_gp over the features S, I)");
    }
    rc = gp_pack_se2(d->learned, &h->ext_pack);
    if (rc) { hilo_nmpc_destroy(h); return rc; }
    {
      double nterms = 0.0;   // the solve kernel stages the table in LDS (hilo_models.h GP2_MAXN)
      HILO_HIP_CHECK(hipMemcpy(&nterms, h->ext_pack, sizeof(double), hipMemcpyDeviceToHost));
      if (nterms > GP2_MAXN) {
        hilo_nmpc_destroy(h);
        return fail(HILO_ENOTSUP, "hilo_nmpc_create: the learned term of 'chemostat4_gp' has %d training points (limit %d)", (int)nterms,
                    GP2_MAXN);
      }
    }
    c.ext = h->ext_pack;
  } else if (d->learned) {
    hilo_nmpc_destroy(h);
    return fail(HILO_EINVAL, "hilo_nmpc_create: desc.learned given but model %d has no learned term", d->model_id);
  }
  h->base_free_mask = c.x0_free_mask;
  if (jit) {
    JitRequest rq;
    rq.user_source = d->user_source;
    rq.policy = JIT_TRACK;
    rq.N = d->N;
    rq.big = jit_big;
    rq.sym = (d->n_sub <= 1) && !getenv("HILO_NMPC_TAYLOR");
    rq.private_module = d->n_user_gp > 0;
    rc = jit_nmpc_kernels(rq, device, &h->jit);
    if (!rc && getenv("HILO_JIT_COMPILE_ONLY")) { hilo_nmpc_destroy(h); return HILO_COMPILED_ONLY; }   // cache warmed, no handle
    if (!rc && (h->jit.dims[0] != nx || h->jit.dims[1] != nu || h->jit.dims[2] != np))
      rc = fail(HILO_EINVAL, "hilo_nmpc_create: the compiled UserModel has (nx, nu, np) = (%d, %d, %d), the description says (%d, %d, %d)",
                h->jit.dims[0], h->jit.dims[1], h->jit.dims[2], nx, nu, np);
    if (!rc) rc = nmpc_bind_user_gps(h, d);
    if (rc) { hilo_nmpc_destroy(h); return rc; }
    h->jit_policy = JIT_TRACK;
    h->lds_bytes = 0;   // static LDS inside the compiled kernel
    h->jit_ws_bytes = jit_big ? ocp_iter_doubles(nx, nu, 0, d->N) * sizeof(double) : 0;
  }
  if (e == hipSuccess) e = hipMalloc((void**)&h->dev, sizeof(OcpConst));
  if (e == hipSuccess) e = hipMemcpy(h->dev, &c, sizeof(OcpConst), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&h->v_guess, sizeof(double) * h->n_v);
  if (e == hipSuccess) {
    // mpc.py:1468-1482: the guess is tiled over the horizon (scaled, mpc.py:255,259)
    double* g = new double[h->n_v];
    for (int i = 0; i < h->n_v; ++i) g[i] = 0.0;   // shared slack starts at 0 (mpc.py:1535)
    for (int k = 0; k <= d->N; ++k) {
      for (int i = 0; i < nx; ++i) g[k * nxv + i] = (d->x_guess ? d->x_guess[i] : 0.0) / sx[i];
      if (nth) g[k * nxv + nx] = d->theta_guess;                                              // mpc.py:1194
    }
    for (int k = 0; k < d->N; ++k) {
      for (int i = 0; i < nu; ++i) g[(d->N + 1) * nxv + k * nue + i] = (d->u_guess ? d->u_guess[i] : 0.0) / su[i];
      if (nth) g[(d->N + 1) * nxv + k * nue + nu] = d->u_pf_lb + 0.0001;                      // mpc.py:1195
    }
    for (int q = 0; q < d->N * dn; ++q) g[h->n_vc + q] = (d->x_guess ? d->x_guess[q % nx] : 0.0) / sx[q % nx];   // mpc.py:1321
    e = hipMemcpy(h->v_guess, g, sizeof(double) * h->n_v, hipMemcpyHostToDevice);
    delete[] g;
  }
  if (e != hipSuccess) {
    hilo_nmpc_destroy(h);
    return fail(HILO_EHIP, "hilo_nmpc_create: %s", hipGetErrorString(e));
  }
  *out = h;
  return HILO_OK;
}

extern "C" int hilo_nmpc_dims(const hilo_nmpc* h, int* n_v, int* n_g, int* nx, int* nu, int* np) {
  HILO_REQUIRE(h, "hilo_nmpc_dims: NULL handle");
  if (n_v) *n_v = h->n_v;
  if (n_g) *n_g = h->n_g;
  if (nx) *nx = h->nx;
  if (nu) *nu = h->nu;
  if (np) *np = h->np;
  return HILO_OK;
}

// optimize(..., fix_x0=False) (mpc.py:797-807): the measured state is not imposed; x_0 is a variable inside the state box.
// The engine already carries free x_0 components (path variable, shared slack): the model states join that mask.
extern "C" int hilo_nmpc_set_fix_x0(hilo_nmpc* h, int fix_x0) {
  HILO_REQUIRE(h, "hilo_nmpc_set_fix_x0: NULL handle");
  const unsigned want = fix_x0 ? h->base_free_mask : (h->base_free_mask | ((1u << h->nx) - 1u));
  if (want == h->host.x0_free_mask) return HILO_OK;
  HILO_HIP_CHECK(hipSetDevice(h->device));
  HILO_HIP_CHECK(hipDeviceSynchronize());   // launches in flight still read the constants
  h->host.x0_free_mask = want;
  HILO_HIP_CHECK(hipMemcpy(&h->dev->x0_free_mask, &want, sizeof(unsigned), hipMemcpyHostToDevice));
  return HILO_OK;
}

// optimize(fix_x0=False, x0_lb=..., x0_ub=...) (mpc.py:803-807): own box of x_0 in original units; NULL pointers restore the
// state box.  Host pointers [nx]; relaxed like every other bound.
extern "C" int hilo_nmpc_set_x0_box(hilo_nmpc* h, const double* x0_lb_host, const double* x0_ub_host) {
  HILO_REQUIRE(h, "hilo_nmpc_set_x0_box: NULL handle");
  OcpConst& c = h->host;
  const bool on = x0_lb_host || x0_ub_host;
  if (!on && !(c.flags & 2)) return HILO_OK;
  const double relax = 1e-8;
  for (int i = 0; i < h->nx; ++i) {
    double lb = x0_lb_host ? x0_lb_host[i] / c.sz[i] : -INFINITY, ub = x0_ub_host ? x0_ub_host[i] / c.sz[i] : INFINITY;
    if (lb > -INFINITY) lb -= relax * fmax(1.0, fabs(lb));
    if (ub < INFINITY) ub += relax * fmax(1.0, fabs(ub));
    HILO_REQUIRE(lb < ub, "hilo_nmpc_set_x0_box: empty box for state %d", i);
    c.x0lb[i] = lb; c.x0ub[i] = ub;
  }
  c.flags = on ? (c.flags | 2) : (c.flags & ~2);
  HILO_HIP_CHECK(hipSetDevice(h->device));
  HILO_HIP_CHECK(hipDeviceSynchronize());   // launches in flight still read the constants
  HILO_HIP_CHECK(hipMemcpy(h->dev, &c, __builtin_offsetof(OcpConst, cost), hipMemcpyHostToDevice));
  return HILO_OK;
}

static bool nmpc_is_direct(const hilo_nmpc* h) {
  return !h->tv && !h->coll && !h->gen && !h->big && h->jit_coll_d == 0 && h->tv_width == 0 &&
         (h->jit_policy < 0 || h->jit_ws_bytes == 0);
}

// Result gather of a sharded batch (hilo_mpc_amd/dist.py): the solve writes row b = [u0 (nu) | status | iterations] (fp64) of
// `table` ([batch][stride], stride >= nu + 2, device) itself, so that the collective can start from it without a packing
// kernel.  NULL switches it off.  Plain tracking problems only (the others ignore it).
extern "C" int hilo_nmpc_set_aux_outputs(hilo_nmpc* h, double* g, double* lam_x) {
  HILO_REQUIRE(h, "hilo_nmpc_set_aux_outputs: NULL handle");
  h->aux_g = g;
  h->aux_lam_x = lam_x;
  return HILO_OK;
}

extern "C" int hilo_nmpc_set_var_bounds(hilo_nmpc* h, const double* lbx, const double* ubx) {
  HILO_REQUIRE(h, "hilo_nmpc_set_var_bounds: NULL handle");
  HILO_REQUIRE((lbx == nullptr) == (ubx == nullptr), "hilo_nmpc_set_var_bounds: pass both bound arrays or neither");
  if (lbx && (h->tv || h->coll))
    return fail(HILO_ENOTSUP, "hilo_nmpc_set_var_bounds: the precompiled per-stage-data / collocation variants take their bounds "
                              "from the description; the run-time compiled route (desc.user_source) takes them per call");
  h->var_lb = lbx;
  h->var_ub = ubx;
  return HILO_OK;
}

extern "C" int hilo_nmpc_set_gather(hilo_nmpc* h, double* table, int stride) {
  HILO_REQUIRE(h, "hilo_nmpc_set_gather: NULL handle");
  HILO_REQUIRE(!table || stride >= h->nu + 2, "hilo_nmpc_set_gather: stride %d < nu + 2", stride);
  if (table && !nmpc_is_direct(h))
    return fail(HILO_ENOTSUP, "hilo_nmpc_set_gather: this problem kind does not write the gather row itself; pack it on the host");
  h->gather = table;
  h->gather_stride = stride;
  return HILO_OK;
}

extern "C" int hilo_nmpc_set_plant_out(hilo_nmpc* h, double* x_next) {
  HILO_REQUIRE(h, "hilo_nmpc_set_plant_out: NULL handle");
  if (x_next && !(nmpc_is_direct(h) && h->model_id != HILO_MODEL_CHEMOSTAT4_GP && (h->jit_policy < 0 || h->jit_policy == JIT_TRACK)))
    return fail(HILO_ENOTSUP, "hilo_nmpc_set_plant_out: this problem kind does not advance the plant in its solve; call "
                              "hilo_nmpc_plant_step");
  h->plant_out = x_next;
  return HILO_OK;
}

extern "C" int hilo_nmpc_reset_warm_start(hilo_nmpc* h) {
  HILO_REQUIRE(h, "hilo_nmpc_reset_warm_start: NULL handle");
  h->warm_valid = 0;
  return HILO_OK;
}

// par_buf[b] = [p_b (np) | u_old_b (nu)]: assembled on the device so that the solve sees one contiguous row
__global__ void nmpc_pack_par_kernel(int64_t batch, int np, int nu, const double* __restrict__ p, int64_t p_stride,
                                     const double* __restrict__ u_old, double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = np + nu;
  if (e >= batch * w) return;
  const int64_t b = e / w;
  const int i = (int)(e - b * w);
  out[e] = i < np ? (p ? p[b * p_stride + i] : 0.0) : (u_old ? u_old[b * nu + (i - np)] : 0.0);
}

template <class PB>
static int nmpc_launch_pb(hilo_nmpc* h, int64_t batch, const double* x0, const double* par, const double* v0, int64_t v0s,
                          double* v_opt, double* f_opt, double* lam_g, double* u0, int32_t* status, int32_t* iters,
                          double* kkt, hipStream_t s, int64_t par_stride, OcpExtra ex) {
  if (h->lds_bytes > 64 * 1024)
    HILO_HIP_CHECK(hipFuncSetAttribute((const void*)ocp_solve_kernel<PB, OCP_TPB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)h->lds_bytes));
  hipLaunchKernelGGL((ocp_solve_kernel<PB, OCP_TPB>), dim3((unsigned)batch), dim3(OCP_TPB), h->lds_bytes, s, h->dev, batch, x0, par,
                     par_stride, (const double*)nullptr, (int64_t)0, v0, v0s, 0, 0, v_opt, f_opt, lam_g, u0, 0,
                     status, iters, kkt, h->prof, (double*)nullptr, ex);
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

template <class M>
static int nmpc_launch(hilo_nmpc* h, int64_t batch, const double* x0, const double* par, const double* v0, int64_t v0s,
                       double* v_opt, double* f_opt, double* lam_g, double* u0, int32_t* status, int32_t* iters,
                       double* kkt, hipStream_t s, int64_t par_stride, OcpExtra ex) {
  using PB = NmpcTrack<M>;
  if constexpr (PB::SYM) {
    // symbolic model derivatives cover one Runge-Kutta step per interval; sub-stepped integration keeps the Taylor sweeps
    if (h->host.nsub != 1 || getenv("HILO_NMPC_TAYLOR"))
      return nmpc_launch_pb<NmpcTrack<M, false, false>>(h, batch, x0, par, v0, v0s, v_opt, f_opt, lam_g, u0, status, iters, kkt, s,
                                                         par_stride, ex);
  }
  return nmpc_launch_pb<PB>(h, batch, x0, par, v0, v0s, v_opt, f_opt, lam_g, u0, status, iters, kkt, s, par_stride, ex);
}

static int nmpc_solve_impl(hilo_nmpc* h, int64_t batch, const double* x0, const double* p, int64_t p_stride,
                           const double* stage_data, int64_t sd_stride, const double* v0, const double* u_old, double* v_opt,
                           double* f_opt, double* lam_g, double* u0, int32_t* status, int32_t* iters, double* kkt, void* stream) {
  HILO_REQUIRE(h, "hilo_nmpc_solve: NULL handle");
  HILO_REQUIRE(batch >= 0, "hilo_nmpc_solve: negative batch");
  if (batch == 0) return HILO_OK;
  HILO_REQUIRE(x0 && v_opt && f_opt && u0 && status && iters, "hilo_nmpc_solve: NULL argument");
  HILO_REQUIRE(h->np == 0 || p || h->tv || h->tv_width, "hilo_nmpc_solve: the model has %d parameters but p is NULL (mpc.py:771-780)", h->np);
  HILO_REQUIRE((h->tv != nullptr || h->tv_width > 0) == (stage_data != nullptr),
               "hilo_nmpc_solve: handles created with desc.time_varying are solved by hilo_nmpc_solve_tv (and only those)");
  HILO_REQUIRE(p_stride == 0 || p_stride >= h->np, "hilo_nmpc_solve: p_stride %lld < np", (long long)p_stride);
  HILO_HIP_CHECK(hipSetDevice(h->device));
  hipStream_t s = (hipStream_t)stream;
  const int w = h->np + h->nu;
  // Plain tracking problems (precompiled or run-time compiled, no collocation output pass) take the parameter row from the
  // caller's arrays, write the warm-start copy and the gather row themselves: a step is ONE launch.  The other variants keep
  // the packed [p | u_old] row and the device copy of the solution.
  const bool direct = nmpc_is_direct(h);
  if (direct) {
    if (h->warm_batch != batch) {
      if (h->v_warm) HILO_HIP_CHECK(hipFree(h->v_warm));
      h->v_warm = nullptr;
      hipError_t e = hipMalloc((void**)&h->v_warm, sizeof(double) * h->n_v * batch);
      if (e != hipSuccess) return fail(HILO_ENOMEM, "warm-start buffer: %s", hipGetErrorString(e));
      h->warm_batch = batch;
      h->warm_valid = 0;
    }
  } else {
    if (h->par_batch != batch) {
      if (h->par_buf) HILO_HIP_CHECK(hipFree(h->par_buf));
      h->par_buf = nullptr;
      hipError_t e = hipMalloc((void**)&h->par_buf, sizeof(double) * (size_t)w * batch);
      if (e != hipSuccess) return fail(HILO_ENOMEM, "parameter buffer: %s", hipGetErrorString(e));
      h->par_batch = batch;
    }
    const int64_t tot = batch * w;
    hipLaunchKernelGGL(nmpc_pack_par_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, batch, h->np, h->nu, p,
                       p_stride, u_old, h->par_buf);
    HILO_HIP_CHECK(hipGetLastError());
  }
  OcpExtra ex = OcpExtra();
  if (direct) {
    ex.par2 = u_old;
    ex.npar1 = h->np;
    ex.v_copy = h->v_warm;
    ex.gather = h->gather;
    ex.gather_stride = h->gather_stride;
    ex.x_next = h->plant_out;
  }
  if (h->jit_coll_d == 0 && !h->coll) {   // layouts the engine writes itself (no collocation output pass)
    ex.lam_x = h->aux_lam_x;
    ex.g = lam_g ? h->aux_g : nullptr;
  }
  ex.lbx = h->var_lb;
  ex.ubx = h->var_ub;
  ex.bx_stride = h->n_v;
  const double* par_arg = direct ? (p ? p : x0) : h->par_buf;   // np == 0: never dereferenced
  const int64_t par_stride_arg = direct ? p_stride : (int64_t)w;
  // initial guess: explicit v0, else the previous solution (warm start, mpc.py:725-726), else the tiled guess
  const double* vstart = v0;
  int64_t vstride = h->n_v;
  if (!vstart) {
    if (h->warm_valid && h->warm_batch == batch) vstart = h->v_warm;
    else { vstart = h->v_guess; vstride = 0; }
  }
  int rc = HILO_ENOTSUP;
  if (h->jit_policy >= 0) {
    const size_t wsb = h->jit_ws_bytes;
    if (wsb && h->ws_batch != batch) {
      if (h->ws) HILO_HIP_CHECK(hipFree(h->ws));
      h->ws = nullptr;
      hipError_t e = hipMalloc((void**)&h->ws, wsb * (size_t)batch);
      if (e != hipSuccess) return fail(HILO_ENOMEM, "iterate workspace (%zu B per instance): %s", wsb, hipGetErrorString(e));
      h->ws_batch = batch;
    }
    double *vout = v_opt, *lout = lam_g;
    if (h->jit_coll_d > 0) {   // the engine writes its compact solution; the output pass adds the collocation states / rows
      if (h->vc_batch != batch) {
        if (h->vc) HILO_HIP_CHECK(hipFree(h->vc));
        if (h->lamc) HILO_HIP_CHECK(hipFree(h->lamc));
        h->vc = h->lamc = nullptr;
        hipError_t e = hipMalloc((void**)&h->vc, sizeof(double) * (size_t)h->n_vc * batch);
        if (e == hipSuccess) e = hipMalloc((void**)&h->lamc, sizeof(double) * (size_t)(h->n_gc ? h->n_gc : h->N * h->nxv) * batch);
        if (e != hipSuccess) return fail(HILO_ENOMEM, "collocation output buffers: %s", hipGetErrorString(e));
        h->vc_batch = batch;
      }
      vout = h->vc; lout = h->lamc;
    }
    rc = jit_launch_solve(h->jit.solve, h->dev, batch, x0, par_arg, par_stride_arg, stage_data, sd_stride, vstart,
                          vstride, vout, f_opt, lout, u0, status, iters, kkt, h->prof, wsb ? h->ws : nullptr, s, ex);
    if (!rc && h->jit_coll_d > 0)
      rc = jit_launch_coll_out(h->jit.coll_out, h->dev, batch, h->N, h->vc, h->lamc, h->par_buf, (int64_t)(h->np + h->nu), stage_data,
                               sd_stride, v_opt, lam_g, s);
  } else if (h->tv) {
    GenLaunchArgs a{h->dev, batch, x0, h->par_buf, (int64_t)(h->np + h->nu), vstart, vstride, v_opt, f_opt, lam_g, u0,
                    status, iters, kkt, h->prof, h->lds_bytes, s, nullptr};
    rc = h->tv->launch(a, stage_data, sd_stride);
  } else if (h->coll) {
    if (h->vc_batch != batch) {
      if (h->vc) HILO_HIP_CHECK(hipFree(h->vc));
      if (h->lamc) HILO_HIP_CHECK(hipFree(h->lamc));
      h->vc = h->lamc = nullptr;
      hipError_t e = hipMalloc((void**)&h->vc, sizeof(double) * (size_t)h->n_vc * batch);
      if (e == hipSuccess) e = hipMalloc((void**)&h->lamc, sizeof(double) * (size_t)h->N * h->nx * batch);
      if (e != hipSuccess) return fail(HILO_ENOMEM, "collocation output buffers: %s", hipGetErrorString(e));
      h->vc_batch = batch;
    }
    // the engine reads the [x | u] head of each (full-layout) start row and writes its compact solution
    GenLaunchArgs a{h->dev, batch, x0, h->par_buf, (int64_t)(h->np + h->nu), vstart, vstride, h->vc, f_opt, h->lamc, u0,
                    status, iters, kkt, h->prof, h->lds_bytes, s, nullptr};
    rc = h->coll->launch(a);
    if (!rc) rc = h->coll->output(h->dev, batch, h->N, h->vc, lam_g ? h->lamc : nullptr, h->par_buf, (int64_t)(h->np + h->nu),
                                  v_opt, lam_g, s);
  } else if (h->gen || h->big) {
    const size_t wsb = h->gen ? h->gen->ws_bytes(h->N) : h->big->ws_bytes(h->N);
    if (wsb && h->ws_batch != batch) {
      if (h->ws) HILO_HIP_CHECK(hipFree(h->ws));
      h->ws = nullptr;
      hipError_t e = hipMalloc((void**)&h->ws, wsb * (size_t)batch);
      if (e != hipSuccess) return fail(HILO_ENOMEM, "iterate workspace (%zu B per instance): %s", wsb, hipGetErrorString(e));
      h->ws_batch = batch;
    }
    GenLaunchArgs a{h->dev, batch, x0, h->par_buf, (int64_t)(h->np + h->nu), vstart, vstride, v_opt, f_opt, lam_g, u0,
                    status, iters, kkt, h->prof, h->lds_bytes, s, h->ws};
    a.ex.lam_x = ex.lam_x;
    a.ex.g = ex.g;
    a.ex.lbx = ex.lbx; a.ex.ubx = ex.ubx; a.ex.bx_stride = ex.bx_stride;
    rc = h->gen ? h->gen->launch(a) : h->big->launch(a);
  } else {
    switch (h->model_id) {
#define X(ID, T) case ID: rc = nmpc_launch<T>(h, batch, x0, par_arg, vstart, vstride, v_opt, f_opt, lam_g, u0, status, iters, kkt, s, par_stride_arg, ex); break;
      HILO_NMPC_MODELS(X)
#undef X
    }
  }
  if (rc) return rc;
  if (direct) { h->warm_valid = 1; return HILO_OK; }   // the solve wrote the warm-start copy itself
  // keep the solution for the next call (un-shifted, like the reference)
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

extern "C" int hilo_nmpc_solve(hilo_nmpc* h, int64_t batch, const double* x0, const double* p, int64_t p_stride,
                               const double* v0, const double* u_old, double* v_opt, double* f_opt, double* lam_g,
                               double* u0, int32_t* status, int32_t* iters, double* kkt, void* stream) {
  return nmpc_solve_impl(h, batch, x0, p, p_stride, nullptr, 0, v0, u_old, v_opt, f_opt, lam_g, u0, status, iters, kkt, stream);
}

extern "C" int hilo_nmpc_solve_tv(hilo_nmpc* h, int64_t batch, const double* x0, const double* stage_data, int64_t sd_stride,
                                  const double* v0, const double* u_old, double* v_opt, double* f_opt, double* lam_g,
                                  double* u0, int32_t* status, int32_t* iters, double* kkt, void* stream) {
  HILO_REQUIRE(h && (h->tv || h->tv_width), "hilo_nmpc_solve_tv: the handle was not created with desc.time_varying");
  HILO_REQUIRE(batch <= 0 || stage_data, "hilo_nmpc_solve_tv: NULL stage data");
  const int64_t need = (int64_t)(h->N + 1) * (h->nx + h->nu + h->np);
  HILO_REQUIRE(sd_stride == 0 || sd_stride >= need, "hilo_nmpc_solve_tv: sd_stride %lld < %lld", (long long)sd_stride,
               (long long)need);
  return nmpc_solve_impl(h, batch, x0, nullptr, 0, stage_data, sd_stride, v0, u_old, v_opt, f_opt, lam_g, u0, status, iters, kkt,
                         stream);
}

// Developer aid: per-phase shader-clock totals of instance 0 of the next solves
// (derivatives, errors+barrier update, Riccati, step lengths, line search, update, #factorisations, #trial points);
// cycles_host[8]
extern "C" int hilo_nmpc_profile(hilo_nmpc* h, int enable, long long* cycles_host) {
  HILO_REQUIRE(h, "hilo_nmpc_profile: NULL handle");
  HILO_HIP_CHECK(hipSetDevice(h->device));
  if (cycles_host && h->prof) {
    HILO_HIP_CHECK(hipDeviceSynchronize());
    HILO_HIP_CHECK(hipMemcpy(cycles_host, h->prof, sizeof(long long) * PH_COUNT, hipMemcpyDeviceToHost));
  }
  if (enable && !h->prof) {
    HILO_HIP_CHECK(hipMalloc((void**)&h->prof, sizeof(long long) * PH_COUNT));
    HILO_HIP_CHECK(hipMemset(h->prof, 0, sizeof(long long) * PH_COUNT));
  } else if (!enable && h->prof) {
    HILO_HIP_CHECK(hipFree(h->prof));
    h->prof = nullptr;
  }
  return HILO_OK;
}

extern "C" int hilo_nmpc_plant_step(hilo_nmpc* h, int64_t batch, const double* x, const double* u, const double* p,
                                    int64_t p_stride, double* x_next, void* stream) {
  HILO_REQUIRE(h, "hilo_nmpc_plant_step: NULL handle");
  if (batch <= 0) return HILO_OK;
  HILO_REQUIRE(x && u && x_next && (h->np == 0 || p), "hilo_nmpc_plant_step: NULL argument");
  HILO_HIP_CHECK(hipSetDevice(h->device));
  if (h->jit_policy >= 0) return jit_launch_plant(h->jit.plant, h->dev, batch, x, u, p, p_stride, x_next, (hipStream_t)stream);
  const unsigned grid = (unsigned)((batch + 255) / 256);
  switch (h->model_id) {
#define X(ID, T) case ID: hipLaunchKernelGGL((plant_step_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, h->dev, batch, x, u, p, p_stride, x_next); break;
    HILO_NMPC_MODELS(X)
#undef X
  }
  HILO_HIP_CHECK(hipGetLastError());
  return HILO_OK;
}

#ifdef HILO_OCP_DPROF
extern "C" int hilo_debug_dprof(long long* out, int reset) {
  if (out) HILO_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(hilo::g_dprof), sizeof(long long) * 32));
  if (reset) { long long z[32] = {0}; HILO_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(hilo::g_dprof), z, sizeof(z))); }
  return HILO_OK;
}
#endif
