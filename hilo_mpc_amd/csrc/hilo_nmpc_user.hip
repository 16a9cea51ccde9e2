// NMPC.setup() for problems on the general run-time compiled policy (csrc/hilo_nmpc_user.h): fills the problem constants
// in the layout of that policy, asks hilo_jit.hip for the kernels, builds the handle.
//
// What the reference does at this point: `NMPC._setup` (hilo_mpc/modules/controller/mpc.py:1133-1787) builds the NLP graph and
// `ca.nlpsol` compiles its derivatives.  The layout rules restated here are those of hilo_nmpc.hip (same file:line
// references) extended by: control horizon Nc < N (v = [x (N+1) | u (Nc) | e | ip], mpc.py:1476-1485, :1629-1630), the
// path variable inside the collocation scheme, generic costs, the continuous objective.
#include <string.h>

#include "hilo_nmpc_handle.h"
#include "hilo_nmpc_user.h"

extern "C" int hilo_model_dims(int model_id, int* nx, int* nu, int* np, int* ny, int* discrete);
extern "C" void hilo_nmpc_destroy(hilo_nmpc* h);

namespace hilo {

static void copy_or1(double* dst, const double* src, int n, double dflt) {
  for (int i = 0; i < n; ++i) dst[i] = src ? src[i] : dflt;
}

int nmpc_user_create(const hilo_nmpc_desc* d, int device, hilo_nmpc** out) {
  HILO_REQUIRE(d->N >= 1 && d->N <= 512, "hilo_nmpc_create: horizon %d out of range [1, 512]", d->N);
  HILO_REQUIRE(d->dt > 0.0, "hilo_nmpc_create: dt must be positive");
  const int N = d->N, Nc = d->Nc > 0 ? d->Nc : d->N;
  HILO_REQUIRE(Nc <= N, "hilo_nmpc_create: control horizon %d exceeds the prediction horizon %d", Nc, N);
  int mx, mu, np, ny = 0, discrete = 0;
  if (d->model_id == HILO_MODEL_USER) {
    mx = d->user_nx; mu = d->user_nu; np = d->user_np; discrete = d->user_discrete;
    HILO_REQUIRE(mx >= 1 && mu >= 0 && np >= 0, "hilo_nmpc_create: bad user model dimensions");
  } else {
    int rc = hilo_model_dims(d->model_id, &mx, &mu, &np, &ny, &discrete);
    if (rc) return rc;
  }
  if (d->learned) return fail(HILO_ENOTSUP, "run-time compiled problems take their learned terms through desc.user_gp");
  const int nth = d->n_path_var;
  HILO_REQUIRE(nth >= 0 && nth <= 1, "hilo_nmpc_create: at most one path variable is supported (got %d)", nth);
  const int D = d->collocation_degree;
  HILO_REQUIRE(D >= 0 && D <= COLL_MAXD, "hilo_nmpc_create: collocation degree %d out of range [0, %d]", D, COLL_MAXD);
  HILO_REQUIRE(!D || (d->coll_A && d->coll_D), "hilo_nmpc_create: collocation needs the basis (coll_A, coll_D)");
  HILO_REQUIRE(!D || !discrete, "hilo_nmpc_create: collocation needs the continuous model");
  const bool cont = d->objective_continuous != 0 && !discrete;
  HILO_REQUIRE(!(cont && D) || d->coll_B, "hilo_nmpc_create: the continuous objective with collocation needs coll_B");
  if (D && d->n_tcon > 0 && d->tcon_soft)
    return fail(HILO_ENOTSUP, "collocation together with a SOFT terminal constraint is not built (hard ones are)");
  // ---- inequality rows (same construction as hilo_nmpc.hip) ----
  int ne = 0, nrow = 0, n_con_ref = 0, ntrow = 0, n_tcon_ref = 0, ne_stage = 0, ne_cus0 = 0;
  int row_expr[OCP_MAXNC], row_sign[OCP_MAXNC], row_e[OCP_MAXNC], row_ref[OCP_MAXNC];
  int trow_expr[OCP_MAXNC], trow_sign[OCP_MAXNC], trow_e[OCP_MAXNC], trow_ref[OCP_MAXNC];
  double row_lb[OCP_MAXNC], row_ub[OCP_MAXNC], trow_lb[OCP_MAXNC], trow_ub[OCP_MAXNC];
  if (d->n_con > 0) {
    ne = d->con_soft ? d->n_con : 0;
    n_con_ref = d->con_soft ? 2 * d->n_con : d->n_con;   // rows per stage in the reference's g (mpc.py:1711-1712)
    for (int j = 0; j < d->n_con; ++j) {
      const double lb = d->con_lb ? d->con_lb[j] : -INFINITY, ub = d->con_ub ? d->con_ub[j] : INFINITY;
      HILO_REQUIRE(lb <= ub, "hilo_nmpc_create: constraint %d has lb > ub", j);
      if (d->con_soft) {   // c - e <= ub | -c - e <= -lb; a row without a finite bound constrains nothing and is dropped
        if (ub < INFINITY) {
          HILO_REQUIRE(nrow < OCP_MAXNC, "too many constraint rows");
          row_expr[nrow] = j; row_sign[nrow] = 1; row_e[nrow] = j; row_lb[nrow] = -INFINITY; row_ub[nrow] = ub; row_ref[nrow++] = j;
        }
        if (lb > -INFINITY) {
          HILO_REQUIRE(nrow < OCP_MAXNC, "too many constraint rows");
          row_expr[nrow] = j; row_sign[nrow] = -1; row_e[nrow] = j; row_lb[nrow] = -INFINITY; row_ub[nrow] = -lb;
          row_ref[nrow++] = d->n_con + j;
        }
      } else if (lb > -INFINITY || ub < INFINITY) {
        HILO_REQUIRE(nrow < OCP_MAXNC, "too many constraint rows");
        row_expr[nrow] = j; row_sign[nrow] = 1; row_e[nrow] = -1; row_lb[nrow] = lb; row_ub[nrow] = ub; row_ref[nrow++] = j;
      }
    }
  }
  // finite bounds on algebraic states: hidden hard rows z_a in [lb, ub], bounded at the collocation points only (below)
  const int nrow_user = nrow;
  const int nzb = d->n_zbound;
  HILO_REQUIRE(nzb >= 0 && (nzb == 0 || (D > 0 && d->zb_lb && d->zb_ub)), "hilo_nmpc_create: bounds on algebraic states need collocation");
  for (int q = 0; q < nzb; ++q) {
    HILO_REQUIRE(nrow < OCP_MAXNC, "too many constraint rows");
    HILO_REQUIRE(d->zb_lb[q] <= d->zb_ub[q], "hilo_nmpc_create: z bound %d has lb > ub", q);
    row_expr[nrow] = d->n_con + q; row_sign[nrow] = 1; row_e[nrow] = -1; row_lb[nrow] = d->zb_lb[q]; row_ub[nrow] = d->zb_ub[q];
    row_ref[nrow++] = -1;
  }
  ne_stage = ne;
  if (d->n_tcon > 0) {
    if (d->tcon_soft) ne += d->n_tcon;   // slacks e_T behind the stage slacks in v (mpc.py:1540-1548)
    n_tcon_ref = d->tcon_soft ? 2 * d->n_tcon : d->n_tcon;
    for (int j = 0; j < d->n_tcon; ++j) {
      const double lb = d->tcon_lb ? d->tcon_lb[j] : -INFINITY, ub = d->tcon_ub ? d->tcon_ub[j] : INFINITY;
      HILO_REQUIRE(lb <= ub, "hilo_nmpc_create: terminal constraint %d has lb > ub", j);
      auto add = [&](int sign, int e, double rlb, double rub, int ref) {
        trow_expr[ntrow] = j; trow_sign[ntrow] = sign; trow_e[ntrow] = e; trow_lb[ntrow] = rlb; trow_ub[ntrow] = rub;
        trow_ref[ntrow++] = ref;
      };
      HILO_REQUIRE(nrow + ntrow + (d->tcon_soft ? (ub < INFINITY) + (lb > -INFINITY) : 1) <= OCP_MAXNC, "too many constraint rows");
      if (d->tcon_soft) {
        if (ub < INFINITY) add(1, ne_stage + j, -INFINITY, ub, j);
        if (lb > -INFINITY) add(-1, ne_stage + j, -INFINITY, -lb, d->n_tcon + j);
      } else {
        add(1, -1, lb, ub, j);
      }
    }
  }
  // rows of a custom constraint function (desc.n_acc; hilo_mpc_amd/custom.py): one accumulator state each, imposed as the LAST
  // terminal rows (hilo_nmpc_user.h::term_rows); in the engine's lam_g they follow the terminal rows - the host moves them to the end
  // of g, where the reference has them (mpc.py:1729-1745)
  const int nq = d->n_acc, npsi = d->n_acc_expr;
  ne_cus0 = ne;
  HILO_REQUIRE(nq >= 0 && nq <= 2 && npsi >= 0 && npsi <= 4, "hilo_nmpc_create: at most 2 custom rows over at most 4 stage expressions");
  if (nq > 0) {
    HILO_REQUIRE(npsi > 0 && d->acc_coef && d->acc_lb && d->acc_ub, "hilo_nmpc_create: custom rows need acc_coef, acc_lb, acc_ub");
    if (d->n_tcon > 0 && d->tcon_soft) return fail(HILO_ENOTSUP, "custom rows together with a SOFT terminal constraint are not built");
    if (d->Nc > 0 && d->Nc < d->N) return fail(HILO_ENOTSUP, "custom rows together with a control horizon Nc < N are not built");
    // terminal row table: expression index USER_QROW + r marks a row of custom function r (sign and slack like every soft row)
    if (d->acc_soft) ne += nq;           // slacks e_cus behind the other slacks in v (mpc.py:1551-1556)
    for (int pass = 0; pass < (d->acc_soft ? 2 : 1); ++pass)
      for (int r = 0; r < nq; ++r) {
        HILO_REQUIRE(d->acc_lb[r] <= d->acc_ub[r], "hilo_nmpc_create: custom row %d has lb > ub", r);
        const double lb = d->acc_lb[r], ub = d->acc_ub[r];
        if (d->acc_soft && !(pass == 0 ? ub < INFINITY : lb > -INFINITY)) continue;     // not imposed: zero multiplier in its place
        HILO_REQUIRE(nrow + ntrow < OCP_MAXNC, "too many constraint rows");
        trow_expr[ntrow] = USER_QROW + r;
        if (!d->acc_soft) { trow_sign[ntrow] = 1; trow_e[ntrow] = -1; trow_lb[ntrow] = lb; trow_ub[ntrow] = ub; }
        else if (pass == 0) { trow_sign[ntrow] = 1; trow_e[ntrow] = ne_cus0 + r; trow_lb[ntrow] = -INFINITY; trow_ub[ntrow] = ub; }      // fun - e <= ub
        else { trow_sign[ntrow] = -1; trow_e[ntrow] = ne_cus0 + r; trow_lb[ntrow] = -INFINITY; trow_ub[ntrow] = -lb; }                 // -(fun + e) <= -lb
        trow_ref[ntrow++] = n_tcon_ref + pass * nq + r;
      }
    n_tcon_ref += d->acc_soft ? 2 * nq : nq;
  }
  // collocation: the reference imposes the stage constraints at every collocation point as well as at the node (mpc.py:1338-1356,
  // :1700-1725) - the engine's rows of a stage are the node's rows followed by those of the d collocation points
  const int nrow_pt = nrow;
  if (D && nrow > 0) {
    HILO_REQUIRE(nrow * (D + 1) + ntrow <= OCP_MAXNC, "hilo_nmpc_create: %d constraint rows at %d points per interval exceed %d", nrow,
                 D + 1, OCP_MAXNC);
    for (int i = 1; i <= D; ++i)
      for (int r = 0; r < nrow_pt; ++r) { row_lb[i * nrow_pt + r] = row_lb[r]; row_ub[i * nrow_pt + r] = row_ub[r]; }
    for (int r = nrow_user; r < nrow_pt; ++r) { row_lb[r] = -INFINITY; row_ub[r] = INFINITY; }   // z bounds: not at the node (no z variable there)
    nrow = nrow_pt * (D + 1);
  }
  const int nc = nrow + ntrow;
  const bool hold = Nc < N;
  const int mxa = mx + nth, mua = mu + nth;
  const int nh = hold ? mua : 0;
  const int nxe = mxa + ne + nq + nh, nue = mua, nz = nxe + nue;
  HILO_REQUIRE(nxe <= OCP_MAXNX && nue <= OCP_MAXNU,
               "hilo_nmpc_create: %d engine states / %d inputs (model + path variable + slacks + held inputs) exceed %d / %d", nxe,
               nue, OCP_MAXNX, OCP_MAXNU);
  const int nps = d->n_path_stage, npt = d->n_path_term;
  HILO_REQUIRE(nps >= 0 && npt >= 0 && nps <= 8 && npt <= 8, "hilo_nmpc_create: at most 8 path terms per cost");
  HILO_REQUIRE((nps + npt == 0) || nth, "hilo_nmpc_create: path terms need a path variable");
  const UserLayout L(mx, mu, nth, ne, nps, npt, nq, npsi, N);
  HILO_REQUIRE(L.o_end <= OCP_NCOST, "hilo_nmpc_create: cost block too small for this problem");
  const bool tv = d->time_varying != 0;
  const int nsd = tv ? mx + mu + np : 0;
  const int nconst = (int)((__builtin_offsetof(OcpConst, cost) + sizeof(double) * L.o_end + 7) / 8);
  size_t fixed_b = ocp_fixed_doubles(nxe, nue, nconst, np + mu, nsd, 0, N) * sizeof(double);
  const size_t iter_b = ocp_iter_doubles(nxe, nue, nc, N) * sizeof(double);
  // (HILO_FORCE_BIG: developer / test knob - a problem that fits LDS runs in workspace mode, the path of the long horizons)
  const bool big = fixed_b + iter_b > 160 * 1024 || getenv("HILO_FORCE_BIG") != nullptr;
  // workspace mode under collocation (NmpcUser::PREP / XCW, same formulas): per interval the collocation states with their tangents
  // and adjoint weights when the lanes of a wave solve the system together (coll_pass), else the states and the factors of their
  // Newton matrix (hilo_colloc.h::prepare); one interval's block is staged in LDS
  const int dnc = D * (mx + nth), nwd = mx + mu + 2 * nth;
  const bool coop = big && D && (nwd + 1 <= dnc ? dnc : dnc + nwd + 1) <= 64 && mua <= mxa;
  const size_t prep_w = !(big && D) ? 0 : (coop ? (size_t)dnc * (nwd + 2) + nwd : (size_t)dnc * (1 + dnc));
  const size_t xc_w = coop ? (size_t)(dnc + nwd) : 0;
  {   // the LDS staging area (Ocp::PREPL): PREPB blocks of the direction passes, or the cooperative pass's states and factors
    const size_t cgs = nwd + 1 <= dnc ? dnc : dnc + nwd + 1, cg = coop ? 64 / cgs : 1;
    const size_t blocks = prep_w * (coop ? 3 : 1), cstage = coop ? cg * dnc * (1 + (size_t)dnc) : 0;
    fixed_b += (blocks > cstage ? blocks : cstage) * sizeof(double);
  }
  if (fixed_b > 160 * 1024) return fail(HILO_ENOTSUP, "horizon %d needs %zu B of LDS for the problem constants alone", N, fixed_b);
  const size_t prep_b = (size_t)N * (prep_w + xc_w) * sizeof(double);

  hilo_nmpc* h = new hilo_nmpc();
  memset(h, 0, sizeof(*h));
  h->device = device; h->model_id = d->model_id; h->nx = mx; h->nu = mu; h->np = np; h->N = N;
  h->nu_out = mu; h->jit_policy = JIT_USER;
  h->nxe = nxe; h->nue = nue; h->nxv = mxa; h->ntail = ne + nq; h->Nc = Nc;   // (the accumulators ride in the tail of v: hidden by the host)
  h->tv_width = nsd;
  h->jit_ws_bytes = big ? iter_b + prep_b : 0;
  h->jit_coll_d = D;
  const int nza = d->model_id == HILO_MODEL_USER ? d->user_nz : 0;
  HILO_REQUIRE(nza >= 0 && nza <= 4, "hilo_nmpc_create: at most 4 algebraic states (got %d)", nza);
  // algebraic states under an EXPLICIT Runge-Kutta transcription (mpc.py:1375-1412, modeling.py:1213-1281): one block of algebraic
  // variables per stage and interval, eliminated inside the shooting map (the emitted model solves them, codegen.py::dae_model_source
  // with alg_at_slope) and rebuilt with the multipliers of their rows by the output pass (hilo_nmpc_user.h::erk_dae_output) - for the
  // plain problem shape of the reference's own case (tests/test_NMPC.py:1950-1975): quadratic costs and boxes
  const int erk_s = d->erk_order >= 1 ? d->erk_order : 4;
  const bool erk_dae = nza > 0 && D == 0;
  if (erk_dae) {
    if (discrete) { delete h; return fail(HILO_ENOTSUP, "algebraic states of a discrete model are not built"); }
    h->jit_coll_d = 100 + erk_s;      // (non-zero: the output pass runs)
    if (nrow + ntrow > 0 || nth > 0 || nq > 0 || Nc < N || (d->n_sub > 1) || ne > 0) {
      delete h;
      return fail(HILO_ENOTSUP, "algebraic states under an explicit Runge-Kutta transcription are built for quadratic costs and box "
                                "constraints (no nonlinear / custom constraints, path variable, control horizon, sub-steps): use 'collocation'");
    }
  }
  h->n_vc = (N + 1) * mxa + Nc * mua + ne + nq;
  h->n_v = h->n_vc + (nza ? (N + 1) * nza : 0) + N * D * (mxa + nza) + (erk_dae ? N * erk_s * nza : 0);   // mpc.py:1440-1453, :1488-1548 (+ nq hidden tail entries)
  // mpc.py:1338-1372 (per collocation point: constraint rows, then the collocation equations), :1657-1669, :1684-1725
  h->n_g = N * (mxa + n_con_ref + D * (mxa + nza) + (D ? D * n_con_ref : 0) + (erk_dae ? erk_s * nza : 0)) + n_tcon_ref;
  h->n_gc = D ? N * (mxa + nrow) + ntrow : 0;   // the engine's compact multiplier row handed to the output pass
  OcpConst& c = h->host;
  memset(&c, 0, sizeof(c));
  ocp_default_options(c);
  c.N = N; c.Nc = Nc; c.order = d->erk_order >= 1 ? d->erk_order : 4; c.nsub = d->n_sub >= 1 ? d->n_sub : 1;
  c.dt = d->dt;
  c.flags = 1;
  if (D) {
    c.coll.d = D;
    for (int i = 0; i < D * D; ++i) c.coll.A[i] = d->coll_A[i];
    for (int i = 0; i <= D; ++i) { c.coll.Dc[i] = d->coll_D[i]; c.coll.Bq[i] = d->coll_B ? d->coll_B[i] : 0.0; }
  }
  if (d->max_iter > 0) c.max_iter = d->max_iter;
  if (d->acceptable_iter > 0) c.acceptable_iter = d->acceptable_iter;
  if (d->tol > 0) c.tol = d->tol;
  if (d->acceptable_tol > 0) c.acceptable_tol = d->acceptable_tol;
  if (d->mu_init > 0) c.mu_init = d->mu_init;
  if (d->max_hessian_perturbation > 0) c.delta_w_max = d->max_hessian_perturbation;
  double sx[OCP_MAXNX], su[OCP_MAXNU];
  copy_or1(sx, d->x_scaling, mx, 1.0);
  copy_or1(su, d->u_scaling, mu, 1.0);
  for (int i = 0; i < nz; ++i) c.sz[i] = 1.0;   // path variable, slacks, held inputs, virtual input: unit scaling (mpc.py:1200-1201)
  for (int i = 0; i < mx; ++i) c.sz[i] = sx[i];
  for (int i = 0; i < mu; ++i) c.sz[nxe + i] = su[i];
  const double relax = d->bound_relax_factor >= 0.0 ? d->bound_relax_factor : 1e-8;
  c.bound_relax = relax;
  auto relaxed_lb = [&](double lb) { return lb > -INFINITY ? lb - relax * fmax(1.0, fabs(lb)) : lb; };
  auto relaxed_ub = [&](double ub) { return ub < INFINITY ? ub + relax * fmax(1.0, fabs(ub)) : ub; };
  // ---- cost block (UserLayout): model z index -> augmented z index [x, theta | u, u_theta]
  auto az = [&](int i) { return i < mx ? i : mxa + (i - mx); };
  const int mz = mx + mu, mza = mxa + mua;
  for (int i = 0; i < mz; ++i) {
    for (int j = 0; j < mz; ++j) c.cost[L.o_wz + az(i) * mza + az(j)] = d->Wz ? d->Wz[i * mz + j] : 0.0;
    c.cost[L.o_zref + az(i)] = d->zref ? d->zref[i] : 0.0;
  }
  for (int i = 0; i < mx; ++i) {
    for (int j = 0; j < mx; ++j) c.cost[L.o_wn + i * mxa + j] = d->WN ? d->WN[i * mx + j] : 0.0;
    c.cost[L.o_xrefn + i] = d->xrefN ? d->xrefN[i] : 0.0;
  }
  for (int i = 0; i < mu * mu; ++i) c.cost[L.o_wdu + i] = d->Wdu ? d->Wdu[i] : 0.0;
  c.cost[L.o_hasdu] = d->Wdu ? 1.0 : 0.0;
  if (nth && d->has_u_pf_ref) {   // mpc.py:1202-1204
    const int iu = mxa + mu;
    c.cost[L.o_wz + iu * mza + iu] = d->u_pf_weight;
    c.cost[L.o_zref + iu] = d->u_pf_ref;
  }
  for (int a = 0; a < ne_stage; ++a)    // e^T W e once per stage (mpc.py:1708), W = 1e4 I by default (modeling.py:875)
    for (int b = 0; b < ne_stage; ++b)
      c.cost[L.o_we + a * ne + b] = d->con_weight ? d->con_weight[a * ne_stage + b] : (a == b ? 1e4 : 0.0);
  for (int a = ne_stage; a < ne_cus0; ++a)   // e_T^T W e_T once (mpc.py:1686)
    for (int b = ne_stage; b < ne_cus0; ++b)
      c.cost[L.o_wet + a * ne + b] = d->tcon_weight ? d->tcon_weight[(a - ne_stage) * d->n_tcon + (b - ne_stage)] : (a == b ? 1e4 : 0.0);
  for (int a = ne_cus0; a < ne; ++a) c.cost[L.o_wet + a * ne + a] = 1e4;   // 1e4 e_cus^T e_cus once (mpc.py:1732-1733)
  int rcode = HILO_OK;
  for (int a = 0; a < nps; ++a) {
    if (!d->path_stage_idx || !d->path_stage_W || d->path_stage_idx[a] < 0 || d->path_stage_idx[a] >= mx)
      rcode = fail(HILO_EINVAL, "hilo_nmpc_create: bad stage path term %d", a);
    else {
      c.cost[L.o_idxs + a] = d->path_stage_idx[a];
      for (int b = 0; b < nps; ++b) c.cost[L.o_ws + a * nps + b] = d->path_stage_W[a * nps + b];
    }
  }
  for (int a = 0; a < npt; ++a) {
    if (!d->path_term_idx || !d->path_term_W || d->path_term_idx[a] < 0 || d->path_term_idx[a] >= mx)
      rcode = fail(HILO_EINVAL, "hilo_nmpc_create: bad terminal path term %d", a);
    else {
      c.cost[L.o_idxt + a] = d->path_term_idx[a];
      for (int b = 0; b < npt; ++b) c.cost[L.o_wt + a * npt + b] = d->path_term_W[a * npt + b];
    }
  }
  for (int r = 0; r < nq; ++r)
    for (int k = 0; k <= N; ++k)
      for (int j = 0; j < npsi; ++j) c.cost[L.o_acc + (r * (N + 1) + k) * npsi + j] = d->acc_coef[((size_t)r * (N + 1) + k) * npsi + j];
  if (rcode) { delete h; return rcode; }
  c.nc = nrow; c.nc_term = ntrow; c.n_con_ref = n_con_ref; c.n_tcon_ref = n_tcon_ref;
  c.cost[L.o_tsoft] = d->tcon_soft ? 1.0 : 0.0;
  c.cost[L.o_nrow] = nrow_pt; c.cost[L.o_ncr] = n_con_ref; c.cost[L.o_ntr] = n_tcon_ref;
  for (int m = 0; m < OCP_MAXNC; ++m) { c.dlb[m] = -INFINITY; c.dub[m] = INFINITY; }
  for (int m = 0; m < nrow_pt; ++m) {
    c.cost[L.o_rowx + m] = row_expr[m]; c.cost[L.o_rows + m] = row_sign[m]; c.cost[L.o_rowe + m] = row_e[m];
    c.cost[L.o_rref + m] = row_ref[m];
  }
  for (int m = 0; m < nrow; ++m) {
    c.dlb[m] = relaxed_lb(row_lb[m]); c.dub[m] = relaxed_ub(row_ub[m]);
    c.row_ref[m] = (short)(D ? m : row_ref[m]);
  }
  for (int r = 0; r < ntrow; ++r) {
    c.cost[L.o_trowx + r] = trow_expr[r]; c.cost[L.o_trows + r] = trow_sign[r]; c.cost[L.o_trowe + r] = trow_e[r];
    c.cost[L.o_trref + r] = trow_ref[r];
    c.dlb[nrow + r] = relaxed_lb(trow_lb[r]); c.dub[nrow + r] = relaxed_ub(trow_ub[r]);
    c.trow_ref[r] = (short)(D ? r : trow_ref[r]);
  }
  if (D) {
    // the solve kernel writes its multipliers compactly (identity row maps), the output pass puts them into the reference's order;
    // the slacks sit BEHIND the collocation blocks in the rows the engine reads (v0, lbx / ubx; mpc.py:1529 follows :1497-1527)
    c.n_con_ref = nrow; c.n_tcon_ref = ntrow;
    if (ne + nq > 0) c.tail_off = h->n_v - (ne + nq);
  }
  // ---- structural sparsity of the interval Hessian (desc.hess_pattern over the augmented model z): which pair directions the
  // Taylor sweeps visit.  Engine-only variables: the shared slacks couple only through off-diagonal penalty weights (rows are
  // linear in them); held inputs stand in for the inputs beyond the control horizon - kept dense.
  if (d->hess_pattern && !hold) {
    for (auto& w : c.pair_mask) w = 0ull;
    auto set_pair = [&](int i, int j) {   // engine indices
      if (i == j) return;
      const int a = i < j ? i : j, b = i < j ? j : i;
      const int p = a * (nz - 1) - a * (a - 1) / 2 + (b - a - 1);   // dir_of(a, b, nz) - nz
      if (p < 256) c.pair_mask[p >> 6] |= 1ull << (p & 63);
    };
    auto eng = [&](int a) { return a < mxa ? a : nxe + (a - mxa); };
    for (int a = 0; a < mza; ++a)
      for (int b = a + 1; b < mza; ++b)
        if (d->hess_pattern[a * mza + b] || d->hess_pattern[b * mza + a]) set_pair(eng(a), eng(b));
    for (int a = 0; a < ne; ++a)
      for (int b = a + 1; b < ne; ++b)
        if (c.cost[L.o_we + a * ne + b] != 0.0 || c.cost[L.o_we + b * ne + a] != 0.0 || c.cost[L.o_wet + a * ne + b] != 0.0 ||
            c.cost[L.o_wet + b * ne + a] != 0.0)
          set_pair(mxa + a, mxa + b);
  }
  for (int i = mx; i < mxa + ne; ++i) c.x0_free_mask |= 1u << i;           // theta_0 and the slacks are variables (mpc.py:785-789)
  c.x0_free_mask |= (unsigned)d->x0_free_mask & ((1u << mx) - 1u);         // model states declared free (desc.x0_free_mask)
  for (int a = 0; a < ne; ++a) c.k0_only_mask |= 1u << (mxa + a);          // one box per shared slack
  h->base_free_mask = c.x0_free_mask;
  // ---- boxes of the engine's z = [x | theta | e | uh | u | u_theta], scaled like mpc.py:253-259, relaxed like IPOPT ----
  for (int i = 0; i < nz; ++i) {
    double lb = -INFINITY, ub = INFINITY;
    if (i < mx) { if (d->x_lb) lb = d->x_lb[i] / c.sz[i]; if (d->x_ub) ub = d->x_ub[i] / c.sz[i]; }
    else if (i < mxa) { lb = d->theta_lb; ub = d->theta_ub; }                                      // mpc.py:1198-1199
    else if (i < mxa + ne) {                                                                       // :1533-1534, :1544-1545
      const int a = i - mxa;
      lb = 0.0;
      ub = a < ne_stage ? (d->con_max_violation ? d->con_max_violation[a] : INFINITY)
           : a < ne_cus0 ? (d->tcon_max_violation ? d->tcon_max_violation[a - ne_stage] : INFINITY)
                         : (d->acc_max_violation ? d->acc_max_violation[a - ne_cus0] : INFINITY);
    }
    else if (i < nxe) {}                                                                            // held inputs: states without a box
    else if (i < nxe + mu) { const int j = i - nxe; if (d->u_lb) lb = d->u_lb[j] / c.sz[i]; if (d->u_ub) ub = d->u_ub[j] / c.sz[i]; }
    else { lb = d->u_pf_lb; ub = d->u_pf_ub; }                                                      // mpc.py:1196-1197
    lb = relaxed_lb(lb); ub = relaxed_ub(ub);
    if (!(lb < ub)) { delete h; return fail(HILO_EINVAL, "hilo_nmpc_create: empty box for variable %d", i); }
    c.lbz[i] = lb; c.ubz[i] = ub;
  }
  // ---- compile / load ----
  JitRequest rq;
  rq.user_source = d->user_source;
  rq.policy = JIT_USER;
  rq.nth = nth; rq.ne = ne; rq.nc = nc; rq.coll_d = D; rq.N = N;
  rq.hold = hold; rq.cont = cont; rq.tv = tv; rq.big = big; rq.has_fun = d->user_has_fun != 0;
  rq.nq = nq;
  rq.private_module = d->n_user_gp > 0;
  for (int i = 0; i < mza; ++i) {   // row masks of the non-zero stage weights (hilo_nmpc_user.h::lagrange): part of the compiled problem
    unsigned m = 0u;
    for (int j = 0; j < mza; ++j)
      if (c.cost[L.o_wz + i * mza + j] != 0.0) m |= 1u << j;
    c.cost[L.o_wzm + i] = (double)m;
    if (i < 24) rq.wz_mask[i] = m;
  }
  rq.has_wz_mask = mza <= 24;
  int rc = jit_nmpc_kernels(rq, device, &h->jit);
  if (!rc && getenv("HILO_JIT_COMPILE_ONLY")) { hilo_nmpc_destroy(h); return HILO_COMPILED_ONLY; }   // cache warmed, no handle
  if (!rc && (h->jit.dims[0] != mx || h->jit.dims[1] != mu || h->jit.dims[2] != np || h->jit.dims[6] != nxe || h->jit.dims[7] != nue))
    rc = fail(HILO_EINVAL, "hilo_nmpc_create: the compiled problem has model (nx, nu, np) = (%d, %d, %d), engine (%d, %d); the "
                           "description says (%d, %d, %d), (%d, %d)", h->jit.dims[0], h->jit.dims[1], h->jit.dims[2], h->jit.dims[6],
              h->jit.dims[7], mx, mu, np, nxe, nue);
  if (!rc) rc = nmpc_bind_user_gps(h, d);
  if (rc) { hilo_nmpc_destroy(h); return rc; }
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipMalloc((void**)&h->dev, sizeof(OcpConst));
  if (e == hipSuccess) e = hipMemcpy(h->dev, &c, sizeof(OcpConst), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&h->v_guess, sizeof(double) * h->n_v);
  if (e == hipSuccess) {
    // mpc.py:1468-1482: the guess is tiled over the horizon (scaled, mpc.py:255,259); slacks start at 0 (mpc.py:1535) - those of soft
    // custom rows at the NUMBER of rows (mpc.py:1555 writes the size where the others write zeros)
    double* g = new double[h->n_v];
    for (int i = 0; i < h->n_v; ++i) g[i] = 0.0;
    for (int a = ne_cus0; a < ne; ++a) g[h->n_v - (ne + nq) + a] = (double)nq;
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < mx; ++i) g[k * mxa + i] = (d->x_guess ? d->x_guess[i] : 0.0) / sx[i];
      if (nth) g[k * mxa + mx] = d->theta_guess;                                            // mpc.py:1194
    }
    for (int k = 0; k < Nc; ++k) {
      for (int i = 0; i < mu; ++i) g[(N + 1) * mxa + k * mua + i] = (d->u_guess ? d->u_guess[i] : 0.0) / su[i];
      if (nth) g[(N + 1) * mxa + k * mua + mu] = d->u_pf_lb + 0.0001;                       // mpc.py:1195
    }
    // mpc.py:1321: the collocation states start at the state guess (only the [x | u] head and the slacks are read back by the engine)
    const int head = (N + 1) * mxa + Nc * mua, zn = nza ? (N + 1) * nza : 0;
    for (int k = 0; k < N && D; ++k)
      for (int i = 0; i < D * mxa; ++i) {
        const int a = i % mxa;
        g[head + zn + k * D * (mxa + nza) + i] = a < mx ? (d->x_guess ? d->x_guess[a] : 0.0) / sx[a] : d->theta_guess;
      }
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

int nmpc_bind_user_gps(hilo_nmpc* h, const hilo_nmpc_desc* d) {
  HILO_REQUIRE(d->n_user_gp >= 0 && d->n_user_gp <= 4, "hilo_nmpc_create: at most 4 learned terms (got %d)", d->n_user_gp);
  if (d->n_user_gp == 0) return HILO_OK;
  HILO_REQUIRE(h->jit.gp_table, "hilo_nmpc_create: the compiled module exports no learned-term table");
  const double* table[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < d->n_user_gp; ++k) {
    HILO_REQUIRE(d->user_gp[k], "hilo_nmpc_create: user_gp[%d] is NULL", k);
    int rc = gp_pack_se(d->user_gp[k], &h->user_gp_pack[k]);
    if (rc) return rc;
    table[k] = h->user_gp_pack[k];
  }
  HILO_HIP_CHECK(hipMemcpy((void*)h->jit.gp_table, table, sizeof(table), hipMemcpyHostToDevice));
  return HILO_OK;
}

}  // namespace hilo

using namespace hilo;

extern "C" int hilo_jit_precompile(const char* user_source, int policy, int nth, int ne, int nc, int coll_d, int N, int hold, int cont,
                                   int tv, int big, int has_fun) {
  HILO_REQUIRE(user_source, "hilo_jit_precompile: NULL source");
  JitRequest rq;
  rq.user_source = user_source;
  rq.policy = policy; rq.nth = nth; rq.ne = ne; rq.nc = nc; rq.coll_d = coll_d; rq.N = N;
  rq.hold = hold != 0; rq.cont = cont != 0; rq.tv = tv != 0; rq.big = big != 0; rq.has_fun = has_fun != 0;
  JitKernels k;
  setenv("HILO_JIT_COMPILE_ONLY", "1", 1);
  const int rc = jit_nmpc_kernels(rq, 0, &k);
  unsetenv("HILO_JIT_COMPILE_ONLY");
  return rc;
}
