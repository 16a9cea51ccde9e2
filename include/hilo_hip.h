/*
 * hilo_hip.h - C ABI of libhilo_hip.so: MI355X-native batched MPC / MHE / Kalman / GP solve path.
 *
 * This is the drop-in boundary.  HILO-MPC (the reference, pure Python on CasADi) has no C ABI;
 * its boundary is a Python callable stored in `self._solver` / `self._function` at `setup()` time.  Each entry
 * point below names the reference interface it replaces (file:line relative to the reference root).
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *  - every data pointer is a DEVICE pointer (HBM resident) unless its name ends in `_host`;
 *    hilo_malloc / hilo_memcpy_* are exported so that a binding needs nothing but this library
 *  - all floating point data is IEEE fp64 (the reference computes in fp64 throughout); indices are int32/int64
 *  - arrays are row-major with the batch index leading: [batch][...]
 *  - a `*_stride` argument is the distance in doubles between consecutive instances; 0 = shared by the batch
 *  - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous w.r.t. the host unless noted
 *  - every function returns 0 on success or a negative HILO_E* code; hilo_last_error() gives the message
 *  - handles are not thread-safe; use one handle per host thread / stream
 */
#ifndef HILO_HIP_H
#define HILO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HILO_ABI_VERSION 1

/* error codes */
#define HILO_OK 0
#define HILO_EINVAL (-1)   /* bad argument (dimension mismatch, unknown id, NULL pointer) */
#define HILO_ENOMEM (-2)   /* device allocation failed */
#define HILO_EHIP (-3)     /* HIP runtime error */
#define HILO_ENOTSUP (-4)  /* combination not built into this library */
#define HILO_COMPILED_ONLY 1 /* not an error: with HILO_JIT_COMPILE_ONLY set in the environment, hilo_nmpc_create / hilo_kf_create of a
                               run-time compiled problem stop after compiling it into the cache (machines without a GPU: image builds) */
#define HILO_ENOTPD (-5)   /* covariance matrix K + sn2 I not positive definite (the reference adds no jitter, inference.py:206) */

/* model zoo ids (device functors in hilo_mpc_amd/csrc/hilo_models.h) */
#define HILO_MODEL_LTI 0         /* x+ = A x + B u, y = C x; p = [A|B|C] row-major (mpc.py:2198-2245 LMPC; KF) */
#define HILO_MODEL_TOY1D 1       /* tests/test_KFs.py:548-556 */
#define HILO_MODEL_BIOREACTOR3 2 /* tests/test_KFs.py:691-712 */
#define HILO_MODEL_CHEMOSTAT4 3  /* hilo_mpc/library/models.py:163-198 + rate laws :143-148 */
#define HILO_MODEL_PENDULUM4 4   /* tests/test_NMPC.py:12-43 */
#define HILO_MODEL_ROBOT6 5
#define HILO_MODEL_CSTR3 6
#define HILO_MODEL_LINEAR2 7     /* tests/test_KFs.py:247-255 */
#define HILO_MODEL_CHEMOSTAT4_GP 8 /* CHEMOSTAT4 with the biomass growth rate `mu` replaced by a GP mean over (S, I):
                                      Model.substitute_from(gp), dynamic_model.py:3040-3125 (NMPC only) */

/* solver status codes: hilo_mpc/modules/optimizer.py:1085-1104 */
#define HILO_STATUS_SOLVED 1
#define HILO_STATUS_ACCEPTABLE 2
#define HILO_STATUS_INFEASIBLE 3
#define HILO_STATUS_RESTORATION_FAILED 4
#define HILO_STATUS_MAXITER 5
#define HILO_STATUS_OTHER (-1)

/* ------------------------------------------------------------------------------------------------------- */
/* library                                                                                                  */
/* ------------------------------------------------------------------------------------------------------- */
int hilo_abi_version(void);
const char* hilo_last_error(void);           /* thread-local, valid until the next failing call */
int hilo_device_count(int* count);
int hilo_model_dims(int model_id, int* nx, int* nu, int* np, int* ny, int* discrete);

int hilo_malloc(void** dptr, uint64_t bytes, int device);
int hilo_free(void* dptr);
int hilo_memcpy_h2d(void* dst, const void* src_host, uint64_t bytes, void* stream);
int hilo_memcpy_d2h(void* dst_host, const void* src, uint64_t bytes, void* stream);
int hilo_stream_sync(void* stream);

/* ------------------------------------------------------------------------------------------------------- */
/* Kalman filters: KF / EKF / UKF                                                                           */
/* replaces the `ca.Function`s built by `_KalmanFilter.setup` (hilo_mpc/modules/estimator/kf.py:207-277):   */
/*   prediction_step(x0=[x|P], p=[u;p], Q)        kf.py:129-133 (UKF :550-554)                              */
/*   update_step(x0=[x|P](|X), y, p=[u;p], R)     kf.py:182-186 (UKF :600-604)                              */
/*   function(x0, y, p, Q, R) = update(predict())  kf.py:258-265, called from `estimate` kf.py:296-306       */
/* ------------------------------------------------------------------------------------------------------- */
#define HILO_KF_KF 0
#define HILO_KF_EKF 1
#define HILO_KF_UKF 2

typedef struct hilo_kf hilo_kf;

typedef struct hilo_kf_desc {
  int32_t model_id;   /* HILO_MODEL_* */
  int32_t kind;       /* HILO_KF_*; KF and EKF share the arithmetic (kf.py:89-96) */
  int32_t continuous; /* 0: model is (or was discretised to) a map x+ = Phi(x): P- = F P F^T + Q (kf.py:95-96)
                         1: continuous model: [x; vec P] integrated over dt (kf.py:97-110); the reference uses
                            CVODES, this library fixed-step RK4 with n_sub sub-steps */
  int32_t erk_order;  /* `Model.discretize('erk', order)` order 1..4 for continuous==0 on a continuous model */
  int32_t n_sub;      /* sub-steps per sampling interval (>=1) */
  int32_t lti_nx, lti_nu, lti_ny; /* only for HILO_MODEL_LTI */
  double dt;          /* sampling interval */
  double alpha, beta, kappa; /* UKF tuning (kf.py:446-454 defaults 1e-3, 2, 0) */
  /* model_id = HILO_MODEL_USER (100): the model is the C++ source of `struct UserModel` (the shape of csrc/hilo_models.h; what
     hilo_mpc_amd/codegen.py emits for `Model.set_dynamical_equations` / `set_measurement_equations`), compiled with hiprtc at
     create and cached like the controllers' (hilo_nmpc_desc.user_source).  NULL for the zoo models. */
  const char* user_source;
  /* learned terms of the user model (`Model.substitute_from(gp)` before `EKF(model)` / `UKF(model)` / `ParticleFilter(model)`): the
     source refers to them as hilo_user_gp[k] exactly like a controller's (hilo_nmpc_desc.user_gp); the filter gets a module
     instance of its own, so two filters compiled from the same source keep their own tables.  The GPs must outlive nothing:
     their posterior is packed at create. */
  int32_t n_user_gp;
  const struct hilo_gp* user_gp[4];   /* trained handles of hilo_gp_create (section 4) */
} hilo_kf_desc;

int hilo_kf_create(const hilo_kf_desc* desc, int device, hilo_kf** out);
/* compile the filter kernels of a model source into the cache without loading them (no GPU needed; CPU test-suite, container builds) */
int hilo_jit_precompile_kf(const char* user_source);
void hilo_kf_destroy(hilo_kf* kf);

/* Particle filter (hilo_mpc/modules/estimator/pf.py).  The reference assembles ONE function at setup() (pf.py:300-318) and
   calls it once per estimate (:372); this is that function for a batch of filters of `n_samples` particles each, on the model
   / sampling interval / discretisation of a filter handle (the `kind` of the handle is irrelevant):
       X_prop = Phi(X, u, p) + w      Y = h(X_prop, u, p) + v      q_j = normpdf(Y_j; y, sqrt(R)) / sum_j ...   (pf.py:99, :140-158)
   All particle arrays are particle-major, [B][n_samples][nx] / [B][n_samples][ny] (the memory order of CasADi's column-major
   nx x n_samples matrices); y [B][ny]; R [B][ny][ny] (r_stride 0: shared), only its diagonal is used (pf.py:155-156); a model
   without measurement equations measures all states (pf.py:131-134: ny := nx).  The random draws w, v stay with the caller
   (pf.py:364-368 draws them with numpy). */
int hilo_pf_function(hilo_kf* kf, int64_t batch, int n_samples, const double* X, const double* y,
                     const double* up, int64_t up_stride, const double* w, const double* v,
                     const double* R, int64_t r_stride, double* X_prop, double* Y, double* q, void* stream);
/* `ind = np.random.choice(N, size=N, replace=True, p=q); X = X[:, ind]; Y = Y[:, ind]` (pf.py:404-407) with the caller's
   uniform draws in [0, 1) - numpy's own algorithm: index = searchsorted(cumsum(q) / sum(q), u, side='right').
   uniforms [B][n_samples]; index (int32) [B][n_samples]; n_samples <= 8192. */
int hilo_pf_resample(hilo_kf* kf, int64_t batch, int n_samples, const double* X_prop, const double* Y, const double* q,
                     const double* uniforms, double* X, double* Y_out, int32_t* index, void* stream);
/* Statistics of the particle set (pf.py:409-420): x_mean [B][nx], y_mean [B][ny], P = np.cov(X) [B][nx][nx] (unbiased),
   x_min / x_max [B][nx] (the spread the roughening scales its noise with).  `add` (NULL or [B][n_samples][nx]) is added to
   the particles first: the roughening step `X += dx` (pf.py:415). */
int hilo_pf_stats(hilo_kf* kf, int64_t batch, int n_samples, double* X, const double* Y, const double* add,
                  double* x_mean, double* y_mean, double* P, double* x_min, double* x_max, void* stream);
/* widths of the packed tiles: xp = nx+1 ([x|P]); pred = nx+1 (KF/EKF) or 1+nx+(2nx+1) (UKF [x|P|X]) */
int hilo_kf_dims(const hilo_kf* kf, int* nx, int* nu, int* np, int* ny, int* pred_width);

int hilo_kf_predict(hilo_kf* kf, int64_t batch,
                    const double* xP,                 /* [B][nx][nx+1] */
                    const double* up, int64_t up_stride, /* [B][nu+np] = vertcat(u, p) (kf.py:130) */
                    const double* Q, int64_t q_stride,   /* [B][nx][nx] */
                    double* pred,                     /* [B][nx][pred_width] */
                    void* stream);
int hilo_kf_update(hilo_kf* kf, int64_t batch,
                   const double* pred,                /* [B][nx][pred_width] */
                   const double* y,                   /* [B][ny] */
                   const double* up, int64_t up_stride,
                   const double* R, int64_t r_stride, /* [B][ny][ny] */
                   double* xP_out,                    /* [B][nx][nx+1] */
                   double* y_pred,                    /* [B][ny] */
                   void* stream);
/* one `estimate()` step: update(predict(.)), fused in one kernel */
int hilo_kf_step(hilo_kf* kf, int64_t batch, const double* xP, const double* y,
                 const double* up, int64_t up_stride, const double* Q, int64_t q_stride,
                 const double* R, int64_t r_stride, double* xP_out, double* y_pred, void* stream);

/* `steps` estimate() steps in ONE launch: `self._function.mapaccum(steps)` (kf.py:296-306).  y [steps][B][ny]; inputs / parameters
   the same for every step (up_step_stride = 0) or [steps][B][nu+np] (up_step_stride = elements between two steps); Q, R as
   for a single step.  keep_all != 0: xP_out [steps][B][nx][nx+1] holds the tile after every step (what mapaccum returns), else
   [B][nx][nx+1] the last one; y_pred [steps][B][ny]. */
int hilo_kf_steps(hilo_kf* kf, int64_t batch, int steps, const double* xP, const double* y,
                  const double* up, int64_t up_stride, int64_t up_step_stride, const double* Q, int64_t q_stride,
                  const double* R, int64_t r_stride, double* xP_out, int keep_all, double* y_pred, void* stream);
/* The same with the inputs u [steps or 1][B][nu] and the parameters p [B][np] in their OWN arrays (strides in doubles; 0 = one row   */
/* shared by the batch) - they are separate arguments of the reference's function (kf.py:130), and a binding need not pack them.    */
int hilo_kf_steps_split(hilo_kf* kf, int64_t batch, int steps, const double* xP, const double* y,
                        const double* u, int64_t u_stride, int64_t u_step_stride, const double* p, int64_t p_stride,
                        const double* Q, int64_t q_stride, const double* R, int64_t r_stride, double* xP_out, int keep_all,
                        double* y_pred, void* stream);

/* ------------------------------------------------------------------------------------------------------- */
/* Gaussian process: exact inference + prediction                                                           */
/* replaces `ca.Function('prediction',[X,w,p],[mean,var])` (hilo_mpc/modules/machine_learning/gp/gp.py:      */
/* 623-629, called from `predict` gp.py:709) whose body is `ExactInference.get_posterior`                   */
/* (gp/inference.py:172-221) over `Kernel.__call__` (gp/kernel.py:97-205) and `Mean.__call__` (mean.py:90)  */
/* ------------------------------------------------------------------------------------------------------- */
/* kernel / mean programs: postfix list of nodes, each node = [opcode, n_active, active_dims..., n_par, par...]
   stored as doubles.  opcodes: */
#define HILO_K_CONST 0       /* par: c = bias^2                                      kernel.py:465-485 */
#define HILO_K_GAMMAEXP 1    /* par: sf2, alpha, p/2, M[n_active]  (SE: .5, 1)       kernel.py:650-701 */
#define HILO_K_MATERN 2      /* par: sf2, sqrt(2nu), ncoef, coef..., M[n_active]     kernel.py:783-826 */
#define HILO_K_RQ 3          /* par: sf2, alpha, M[n_active]                         kernel.py:972-1003 */
#define HILO_K_PP 4          /* par: sf2, q, j, M[n_active]                          kernel.py:1069-1109 */
#define HILO_K_POLY 5        /* par: sf2, offset, degree                             kernel.py:1202-1234 */
#define HILO_K_NN 6          /* par: sf2, w                                          kernel.py:1309-1332 */
#define HILO_K_PERIODIC 7    /* par: 2*log(sf), l, period  (1 active dim)            kernel.py:1394-1423 */
#define HILO_K_SUM 16        /* pops 2                                               kernel.py:1562-1593 */
#define HILO_K_PRODUCT 17    /* pops 2                                               kernel.py:1596-1627 */
#define HILO_K_POWER 18      /* par: power; child evaluated at (x,x)                 kernel.py:1630-1666 */
#define HILO_M_CONST 32      /* par: bias                                            mean.py:280-305 */
#define HILO_M_POLY 33       /* par: offset, degree, coef[n_active]                  mean.py:422-470 */
#define HILO_M_SUM 48
#define HILO_M_PRODUCT 49
#define HILO_M_POWER 50      /* par: power */
#define HILO_M_SCALE 51      /* par: scale */

typedef struct hilo_gp hilo_gp;

/* Builds K(X,X) + sn2 I, its Cholesky factor, alpha and the log marginal likelihood on the device
   (inference.py:199-210).  X_train is feature-major [nf][n] exactly like the reference's X (gp.py:605). */
int hilo_gp_create(int device, int nf, int n,
                   const double* X_train_host,  /* [nf][n] */
                   const double* y_train_host,  /* [n] */
                   const double* kprog_host, int kprog_len,
                   const double* mprog_host, int mprog_len,
                   double noise_variance,       /* sn2 = exp(2 * log sqrt(noise_variance)) (inference.py:199) */
                   hilo_gp** out);
void hilo_gp_destroy(hilo_gp* gp);
int hilo_gp_log_marginal_likelihood(hilo_gp* gp, double* lml_host);
/* New hyper-parameters on the same training data (one objective value of `GaussianProcess.fit_model`, gp.py:660-697): uploads
   the kernel program (same length: the kernel structure is fixed) and the noise variance, re-factorises into the handle's
   buffers.  Returns HILO_ENOTPD at an indefinite trial point (the handle then needs another refit before it predicts). */
int hilo_gp_refit(hilo_gp* gp, const double* kprog_host, int kprog_len, double noise_variance);
/* New hyper-parameters of the mean function (`Mean` hyper-parameters are fitted together with the kernel's, gp.py:408-414): same
   program length; the next hilo_gp_refit evaluates the mean with them. */
int hilo_gp_set_mean_program(hilo_gp* gp, const double* mprog_host, int mprog_len);
/* Gradient of the log marginal likelihood (inference.py:210) at the handle's current hyper-parameters by the trace formula
   1/2 tr((alpha alpha^T - K_y^-1) dK_y/dtheta_j) on the device, one factorisation for all j.  Per theta_j the
   caller passes the kernel programs and noise variances at theta +- h_j e_j (HOST: [n_theta][2][kprog_len], [n_theta][2],
   [n_theta]); dK_y/dtheta_j is their central difference, evaluated element-wise inside the trace kernel. */
int hilo_gp_lml_gradient(hilo_gp* gp, int n_theta, const double* kprogs_pm_host, const double* noise_pm_host,
                         const double* h_host, double* grad_host);
/* gp.py:699-718.  Xq feature-major [nf][m]; mean/var [m]; var may be NULL (mean only). */
int hilo_gp_predict(hilo_gp* gp, int64_t m, const double* Xq, int noise_free, double* mean, double* var,
                    void* stream);
/* covariance matrix only (Kernel.__call__, kernel.py:97-140): K [n1][n2] for feature-major X1 [nf][n1], X2 [nf][n2] */
int hilo_gp_kernel_matrix(int device, int nf, const double* kprog_host, int kprog_len, int64_t n1,
                          const double* X1, int64_t n2, const double* X2, double* K, void* stream);
int hilo_gp_mean(int device, int nf, const double* mprog_host, int mprog_len, int64_t n, const double* X,
                 double* mu, void* stream);

/* ------------------------------------------------------------------------------------------------------- */
/* NMPC: batched direct multiple shooting + interior point                                                  */
/* replaces `ca.nlpsol('solver','ipopt',{'f','x','p','g'})` built at hilo_mpc/modules/controller/mpc.py:      */
/* 1778-1787 and called as `solver(x0=v0, lbx, ubx, lbg, ubg, p)` at mpc.py:722 by `NMPC._optimize`; the       */
/* transcription it embodies is mpc.py:1455-1787 for a pre-discretised model + `integration_method='discrete'`  */
/* ------------------------------------------------------------------------------------------------------- */
typedef struct hilo_nmpc hilo_nmpc;

typedef struct hilo_nmpc_desc {
  int32_t model_id;     /* HILO_MODEL_* */
  int32_t N;            /* prediction horizon (mpc.py `horizon`) */
  int32_t Nc;           /* control horizon (mpc.py:1629-1630); 0 = N.  Nc < N needs user_policy 2 (run-time compiled) */
  int32_t erk_order;    /* `model.discretize('rk4')` = 4, `('erk', order)` = 1..4 (modeling.py:1239-1250) */
  int32_t n_sub;        /* sub-steps per interval (1 = the reference's single ERK step) */
  int32_t max_iter;     /* 0 -> 3000 (IPOPT default) */
  int32_t acceptable_iter; /* 0 -> 15 */
  int32_t reserved;
  double dt;            /* sampling interval (mpc.py:511) */
  double tol;           /* 0 -> 1e-8 */
  double acceptable_tol;/* 0 -> 1e-6 */
  double mu_init;       /* 0 -> 0.1 */
  double bound_relax_factor; /* < 0 -> 1e-8 (IPOPT default); 0 disables */
  /* HOST pointers, all optional (NULL = zero weight / no bound / unit scaling / zero guess).
     Quadratic costs in the form QuadraticCost builds (modeling.py:243-283), on scaled z = (x, u):
       stage: (z - zref)^T Wz (z - zref);  terminal: (x - xrefN)^T WN (x - xrefN);
       input change (only interval 0, mpc.py:1631-1635): (u_0 - u_old)^T Wdu (u_0 - u_old) */
  const double* Wz;     /* [nz][nz], nz = nx+nu */
  const double* zref;   /* [nz]  already divided by the scaling (modeling.py:310) */
  const double* WN;     /* [nx][nx] */
  const double* xrefN;  /* [nx] */
  const double* Wdu;    /* [nu][nu] */
  const double* x_lb; const double* x_ub; const double* u_lb; const double* u_ub;  /* original units */
  const double* x_scaling; const double* u_scaling;                                /* optimizer.py:1476-1506 */
  const double* x_guess; const double* u_guess;                                    /* original units */
  /* learned term of the model (`Model.substitute_from(gp)`, dynamic_model.py:3040-3125): required for
     HILO_MODEL_CHEMOSTAT4_GP (label `mu`, features S, I; the posterior mean is copied at create), else NULL */
  const hilo_gp* learned;
  /* ---- path following (mpc.py:1025-1053 `create_path_variable`, :1173-1204): at most one path variable theta.  It
     becomes state index nx with the virtual input u_theta (input index nu), theta+ = theta + dt u_theta (:1191);
     theta_0 is a free bounded variable (:785-789 pins only the original states).  Path cost terms
     (x[idx] - r(theta))^T W (x[idx] - r(theta)) (hilo_mpc/util/modeling.py:252-283), stage and terminal. ---- */
  int32_t n_path_var;                    /* 0 or 1 */
  int32_t has_u_pf_ref;                  /* adds (u_theta - u_pf_ref)^2 u_pf_weight to the stage cost (:1202-1204) */
  double theta_guess, theta_lb, theta_ub, u_pf_lb, u_pf_ub, u_pf_ref, u_pf_weight;
  int32_t n_path_stage, n_path_term;     /* tracked states per cost, <= 4 each */
  const int32_t* path_stage_idx; const double* path_stage_W;   /* [n], [n][n] */
  const int32_t* path_term_idx;  const double* path_term_W;
  const double* path_prog;               /* n_path_stage + n_path_term expression programs r(theta) (HILO_X_*) */
  int32_t path_prog_len;
  /* ---- nonlinear stage constraint lb <= c(x_k,u_k) <= ub, k = 0..N-1 (`GenericConstraint`, modeling.py:820-1005;
     mpc.py:1271-1283, :1700-1725), on un-scaled variables (modeling.py:843-849).  soft: one slack vector e >= 0 shared by
     all stages (mpc.py:1529-1537), rows c - e <= ub, -c - e <= -lb, e^T W e added once per stage (:1708); e is appended
     to the decision vector. ---- */
  int32_t n_con;                         /* expressions, <= 2 */
  int32_t con_soft;
  int32_t con_prog_len;
  const double* con_prog;                /* n_con programs in the model states (VARX), inputs (VARU), parameters */
  const double* con_lb; const double* con_ub;      /* [n_con]; -inf / +inf allowed */
  const double* con_weight;              /* [n_con][n_con] or NULL -> 1e4 I (modeling.py:875) */
  const double* con_max_violation;       /* [n_con] or NULL -> inf */
  /* terminal constraint; hard: lb <= c_T(x_end) <= ub on the integrated end state Phi(x_{N-1}, u_{N-1}) (mpc.py:1693-1700;
     `nmpc.terminal_constraint`), un-scaled states; its rows sit between the last defect and the last stage rows in g */
  int32_t n_tcon;                        /* expressions, n_con + n_tcon rows <= 4 in this build */
  int32_t tcon_prog_len;
  const double* tcon_prog;
  const double* tcon_lb; const double* tcon_ub;
  /* soft terminal constraint (mpc.py:1684-1692): c_T(x_{N-1}) - e_T <= ub, -c_T(x_{N-1}) - e_T <= -lb on the state the last
     interval starts from, one slack e_T in [0, max_violation] after the stage slack in v (mpc.py:1540-1548), e_T^T W e_T once
     in the objective.  One shared slack in total in this build: n_tcon = 1 and the stage constraint hard or absent. */
  int32_t tcon_soft; int32_t reserved4;
  const double* tcon_weight;             /* [n_tcon][n_tcon] or NULL -> 1e4 I */
  const double* tcon_max_violation;      /* [n_tcon] or NULL -> inf */
  /* ---- integration_method = 'collocation' on the CONTINUOUS model (the reference's default, optimizer.py:1410-1418;
     hilo_mpc/util/modeling.py:1091-1211, mpc.py:1307-1372): degree d Lagrange basis at Radau / Legendre points.  The caller
     passes the basis it built (modeling.py:1091-1127): coll_A = (C[1:,1:]^T)^-1 (d x d, the method's Runge-Kutta matrix)
     and coll_D = D[0..d].  v gains the collocation states ([x | u | ip], mpc.py:1497-1518), g the collocation rows
     (per stage [collocation rows | continuity], :1657-1669).  0 = explicit Runge-Kutta / discrete model. ---- */
  int32_t collocation_degree;            /* 0, or 3 (1..4 for chemostat4) in this build */
  int32_t reserved2;
  const double* coll_A;                  /* [d][d] */
  const double* coll_D;                  /* [d+1] */
  /* ---- per-stage data: trajectory-tracking references (mpc.py:365-463; modeling.py:262-283) and time-varying parameters
     (mpc.py:335-364, optimizer.py:905-929).  With time_varying != 0 the handle is solved through hilo_nmpc_solve_tv. ---- */
  int32_t time_varying;
  int32_t reserved3;
  /* ---- models and problem functions compiled at RUN TIME (what `Model.set_dynamical_equations` /
     `nmpc.stage_cost.cost = ...` accept as CasADi graphs in the reference, dynamic_model.py:1293-1553, modeling.py:38-87).
     `user_source` is HIP source text that defines, inside namespace hilo,
        UserModel   a functor shaped like the zoo of csrc/hilo_models.h (NX, NU, NP, NY, DISCRETE, templated `ode`, `meas`),
                    or an alias of a zoo functor (`using UserModel = Chemostat4;`)
        UserFun     (user_has_fun != 0) the problem's free-form functions, csrc/hilo_nmpc_user.h: generic stage / terminal
                    cost on the SCALED variables (see that header for the reference quirk), constraint expressions on the
                    un-scaled ones, path references
     It is compiled with hiprtc for gfx950 against the engine headers that ship next to the library (csrc/), the code object
     is cached (HILO_JIT_CACHE or <library dir>/jit_cache).  model_id = HILO_MODEL_USER takes the model from the source;
     a zoo model_id with user_source set compiles the general policy for that zoo functor.
     user_policy: 0 = the tracking policy (same code path as the zoo models), 1 = the general policy with expression
     programs (path following / constraints as above), 2 = the policy of csrc/hilo_nmpc_user.h, which adds: generic costs,
     the continuous objective, collocation together with path following, control horizon Nc < N, per-stage data. ---- */
  const char* user_source;
  int32_t user_nx, user_nu, user_np, user_ny, user_discrete;   /* dimensions of UserModel (checked against the compiled code) */
  int32_t user_has_fun;
  int32_t user_policy;
  int32_t objective_continuous;   /* 1: integrate the Lagrange term with the shooting map (optimizer.py:1423-1426) */
  const double* coll_B;           /* [d+1] quadrature weights B_i of the collocation basis (modeling.py:1124), continuous objective */
  /* learned terms of a run-time compiled model (`Model.substitute_from(gp)`, dynamic_model.py:3040-3125): up to 4 trained GPs
     (squared-exponential kernel over up to 8 features, constant / zero mean); the emitted model refers to them as
     gp_se_mean(hilo_user_gp[k], features) (csrc/hilo_models.h) */
  int32_t n_user_gp;
  int32_t user_nz;                /* algebraic states of the user model (semi-explicit DAE, `set_algebraic_states`): the source's
                                     UserModel::NZ; v / g gain the z blocks and rows of mpc.py:1488-1518 (collocation only) */
  const hilo_gp* user_gp[4];
  /* with user_policy 2 the constraint / path expressions are compiled into UserFun: n_con, n_tcon, n_path_stage, n_path_term
     count them as above, the *_prog pointers stay NULL */
  /* user_policy 2: structural sparsity of the Hessian of an interval's Lagrangian, [mza][mza] bytes over the augmented model
     z = [x, theta | u, u_theta] (1 = the entry can be non-zero; symmetric; NULL = dense).  CasADi evaluates only the non-zero
     Hessian entries of the graph it is given (mpc.py:1778-1787); here a zero entry drops the Taylor sweep of its direction. */
  const unsigned char* hess_pattern;
  double max_hessian_perturbation;   /* IPOPT option of that name (delta_w^max of W&B Alg. IC, default 1e20 there, 1e40 here =
                                        the oracle's): beyond it the inertia correction gives up -> status 4; <= 0 keeps the default */
  /* finite bounds on algebraic states (`set_box_constraints(z_lb=, z_ub=)`, mpc.py:645-701 -> the box of the zp blocks of v,
     :1512-1518): UserFun::con evaluates n_zbound further expressions behind the n_con constraint expressions - the bounded
     algebraic states - which become rows at the collocation points only (the algebraic states are eliminated);
     they do not appear in lam_g */
  int32_t n_zbound;
  int32_t x0_free_mask;    /* run-time compiled problems: bit i set = component i of the model state x_0 is NOT pinned to the measured
                              state but a variable inside the state box (the sampling-interval state of a minimum-time problem,
                              hilo_mpc_amd/nmpc.py::_setup_min_time); 0 = all of x_0 is pinned (mpc.py:785-789) */
  const double* zb_lb; const double* zb_ub;        /* [n_zbound]; -inf / +inf allowed on one side */
  /* custom constraint function over the whole decision vector (`set_custom_constraints_function`, optimizer.py:1180-1208; rows
     lb <= fun(v, x_ind, u_ind) <= ub at the END of g, mpc.py:1729-1745), in the stage-additive form the host derives from it
     (hilo_mpc_amd/custom.py):  c_r(v) = sum_{k=0..N} sum_j acc_coef[r][k][j] psi_j(x_k, u_k)  with the n_acc_expr expressions psi_j
     compiled into UserFun::acc (user_policy 2).  Each of the n_acc rows is carried by an accumulator state of the engine and becomes
     a hard row on the end of the horizon (csrc/hilo_nmpc_user.h): hilo_nmpc_dims reports n_v INCLUDING the n_acc hidden accumulator
     entries at the end of v, and the rows are the last n_acc TERMINAL rows of g / lam_g (in front of the last node's stage rows);
     the reference appends them to g (mpc.py:1744-1745) - hilo_mpc_amd/nmpc.py::_g_order is that permutation. */
  int32_t n_acc, n_acc_expr;
  const double* acc_coef;      /* [n_acc][N + 1][n_acc_expr] */
  const double* acc_lb; const double* acc_ub;      /* [n_acc] bounds of the rows (constant parts of the function already removed) */
  /* soft = True (mpc.py:1551-1556, :1731-1740): one slack e_cus per row in [0, max_violation] BEHIND the other slacks in v,
     1e4 e_cus^T e_cus once in the objective, and two rows per function in g: fun - e_cus <= ub, then fun + e_cus >= lb (rows with an
     infinite bound are not imposed and keep a zero multiplier) */
  int32_t acc_soft;
  const double* acc_max_violation;                 /* [n_acc] or NULL -> inf */
} hilo_nmpc_desc;

#define HILO_MODEL_USER 100    /* model defined by desc.user_source */
/* Compile a user problem into the cache without loading it (no GPU needed): what `__graft_entry__.build()` uses to pre-warm
   the cache, and what the CPU test-suite uses to check that generated sources compile for gfx950. */
int hilo_jit_precompile(const char* user_source, int policy, int nth, int ne, int nc, int coll_d, int N, int hold, int cont,
                        int tv, int big, int has_fun);

/* expression programs: [len, (op, arg) * len/2] back to back; postfix, stack of 8 */
#define HILO_X_CONST 0   /* arg = value */
#define HILO_X_VARX 1    /* arg = state index */
#define HILO_X_VARU 2    /* arg = input index */
#define HILO_X_VARP 3    /* arg = parameter index */
#define HILO_X_ADD 10
#define HILO_X_SUB 11
#define HILO_X_MUL 12
#define HILO_X_DIV 13
#define HILO_X_NEG 14
#define HILO_X_SQ 15
#define HILO_X_SIN 16
#define HILO_X_COS 17
#define HILO_X_EXP 18
#define HILO_X_LOG 19
#define HILO_X_SQRT 20
#define HILO_X_POWI 21   /* arg = integer exponent in [-16, 16] */

int hilo_nmpc_create(const hilo_nmpc_desc* desc, int device, hilo_nmpc** out);   /* = NMPC.setup(), mpc.py:1789 */
void hilo_nmpc_destroy(hilo_nmpc* h);
int hilo_nmpc_dims(const hilo_nmpc* h, int* n_v, int* n_g, int* nx, int* nu, int* np);
int hilo_nmpc_reset_warm_start(hilo_nmpc* h);
/* The other two vectors of the reference's solver result (`self._nlp_solution = sol`, mpc.py:722-723): bound multipliers
   lam_x [batch][n_v] (layout of v; CasADi's sign z_U - z_L; 0 for the pinned x_0) and constraint values g [batch][n_g] (layout
   of lam_g; written when the solve is given a lam_g buffer).  DEVICE buffers the following solves write, or NULL to stop.
   Layouts with a collocation output pass do not fill them (HILO leaves the buffers untouched). */
int hilo_nmpc_set_aux_outputs(hilo_nmpc* h, double* g, double* lam_x);
/* Sharded batches (one process per GPU, hilo_mpc_amd/dist.py): let the solve write row b = [u0 (nu) | status | iterations]
   (fp64) of the device table [batch][stride] itself - the send buffer of the per-step result gather.  NULL switches it off.
   Honoured by plain tracking problems; the reference has no counterpart (single instance, no batching). */
int hilo_nmpc_set_gather(hilo_nmpc* h, double* table, int stride);
/* Closed loops whose plant is the controller's own model (benchmarks, simulations; hilo_nmpc_plant_step): the solve advances the
   plant itself - row b of x_next [B][nx] receives Phi(x0_b, u_0, p_b) for the input it has just computed (one launch per step
   instead of two).  x_next may be the x0 buffer of the call (in-place closed loop); NULL switches it off.  HILO_ENOTSUP for problem
   kinds whose kernel does not offer it (general / run-time compiled policies, the hybrid model, long horizons). */
int hilo_nmpc_set_plant_out(hilo_nmpc* h, double* x_next);
/* lbx / ubx of the reference's solver call, `self._solver(x0=v0, lbx=self._v_lb, ubx=self._v_ub, ...)` (mpc.py:722): DEVICE rows
   [batch][n_v] in the layout of v (scaled variables, original bound values: IPOPT's bound_relax_factor is applied by the solve),
   read by every following hilo_nmpc_solve; NULL, NULL = back to the bounds of the description.  Per instance and per call, so that
   a caller can move bounds between steps as the reference does (mpc.py:797-807).  The entries of a pinned x_0 are ignored (the
   solve pins x_0 to the `x0` argument, which is what mpc.py:801-802 writes into both arrays); lb < ub elsewhere. */
int hilo_nmpc_set_var_bounds(hilo_nmpc* h, const double* lbx, const double* ubx);
/* optimize(fix_x0=...) of mpc.py:797-807: 1 (default) pins x_0 to the measured state; 0 leaves x_0 free inside the state
   box [x_lb, x_ub] (the `x0` argument of hilo_nmpc_solve is then ignored, the start value comes from the warm start / guess).
   Synchronises the device when the setting changes. */
int hilo_nmpc_set_fix_x0(hilo_nmpc* h, int fix_x0);
/* optimize(fix_x0=False, x0_lb=..., x0_ub=...) of mpc.py:803-807: an own box for x_0 (HOST pointers [nx], original units;
   NULL / NULL restores the state box).  Synchronises the device when the setting changes. */
int hilo_nmpc_set_x0_box(hilo_nmpc* h, const double* x0_lb_host, const double* x0_ub_host);
/* One optimize() for `batch` independent instances (mpc.py:744-857).
   v layout = the reference's decision vector [x_0..x_N | u_0..u_{N-1}] in scaled variables (mpc.py:1462-1485). */
int hilo_nmpc_solve(hilo_nmpc* h, int64_t batch,
                    const double* x0,                 /* [B][nx]  measured state, original units (mpc.py:801) */
                    const double* p, int64_t p_stride,/* [B][np]  constant parameters `cp` (mpc.py:496-497) */
                    const double* v0,                 /* [B][n_v] initial guess or NULL -> previous solution
                                                         (warm start, mpc.py:725-726) / tiled guess (:1468-1482) */
                    const double* u_old,              /* [B][nu]  previous input for the change penalty or NULL */
                    double* v_opt,                    /* [B][n_v] */
                    double* f_opt,                    /* [B] */
                    double* lam_g,                    /* [B][n_g] multipliers of g (sign: L = f + lam^T g) or NULL */
                    double* u0,                       /* [B][nu]  first input, un-scaled (mpc.py:856) */
                    int32_t* status,                  /* [B] HILO_STATUS_* (optimizer.py:1093-1104) */
                    int32_t* iters,                   /* [B] interior-point iterations */
                    double* kkt,                      /* [B] scaled optimality error at the returned point or NULL */
                    void* stream);
/* optimize() with per-stage data (handles created with desc.time_varying): stage_data[b][k] = [zref_k (nz, scaled like the
   references of QuadraticCost, modeling.py:329) | p_k (np)] for k = 0..N; row N holds the terminal reference in its first nx
   entries.  sd_stride = 0 shares one table among the batch, else >= (N+1)(nz+np).  The constant references of the desc are
   not used. */
int hilo_nmpc_solve_tv(hilo_nmpc* h, int64_t batch, const double* x0, const double* stage_data, int64_t sd_stride,
                       const double* v0, const double* u_old, double* v_opt, double* f_opt, double* lam_g, double* u0,
                       int32_t* status, int32_t* iters, double* kkt, void* stream);
/* developer aid: per-phase shader-clock totals of instance 0 (derivatives, errors, Riccati, step, line search,
   update, number of factorisations, number of line-search trial points); enable != 0 starts collecting,
   cycles_host[8] (may be NULL) receives the last launch's counters */
int hilo_nmpc_profile(hilo_nmpc* h, int enable, long long* cycles_host);
/* x+ = Phi(x, u, p) with the controller's own shooting map: closed-loop harness (control_loop.py:343-396) */
int hilo_nmpc_plant_step(hilo_nmpc* h, int64_t batch, const double* x, const double* u, const double* p,
                         int64_t p_stride, double* x_next, void* stream);

/* ------------------------------------------------------------------------------------------------------- */
/* MHE: batched moving-horizon estimation                                                                   */
/* replaces `ca.nlpsol("solver",'ipopt',...)` built at hilo_mpc/modules/estimator/mhe.py:782-790 and called as */
/* `solver(x0=v0, lbx, ubx, lbg, ubg, p=param)` at mhe.py:375 by `MovingHorizonEstimator.estimate`; transcription  */
/* mhe.py:596-760 for a pre-discretised model + `integration_method='discrete'`, state noise, pinned parameters   */
/* ------------------------------------------------------------------------------------------------------- */
typedef struct hilo_mhe hilo_mhe;

typedef struct hilo_mhe_desc {
  int32_t model_id, N, erk_order, n_sub, max_iter, acceptable_iter;
  double dt, tol, acceptable_tol, mu_init, bound_relax_factor;   /* 0 / <0 -> IPOPT defaults, see hilo_nmpc_desc */
  /* HOST pointers (NULL = zero weight / no bound / unit scaling / zero guess); costs act on un-scaled quantities
     (hilo_mpc/util/modeling.py:665-672) */
  const double* Wx;   /* [nx][nx] arrival weight   (modeling.py:747-777, mhe.py:742-745) */
  const double* Wy;   /* [ny][ny] measurement weight (modeling.py:686-712) */
  const double* Ww;   /* [nx][nx] state-noise weight (modeling.py:735-745); NULL: an estimator WITHOUT state noise (no w block in v,
                         mhe.py:599) - models given as source (user_source) */
  const double* x_lb; const double* x_ub; const double* w_lb; const double* w_ub;   /* original units */
  const double* x_scaling; const double* w_scaling; const double* u_scaling;
  const double* x_guess; const double* w_guess;
  /* ---- parameter estimation (mhe.py:614-623: the parameters head the decision vector with bounds p_lb / p_ub and guess
     p_guess; arrival term (p - p_arrival)^T Wp (p - p_arrival), modeling.py:747-777).  estimate_parameters = 0: all
     parameters are pinned to the `p` handed to hilo_mhe_estimate.  != 0: parameter j is estimated iff p_lb[j] < p_ub[j]
     (NULL bound = unbounded = estimated), `p` then carries p_arrival for the estimated and the value for the pinned ones;
     the estimate is the prefix of v_opt (scaled by p_scaling). ---- */
  int32_t estimate_parameters;
  int32_t reserved;
  const double* Wp;          /* [np][np] */
  const double* p_lb; const double* p_ub; const double* p_scaling; const double* p_guess;   /* [np], original units */
  /* ---- run-time compiled estimator (round 3): the model as HIP source of `UserModel` (hilo_nmpc_desc.user_source: the
     functor emitted from the model's expressions, or the alias of a zoo functor), compiled with hiprtc around the estimator's
     policy at create (what `MovingHorizonEstimator.setup` -> `ca.nlpsol` does with the CasADi graph, mhe.py:782-790).
     model_id = HILO_MODEL_USER takes the dimensions below.  collocation_degree > 0: the reference's DEFAULT transcription
     (mhe.py:512-561; needs a continuous model and user_source): v gains the collocation block behind the noise block,
     [p | x | w | ip_0..ip_{N-1}] (mhe.py:657-671), lam_g the per-stage rows [collocation (d nx) | continuity (nx)]. ---- */
  const char* user_source;
  int32_t user_nx, user_nu, user_np, user_ny, user_discrete, collocation_degree;
  const double* coll_A;      /* [d][d]  Runge-Kutta matrix of the collocation method (hilo_nmpc_desc.coll_A) */
  const double* coll_D;      /* [d + 1] continuity weights */
  /* ---- stage constraint of the estimator (`mhe.stage_constraint`, mhe.py:498-508, :536-553, :749-757): n_con expressions of the
     SCALED states and parameters, compiled into `UserFun::con(x, p, c)` behind UserModel in user_source; hard rows
     con_lb <= c <= con_ub at every node k < N and, under collocation, at every collocation point.  (The reference's soft branch
     cannot run in its estimator: the penalty function is never created.) ---- */
  int32_t n_con; int32_t reserved6;
  const double* con_lb; const double* con_ub;   /* [n_con]; -inf / +inf allowed on one side */
} hilo_mhe_desc;

int hilo_mhe_create(const hilo_mhe_desc* desc, int device, hilo_mhe** out);          /* = setup(), mhe.py:418 */
void hilo_mhe_destroy(hilo_mhe* h);
int hilo_mhe_dims(const hilo_mhe* h, int* n_v, int* n_g, int* nx, int* nu, int* np, int* ny);
int hilo_mhe_reset_warm_start(hilo_mhe* h);
/* One estimate() for `batch` independent estimators over a full window (mhe.py:311-416).
   v layout = the reference's decision vector [p | x_0..x_N | w_0..w_{N-1} (| collocation states)] (scaled, mhe.py:614-671). */
int hilo_mhe_estimate(hilo_mhe* h, int64_t batch,
                      const double* x_arrival,            /* [B][nx]   arrival guess, original units (mhe.py:347-351) */
                      const double* p, int64_t p_stride,  /* [B][np]   pinned model parameters */
                      const double* u_meas,               /* [B][N][nu] = param['u_meas'] transposed (mhe.py:361-362) */
                      const double* y_meas,               /* [B][N][ny] = param['y_meas'] transposed (mhe.py:358-359) */
                      const double* v0,                   /* [B][n_v] or NULL -> previous solution (mhe.py:385) / guess */
                      double* v_opt, double* f_opt, double* lam_g,
                      double* x_opt,                      /* [B][nx]  x_N un-scaled: one-step-ahead state (mhe.py:381-384) */
                      int32_t* status, int32_t* iters, double* kkt, void* stream);

/* ------------------------------------------------------------------------------------------------------- */
/* LMPC: batched dense convex QP                                                                            */
/* replaces `ca.conic("solver",'qpoases',{'h': H.sparsity(),'a': Aeq.sparsity()})` built at                    */
/* hilo_mpc/modules/controller/mpc.py:2268-2276 and called as                                                */
/* `solver(h=H, g=g, a=Ad, lbx=v_lb, ubx=v_ub, lba=Ad_lb, uba=Ad_ub)` at mpc.py:2374 by `LMPC.optimize`         */
/*     min 1/2 x^T H x + g^T x   s.t.  lba <= A x <= uba (rows must be equalities),  lbx <= x <= ubx          */
/* ------------------------------------------------------------------------------------------------------- */
typedef struct hilo_qp hilo_qp;
int hilo_qp_create(int n, int m, int device, hilo_qp** out);
void hilo_qp_destroy(hilo_qp* h);
int hilo_qp_set_options(hilo_qp* h, double tol /* <=0 keeps 1e-12 */, int max_iter /* <=0 keeps 100 */);
/* The QP has the stage shape of `LMPC.setup` (mpc.py:2198-2266): v = [x_0 .. x_N | u_0 .. u_{N-1}], H block diagonal over the  */
/* stages (:2252-2256), rows k: A_k x_k + B_k u_k - x_{k+1} = b_k (:2209-2240; NOT the `kron(B, I_N)` input block of :2243),      */
/* x_0 - and only x_0 - pinned by lbx == ubx (:2361-2362).  hilo_qp_solve then takes its Newton steps by a Riccati recursion over */
/* the stages (csrc/hilo_qp_ocp.h) instead of the dense Schur complement; same arguments, same iteration, same outputs.          */
/* *used = 1 if a stage kernel exists for (nx, nu, N <= 63), 0 = the dense kernels stay.  N = 0 withdraws the declaration.        */
int hilo_qp_set_stages(hilo_qp* h, int nx, int nu, int N, int* used);
int hilo_qp_solve(hilo_qp* h, int64_t batch,
                  const double* H, int64_t h_stride,       /* [B][n][n] row-major (stride 0 = shared) */
                  const double* g, int64_t g_stride,       /* [B][n] */
                  const double* A, int64_t a_stride,       /* [B][m][n] */
                  const double* lbx, const double* ubx, int64_t bx_stride, /* [B][n]; lbx == ubx pins a variable */
                  const double* lba, const double* uba, int64_t ba_stride, /* [B][m] */
                  double* x,                               /* [B][n] */
                  double* f,                               /* [B] */
                  double* lam_a,                           /* [B][m] or NULL;  H x + g + A^T lam_a + lam_x = 0 */
                  double* lam_x,                           /* [B][n] or NULL */
                  int32_t* status, int32_t* iters, void* stream);
/* The same solve with the first `npin` variables FIXED at xpin[b][0..npin) - what `LMPC.optimize` does with the measured state:    */
/* `self._lbx[:nx] = self._ubx[:nx] = x0` before every solver call (mpc.py:2361-2362).  lbx / ubx may then be ONE row shared by the   */
/* batch (bx_stride = 0); their first npin entries are not read.  One launch per step instead of two bound-row writes + the solve.    */
int hilo_qp_solve_pinned(hilo_qp* h, int64_t batch, const double* H, int64_t h_stride, const double* g, int64_t g_stride,
                         const double* A, int64_t a_stride, const double* lbx, const double* ubx, int64_t bx_stride,
                         const double* xpin, int npin, int64_t xpin_stride /* >= npin */, const double* lba, const double* uba,
                         int64_t ba_stride, double* x, double* f, double* lam_a, double* lam_x, int32_t* status, int32_t* iters,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HILO_HIP_H */
