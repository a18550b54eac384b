/*
 * ilqr_oracle.h -- CPU oracle for the batched-iLQR hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm of kazuotani14/iLQR (the reference, mounted
 * read-only at /root/reference in the build container).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (ilqr_amd/) never links, imports or
 * calls anything in oracle/.  Every function cites the reference file:line it follows.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - box-QP, clamped line search, quadCost, clamp, the two shipped models and the three
 *     finite-difference operators are checked against the REAL reference code compiled from
 *     /root/reference (oracle/_ref, built by oracle/Makefile) in tests/test_oracle_vs_ref.py,
 *     and against the known answers of the reference's own unit tests (test/test_boxqp.cpp,
 *     test/test_finite_diff.cpp, test/test_dynamicsmodels.cpp, test/test_ilqr_forward_pass.cpp).
 *   - class iLQR (src/ilqr_core.cpp, src/derivatives.cpp) cannot be compiled here without
 *     writing a stand-in for the absent gtest header it includes (include/ilqr.h:12), so its
 *     glue (forward_pass, backward_pass, derivative sweeps, outer loop) is pinned by the golden
 *     anchors SURVEY.md section 8(c) records from the reference binary (initial costs, first
 *     backward-pass dV/k/K, final cost of the canonical 100-iteration acrobot solve, the
 *     iteration count of the integrator solve) -- tests/test_oracle_anchors.py.
 *
 * Conventions: all matrices column-major (Eigen MatrixXd default); T = number of transitions
 * (= u0.size(), src/ilqr_core.cpp:12); state arrays have T+1 entries.
 */
#ifndef ILQR_ORACLE_H_
#define ILQR_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Arithmetic types.  The oracle is compiled three times from this one source (oracle/Makefile):
 *   liboracle_ilqr.so      ORC_REAL = double       the reference's arithmetic (it is fp64 throughout)
 *   liboracle_ilqr_f32.so  ORC_REAL = float        the twin of the product's fp32 mode (BASELINE configs[3]):
 *                                                  every per-knot quantity and the box-QP in float,
 *                                                  per-trajectory accumulators (cost, dV, gnorm, lambda) in double
 *   liboracle_ilqr_f80.so  ORC_REAL = long double  the same algorithm in x87 extended precision (64-bit
 *                                                  mantissa): the "exact" answer the parity tests measure
 *                                                  BOTH the fp64 oracle and the device against, so that a
 *                                                  per-knot gain error can be told from ill-conditioning
 * orc_real: states, controls, derivative records, gains, everything inside a Riccati step / box-QP.
 * orc_acc : per-trajectory scalars accumulated over the horizon or carried across iterations.
 * Scalars passed BY VALUE across this API (dt, limits, tunables) are plain double in every build. */
#ifndef ORC_REAL
#define ORC_REAL double
#endif
#ifndef ORC_ACC
#define ORC_ACC double
#endif
typedef ORC_REAL orc_real;
typedef ORC_ACC orc_acc;
typedef double orc_f64;
/* arithmetic of the finite-difference sweep: orc_real, except in the fp32 build (double; see orc_models.inc) */
#ifdef ORC_FD_IS_DOUBLE
typedef double orc_fd;
#else
typedef ORC_REAL orc_fd;
#endif

#define ORC_MAXN 32
#define ORC_MAXM 32

/* ---- solver constants: include/ilqr.h:14-24, include/boxqp.h:19-24,63, finite_diff.h:9 ---- */
#define ORC_MAX_ITER 100
#define ORC_NALPHA 11

/* solver tunables of include/ilqr.h:14-24 (process-wide; defaults 1e-6, 1e-6, 1.6, 1e11, 1e-8, 0) */
void orc_set_params(orc_f64 tol_fun, orc_f64 tol_grad, orc_f64 lambda_factor, orc_f64 lambda_max,
                    orc_f64 lambda_min, orc_f64 z_min);

/* opt-in fixes (process-wide; 0 = the reference as it is): bit 0 clamped rollout, bit 1 Cholesky failure ends the box-QP */
void orc_set_fixes(int bits);

enum { ORC_MODEL_ACROBOT = 0, ORC_MODEL_DOUBLE_INTEGRATOR = 1, ORC_MODEL_LQ = 2, ORC_MODEL_CHAIN = 3 };

/* status of a solve (where the outer loop of src/ilqr_core.cpp:103-288 left) */
enum {
  ORC_STATUS_RUNNING = 0,
  ORC_STATUS_CONVERGED_GRAD = 1, /* ilqr_core.cpp:154-159 */
  ORC_STATUS_CONVERGED_COST = 2, /* ilqr_core.cpp:257-262 */
  ORC_STATUS_LAMBDA_MAX = 3,     /* ilqr_core.cpp:276-281 */
  ORC_STATUS_MAX_ITER = 4        /* loop ran maxIter times, ilqr_core.cpp:103 */
};

/* The Model plugin interface of include/model.h:6-21 as a C vtable. */
typedef struct orc_model {
  int id;
  int nx, nu; /* x_dims, u_dims */
  orc_real u_min[ORC_MAXM], u_max[ORC_MAXM];
  void (*dynamics)(const struct orc_model*, const orc_real* x, const orc_real* u, orc_real* dx);
  orc_real (*cost)(const struct orc_model*, const orc_real* x, const orc_real* u);
  orc_real (*final_cost)(const struct orc_model*, const orc_real* x);
  /* the same three functions in the finite differences' arithmetic (identical pointers unless fp32) */
  void (*dynamics_fd)(const struct orc_model*, const orc_fd* x, const orc_fd* u, orc_fd* dx);
  orc_fd (*cost_fd)(const struct orc_model*, const orc_fd* x, const orc_fd* u);
  orc_fd (*final_cost_fd)(const struct orc_model*, const orc_fd* x);
  /* parameters */
  orc_real goal[ORC_MAXN];
  /* LQ model (synthetic config 5): xdot = A x + B u ; cost 0.5(x'Qx+u'Ru) ; final 0.5 x'Qf x */
  const orc_real *A, *Bm, *Q, *R, *Qf;
} orc_model;

void orc_model_init_acrobot(orc_model* m);                               /* include/acrobot.h */
void orc_model_init_double_integrator(orc_model* m, const orc_real* goal); /* include/double_integrator.h */
void orc_model_init_lq(orc_model* m, int nx, int nu, const orc_real* A, const orc_real* Bm,
                       const orc_real* Q, const orc_real* R, const orc_real* Qf, orc_f64 umin,
                       orc_f64 umax);
/* a chain of N coupled pendulums (nx = 2 N <= 32, nu = N / 2): the oracle-side twin of examples/user_model_pendulum_chain.hpp */
void orc_model_init_chain(orc_model* m, int N, const orc_f64* params, orc_f64 umin, orc_f64 umax);
void orc_integrate_dynamics(const orc_model* m, const orc_real* x, const orc_real* u, orc_f64 dt,
                            orc_real* x1); /* include/model.h:12-15 */

/* ---- box-QP (src/boxqp.cpp, include/boxqp.h) ---- */
void orc_clamp_to_limits(int n, const orc_real* x, const orc_real* lo, const orc_real* hi, orc_real* out);
orc_real orc_quad_cost(int n, const orc_real* Q, const orc_real* c, const orc_real* x);
/* returns failed flag; x_opt/v_opt only written when the reference writes them */
int orc_quadclamp_line_search(int n, const orc_real* x0, const orc_real* dir, const orc_real* Q,
                              const orc_real* c, const orc_real* lo, const orc_real* hi, orc_real* x_opt,
                              orc_real* v_opt, int* n_steps);
/* R_free: nfree x nfree upper factor, column-major with leading dimension nfree. Returns result code. */
int orc_boxqp(int n, const orc_real* Q, const orc_real* c, const orc_real* x0, const orc_real* lo,
              const orc_real* hi, orc_real* x_opt, int* v_free, orc_real* R_free, int* nfree_out,
              int* iters_out);
/* Eigen 3.3.4 llt_inplace<Lower>::unblocked (Cholesky/LLT.h:302-325). Returns -1 or failing k. */
int orc_llt_lower_unblocked(int n, orc_real* A /* n x n col-major, in place */);

/* ---- finite differences (include/finite_diff.h, src/derivatives.cpp) ---- */
typedef struct orc_traj {
  int nx, nu, T;
  orc_real dt;   /* (rounded to the build's arithmetic once, at allocation) */
  orc_real* x0;  /* nx */
  orc_real* xs;  /* (T+1)*nx */
  orc_real* us;  /* T*nu */
  orc_real* fx;  /* (T+1)*nx*nx */
  orc_real* fu;  /* (T+1)*nx*nu */
  orc_real* cx;  /* (T+1)*nx */
  orc_real* cu;  /* (T+1)*nu */
  orc_real* cxx; /* (T+1)*nx*nx */
  orc_real* cxu; /* (T+1)*nx*nu */
  orc_real* cuu; /* (T+1)*nu*nu */
  orc_real* Vx;  /* (T+1)*nx */
  orc_real* Vxx; /* (T+1)*nx*nx */
  orc_real* k;   /* T*nu */
  orc_real* K;   /* T*nu*nx (each nu x nx col-major) */
  orc_acc dV[2];
  orc_acc cost_s;
  orc_acc lambda, dlambda; /* file-statics of include/ilqr.h:17-18, here per solve */
  int has_gains;          /* K.size()>0, src/ilqr_core.cpp:316 */
  int iters;              /* outer iterations started */
  int status;
  orc_acc gnorm;
  int last_alpha_idx; /* accepted alpha index of the last line search, -1 = NO STEP */
  int n_backward;     /* number of backward_pass() calls (census) */
  int n_rollouts;     /* number of forward_pass() calls */
  void* owned;        /* allocation backing the arrays */
} orc_traj;

orc_traj* orc_traj_alloc(int nx, int nu, int T, orc_f64 dt);
void orc_traj_free(orc_traj* s);

orc_acc orc_forward_pass(const orc_model* m, orc_traj* s, const orc_real* x0, const orc_real* u);
orc_acc orc_init_traj(const orc_model* m, orc_traj* s, const orc_real* x0, const orc_real* u0);
void orc_get_dynamics_derivatives(const orc_model* m, orc_traj* s);
void orc_get_cost_derivatives(const orc_model* m, orc_traj* s);
void orc_get_cost_2nd_derivatives(const orc_model* m, orc_traj* s);
void orc_compute_derivatives(const orc_model* m, orc_traj* s); /* the three above, ilqr_core.cpp:115-120 */
int orc_backward_pass(const orc_model* m, orc_traj* s);
orc_acc orc_gradient_norm(const orc_traj* s);
/* one line search as ilqr_core.cpp:184-226; returns accepted alpha index or -1. */
int orc_line_search(const orc_model* m, orc_traj* s, orc_acc* new_cost, orc_acc* dcost,
                    orc_acc* expected);
/* runs generate_trajectory() (ilqr_core.cpp:79-302) for at most max_iters outer iterations
 * (max_iters<=0 or >100 -> 100).  fixed_work!=0 disables the three termination tests (bench mode). */
int orc_generate_trajectory(const orc_model* m, orc_traj* s, int max_iters, int fixed_work,
                            orc_acc* cost_log /* optional [max_iters] */);
/* one outer iteration body; returns 1 if the loop would break */
int orc_iterate_once(const orc_model* m, orc_traj* s, int* flg_change, int fixed_work);

/* ---- batched drivers over B independent trajectories (OpenMP over b) ---- */
typedef struct orc_batch_result {
  int dummy;
} orc_batch_result;

/* Canonical batch layouts (shared with the C ABI of include/ilqr_amd.h):
 *   x0 [B][nx], u0 [B][T][nu], xs [B][T+1][nx], us [B][T][nu], k [B][T][nu],
 *   K [B][T][nu*nx] (col-major nu x nx), fx [B][T+1][nx*nx] ... cuu [B][T+1][nu*nu]. */
int orc_batch_solve(const orc_model* m, int B, int T, orc_f64 dt, const orc_real* x0,
                    const orc_real* u0, int max_iters, int fixed_work, int nthreads,
                    orc_real* xs_out, orc_real* us_out, orc_real* k_out, orc_real* K_out,
                    orc_acc* cost_out, int* iters_out, int* status_out, orc_acc* lambda_out);

/* n_iters outer iterations from a given state (see the .c); every output pointer may be NULL */
int orc_batch_iterate_from(const orc_model* m, int B, int T, orc_f64 dt, const orc_real* x0,
                           const orc_real* xs, const orc_real* us, const orc_real* k, const orc_real* K,
                           const orc_acc* cost, const orc_acc* lambda, const orc_acc* dlambda,
                           int n_iters, int fixed_work, int nthreads, orc_real* xs_out,
                           orc_real* us_out, orc_real* k_out, orc_real* K_out, orc_acc* cost_out,
                           int* iters_out, int* status_out, orc_acc* lambda_out, orc_acc* dlambda_out,
                           int* alpha_out, orc_acc* gnorm_out, orc_acc* dV_out);

/* teacher-forced single stages over a batch */
int orc_batch_rollout(const orc_model* m, int B, int T, orc_f64 dt, const orc_real* x0,
                      const orc_real* u, const orc_real* xs_nom /* or NULL = open loop */,
                      const orc_real* K /* or NULL */, int nthreads, orc_real* xs_out,
                      orc_real* us_out, orc_acc* cost_out);
int orc_batch_derivatives(const orc_model* m, int B, int T, orc_f64 dt, const orc_real* xs,
                          const orc_real* us, int nthreads, orc_real* fx, orc_real* fu, orc_real* cx,
                          orc_real* cu, orc_real* cxx, orc_real* cxu, orc_real* cuu);
int orc_batch_backward(const orc_model* m, int B, int T, const orc_real* us, const orc_real* fx,
                       const orc_real* fu, const orc_real* cx, const orc_real* cu, const orc_real* cxx,
                       const orc_real* cxu, const orc_real* cuu, const orc_real* k_prev /* [B][T][nu] warm start */,
                       const orc_acc* lambda /* [B] */, int nthreads, orc_real* k_out,
                       orc_real* K_out, orc_acc* dV_out /* [B][2] */, int* diverge_out /* [B] */,
                       orc_real* Vx0_out /* [B][nx] or NULL */, orc_real* Vxx0_out /* [B][nx*nx] or NULL */);

#ifdef __cplusplus
}
#endif
#endif
