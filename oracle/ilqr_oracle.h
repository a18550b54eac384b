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

#define ORC_MAXN 32
#define ORC_MAXM 32

/* ---- solver constants: include/ilqr.h:14-24, include/boxqp.h:19-24,63, finite_diff.h:9 ---- */
#define ORC_MAX_ITER 100
#define ORC_NALPHA 11

/* solver tunables of include/ilqr.h:14-24 (process-wide; defaults 1e-6, 1e-6, 1.6, 1e11, 1e-8, 0) */
void orc_set_params(double tol_fun, double tol_grad, double lambda_factor, double lambda_max,
                    double lambda_min, double z_min);

enum { ORC_MODEL_ACROBOT = 0, ORC_MODEL_DOUBLE_INTEGRATOR = 1, ORC_MODEL_LQ = 2 };

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
  double u_min[ORC_MAXM], u_max[ORC_MAXM];
  void (*dynamics)(const struct orc_model*, const double* x, const double* u, double* dx);
  double (*cost)(const struct orc_model*, const double* x, const double* u);
  double (*final_cost)(const struct orc_model*, const double* x);
  /* parameters */
  double goal[ORC_MAXN];
  /* LQ model (synthetic config 5): xdot = A x + B u ; cost 0.5(x'Qx+u'Ru) ; final 0.5 x'Qf x */
  const double *A, *Bm, *Q, *R, *Qf;
} orc_model;

void orc_model_init_acrobot(orc_model* m);                               /* include/acrobot.h */
void orc_model_init_double_integrator(orc_model* m, const double* goal); /* include/double_integrator.h */
void orc_model_init_lq(orc_model* m, int nx, int nu, const double* A, const double* Bm,
                       const double* Q, const double* R, const double* Qf, double umin,
                       double umax);
void orc_integrate_dynamics(const orc_model* m, const double* x, const double* u, double dt,
                            double* x1); /* include/model.h:12-15 */

/* ---- box-QP (src/boxqp.cpp, include/boxqp.h) ---- */
void orc_clamp_to_limits(int n, const double* x, const double* lo, const double* hi, double* out);
double orc_quad_cost(int n, const double* Q, const double* c, const double* x);
/* returns failed flag; x_opt/v_opt only written when the reference writes them */
int orc_quadclamp_line_search(int n, const double* x0, const double* dir, const double* Q,
                              const double* c, const double* lo, const double* hi, double* x_opt,
                              double* v_opt, int* n_steps);
/* R_free: nfree x nfree upper factor, column-major with leading dimension nfree. Returns result code. */
int orc_boxqp(int n, const double* Q, const double* c, const double* x0, const double* lo,
              const double* hi, double* x_opt, int* v_free, double* R_free, int* nfree_out,
              int* iters_out);
/* Eigen 3.3.4 llt_inplace<Lower>::unblocked (Cholesky/LLT.h:302-325). Returns -1 or failing k. */
int orc_llt_lower_unblocked(int n, double* A /* n x n col-major, in place */);

/* ---- finite differences (include/finite_diff.h, src/derivatives.cpp) ---- */
typedef struct orc_traj {
  int nx, nu, T;
  double dt;
  double* x0;  /* nx */
  double* xs;  /* (T+1)*nx */
  double* us;  /* T*nu */
  double* fx;  /* (T+1)*nx*nx */
  double* fu;  /* (T+1)*nx*nu */
  double* cx;  /* (T+1)*nx */
  double* cu;  /* (T+1)*nu */
  double* cxx; /* (T+1)*nx*nx */
  double* cxu; /* (T+1)*nx*nu */
  double* cuu; /* (T+1)*nu*nu */
  double* Vx;  /* (T+1)*nx */
  double* Vxx; /* (T+1)*nx*nx */
  double* k;   /* T*nu */
  double* K;   /* T*nu*nx (each nu x nx col-major) */
  double dV[2];
  double cost_s;
  double lambda, dlambda; /* file-statics of include/ilqr.h:17-18, here per solve */
  int has_gains;          /* K.size()>0, src/ilqr_core.cpp:316 */
  int iters;              /* outer iterations started */
  int status;
  double gnorm;
  int last_alpha_idx; /* accepted alpha index of the last line search, -1 = NO STEP */
  int n_backward;     /* number of backward_pass() calls (census) */
  int n_rollouts;     /* number of forward_pass() calls */
  void* owned;        /* allocation backing the arrays */
} orc_traj;

orc_traj* orc_traj_alloc(int nx, int nu, int T, double dt);
void orc_traj_free(orc_traj* s);

double orc_forward_pass(const orc_model* m, orc_traj* s, const double* x0, const double* u);
double orc_init_traj(const orc_model* m, orc_traj* s, const double* x0, const double* u0);
void orc_get_dynamics_derivatives(const orc_model* m, orc_traj* s);
void orc_get_cost_derivatives(const orc_model* m, orc_traj* s);
void orc_get_cost_2nd_derivatives(const orc_model* m, orc_traj* s);
void orc_compute_derivatives(const orc_model* m, orc_traj* s); /* the three above, ilqr_core.cpp:115-120 */
int orc_backward_pass(const orc_model* m, orc_traj* s);
double orc_gradient_norm(const orc_traj* s);
/* one line search as ilqr_core.cpp:184-226; returns accepted alpha index or -1. */
int orc_line_search(const orc_model* m, orc_traj* s, double* new_cost, double* dcost,
                    double* expected);
/* runs generate_trajectory() (ilqr_core.cpp:79-302) for at most max_iters outer iterations
 * (max_iters<=0 or >100 -> 100).  fixed_work!=0 disables the three termination tests (bench mode). */
int orc_generate_trajectory(const orc_model* m, orc_traj* s, int max_iters, int fixed_work,
                            double* cost_log /* optional [max_iters] */);
/* one outer iteration body; returns 1 if the loop would break */
int orc_iterate_once(const orc_model* m, orc_traj* s, int* flg_change, int fixed_work);

/* ---- batched drivers over B independent trajectories (OpenMP over b) ---- */
typedef struct orc_batch_result {
  int dummy;
} orc_batch_result;

/* Canonical batch layouts (shared with the C ABI of include/ilqr_amd.h):
 *   x0 [B][nx], u0 [B][T][nu], xs [B][T+1][nx], us [B][T][nu], k [B][T][nu],
 *   K [B][T][nu*nx] (col-major nu x nx), fx [B][T+1][nx*nx] ... cuu [B][T+1][nu*nu]. */
int orc_batch_solve(const orc_model* m, int B, int T, double dt, const double* x0,
                    const double* u0, int max_iters, int fixed_work, int nthreads,
                    double* xs_out, double* us_out, double* k_out, double* K_out,
                    double* cost_out, int* iters_out, int* status_out, double* lambda_out);

/* n_iters outer iterations from a given state (see the .c); every output pointer may be NULL */
int orc_batch_iterate_from(const orc_model* m, int B, int T, double dt, const double* x0,
                           const double* xs, const double* us, const double* k, const double* K,
                           const double* cost, const double* lambda, const double* dlambda,
                           int n_iters, int fixed_work, int nthreads, double* xs_out,
                           double* us_out, double* k_out, double* K_out, double* cost_out,
                           int* iters_out, int* status_out, double* lambda_out, double* dlambda_out,
                           int* alpha_out, double* gnorm_out, double* dV_out);

/* teacher-forced single stages over a batch */
int orc_batch_rollout(const orc_model* m, int B, int T, double dt, const double* x0,
                      const double* u, const double* xs_nom /* or NULL = open loop */,
                      const double* K /* or NULL */, int nthreads, double* xs_out,
                      double* us_out, double* cost_out);
int orc_batch_derivatives(const orc_model* m, int B, int T, double dt, const double* xs,
                          const double* us, int nthreads, double* fx, double* fu, double* cx,
                          double* cu, double* cxx, double* cxu, double* cuu);
int orc_batch_backward(const orc_model* m, int B, int T, const double* us, const double* fx,
                       const double* fu, const double* cx, const double* cu, const double* cxx,
                       const double* cxu, const double* cuu, const double* k_prev /* [B][T][nu] warm start */,
                       const double* lambda /* [B] */, int nthreads, double* k_out,
                       double* K_out, double* dV_out /* [B][2] */, int* diverge_out /* [B] */,
                       double* Vx0_out /* [B][nx] or NULL */, double* Vxx0_out /* [B][nx*nx] or NULL */);

#ifdef __cplusplus
}
#endif
#endif
