/*
 * ilqr_oracle.c -- CPU oracle (plain C99) for the batched-iLQR hot path.
 * TEST INFRASTRUCTURE ONLY: see the header comment of ilqr_oracle.h for who may use it and how
 * its parity with the reference is pinned.  Reference = kazuotani14/iLQR at /root/reference;
 * every function below names the reference file:line it restates.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared  (see oracle/Makefile).
 * -ffp-contract=off because the reference is built for baseline x86-64 (CMakeLists.txt:5,17:
 * "-O3 -fopenmp", no -march) and therefore never fuses multiply-adds.
 */
#include "ilqr_oracle.h"

#include <math.h>
#include <tgmath.h> /* sin/cos/sqrt/fabs/fmax/fmin pick the flavour of their argument type */
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* include/finite_diff.h:9 `static const orc_real eps = 1e-3;`, src/derivatives.cpp:10 `#define eps2 1e-3` */
static const orc_fd EPS = 1e-3;
/* include/ilqr.h:14-24 */
/* (compile-time constants in the reference; settable here -- orc_set_params -- because the C ABI of
 * the product exposes them as ilqr_params and the tests exercise e.g. the gradient-norm exit with
 * a looser tolGrad.  Defaults are the reference's.) */
static orc_acc tolFun = 1e-6;
static orc_acc tolGrad = 1e-6;
static orc_acc lambdaFactor = 1.6;
static orc_acc lambdaMax = 1e11;
static orc_acc lambdaMin = 1e-8;
static orc_acc zMin = 0;
/* The opt-in "fixes" of the product (ILQR_FLAG_REFERENCE_FIXES), mirrored here so that they can be
 * parity-tested too.  OFF by default = the reference as it is.  bit 0: the rollout clamps every control into
 * [u_min, u_max] and integrates the clamped one (the commented "right way", ilqr_core.cpp:327-329); bit 1: a
 * failed Cholesky factorisation of Q[free,free] ends the box-QP with result -1 (the MATLAB original's
 * `indef`), which the backward pass reports as divergence -> lambda is raised (boxqp.cpp:85-88 ignores info());
 * bit 2: lambda regularises Vxx' ([Tassa 2012] eq. 10) instead of Quu (ilqr_core.cpp:365-367). */
static int referenceFixes = 0;
void orc_set_fixes(int bits) { referenceFixes = bits; }
void orc_set_params(orc_f64 tol_fun, orc_f64 tol_grad, orc_f64 lambda_factor, orc_f64 lambda_max,
                    orc_f64 lambda_min, orc_f64 z_min) {
  tolFun = tol_fun;
  tolGrad = tol_grad;
  lambdaFactor = lambda_factor;
  lambdaMax = lambda_max;
  lambdaMin = lambda_min;
  zMin = z_min;
}
static const orc_f64 Alpha[ORC_NALPHA] = {1.0000, 0.5012, 0.2512, 0.1259, 0.0631, 0.0316,
                                         0.0158, 0.0079, 0.0040, 0.0020, 0.0010};
/* include/boxqp.h:19-24 */
static const int qp_maxIter = 100;
/* (minGrad 1e-8, minRelImprove 1e-8, stepDec 0.6, minStep 1e-22, Armijo 0.1: orc_bw.inc) */

/* ------------------------------------------------------------------------------------------ */
/* Models                                                                                      */
/* ------------------------------------------------------------------------------------------ */

#define MT orc_real
#define MN(name) name
#include "orc_models.inc"
#undef MT
#undef MN
#ifdef ORC_FD_IS_DOUBLE /* fp32 build: a second, double-precision flavour for the finite differences */
#define MT orc_fd
#define MN(name) name##_fd
#include "orc_models.inc"
#undef MT
#undef MN
#else
#define acrobot_dynamics_fd acrobot_dynamics
#define acrobot_cost_fd acrobot_cost
#define acrobot_final_cost_fd acrobot_final_cost
#define dint_dynamics_fd dint_dynamics
#define dint_cost_fd dint_cost
#define dint_final_cost_fd dint_final_cost
#define lq_dynamics_fd lq_dynamics
#define lq_cost_fd lq_cost
#define lq_final_cost_fd lq_final_cost
#define chain_dynamics_fd chain_dynamics
#define chain_cost_fd chain_cost
#define chain_final_cost_fd chain_final_cost
#endif

void orc_model_init_acrobot(orc_model* m) {
  memset(m, 0, sizeof(*m));
  m->id = ORC_MODEL_ACROBOT;
  m->nx = 4; /* acrobot.h:27-28 */
  m->nu = 1;
  m->goal[0] = 3.1415; /* acrobot.h:21 */
  m->u_min[0] = -5;    /* acrobot.h:37 */
  m->u_max[0] = 5;
  m->dynamics = acrobot_dynamics;
  m->cost = acrobot_cost;
  m->final_cost = acrobot_final_cost;
  m->dynamics_fd = acrobot_dynamics_fd;
  m->cost_fd = acrobot_cost_fd;
  m->final_cost_fd = acrobot_final_cost_fd;
}

void orc_model_init_double_integrator(orc_model* m, const orc_real* goal) {
  memset(m, 0, sizeof(*m));
  m->id = ORC_MODEL_DOUBLE_INTEGRATOR;
  m->nx = 4; /* double_integrator.h:16-17 */
  m->nu = 2;
  for (int i = 0; i < 4; i++) m->goal[i] = goal[i];
  m->u_min[0] = m->u_min[1] = -0.5; /* double_integrator.h:25-26 */
  m->u_max[0] = m->u_max[1] = 0.5;
  m->dynamics = dint_dynamics;
  m->cost = dint_cost;
  m->final_cost = dint_final_cost;
  m->dynamics_fd = dint_dynamics_fd;
  m->cost_fd = dint_cost_fd;
  m->final_cost_fd = dint_final_cost_fd;
}

void orc_model_init_lq(orc_model* m, int nx, int nu, const orc_real* A, const orc_real* Bm,
                       const orc_real* Q, const orc_real* R, const orc_real* Qf, orc_f64 umin,
                       orc_f64 umax) {
  memset(m, 0, sizeof(*m));
  m->id = ORC_MODEL_LQ;
  m->nx = nx;
  m->nu = nu;
  m->A = A;
  m->Bm = Bm;
  m->Q = Q;
  m->R = R;
  m->Qf = Qf;
  for (int i = 0; i < nu; i++) {
    m->u_min[i] = umin;
    m->u_max[i] = umax;
  }
  m->dynamics = lq_dynamics;
  m->cost = lq_cost;
  m->final_cost = lq_final_cost;
  m->dynamics_fd = lq_dynamics_fd;
  m->cost_fd = lq_cost_fd;
  m->final_cost_fd = lq_final_cost_fd;
}

/* the pendulum chain (orc_models.inc): N links, params[8] = g/l, damping, coupling, w_theta, w_omega, w_u, final scale, target angle */
void orc_model_init_chain(orc_model* m, int N, const orc_f64* params, orc_f64 umin, orc_f64 umax) {
  memset(m, 0, sizeof(*m));
  m->id = ORC_MODEL_CHAIN;
  m->nx = 2 * N;
  m->nu = N / 2;
  for (int i = 0; i < 8; i++) m->goal[i] = (orc_real)params[i];
  for (int i = 0; i < m->nu; i++) {
    m->u_min[i] = umin;
    m->u_max[i] = umax;
  }
  m->dynamics = chain_dynamics;
  m->cost = chain_cost;
  m->final_cost = chain_final_cost;
  m->dynamics_fd = chain_dynamics_fd;
  m->cost_fd = chain_cost_fd;
  m->final_cost_fd = chain_final_cost_fd;
}

/* include/model.h:12-15: x1 = x + dynamics(x,u)*dt */
void orc_integrate_dynamics(const orc_model* m, const orc_real* x, const orc_real* u, orc_f64 dt_,
                            orc_real* x1) {
  const orc_real dt = (orc_real)dt_;
  orc_real dx[ORC_MAXN];
  m->dynamics(m, x, u, dx);
  for (int i = 0; i < m->nx; i++) x1[i] = x[i] + dx[i] * dt;
}
/* the same map in the finite differences' arithmetic (== the above except in the fp32 build) */
static void integrate_dynamics_fd(const orc_model* m, const orc_fd* x, const orc_fd* u, orc_fd dt, orc_fd* x1) {
  orc_fd dx[ORC_MAXN];
  m->dynamics_fd(m, x, u, dx);
  for (int i = 0; i < m->nx; i++) x1[i] = x[i] + dx[i] * dt;
}

/* ------------------------------------------------------------------------------------------ */
/* box-QP                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* (the functions of this section and the backward pass live in orc_bw.inc, included further down: they need the trajectory type) */

/* ------------------------------------------------------------------------------------------ */
/* Trajectory state                                                                            */
/* ------------------------------------------------------------------------------------------ */

orc_traj* orc_traj_alloc(int nx, int nu, int T, orc_f64 dt) {
  orc_traj* s = (orc_traj*)calloc(1, sizeof(orc_traj));
  const size_t n = (size_t)nx, m = (size_t)nu, T1 = (size_t)T + 1;
  size_t tot = n + T1 * n + T * m + T1 * (n * n + n * m + n + m + n * n + n * m + m * m) +
               T1 * (n + n * n) + T * m + T * m * n;
  orc_real* p = (orc_real*)calloc(tot, sizeof(orc_real));
  s->owned = p;
  s->nx = nx;
  s->nu = nu;
  s->T = T;
  s->dt = (orc_real)dt;
  s->x0 = p; p += n;
  s->xs = p; p += T1 * n;
  s->us = p; p += T * m;
  s->fx = p; p += T1 * n * n;
  s->fu = p; p += T1 * n * m;
  s->cx = p; p += T1 * n;
  s->cu = p; p += T1 * m;
  s->cxx = p; p += T1 * n * n;
  s->cxu = p; p += T1 * n * m;
  s->cuu = p; p += T1 * m * m;
  s->Vx = p; p += T1 * n;
  s->Vxx = p; p += T1 * n * n;
  s->k = p; p += T * m;
  s->K = p; p += T * m * n;
  s->lambda = 1; /* include/ilqr.h:17-18 */
  s->dlambda = 1;
  s->last_alpha_idx = -1;
  return s;
}
void orc_traj_free(orc_traj* s) {
  if (!s) return;
  free(s->owned);
  free(s);
}

/* src/ilqr_core.cpp:305-337.  `u` may alias s->us (init_traj passes `us` itself). */
orc_acc orc_forward_pass(const orc_model* m, orc_traj* s, const orc_real* x0, const orc_real* u) {
  const int n = s->nx, mu = s->nu, T = s->T;
  orc_acc total_cost = 0; /* the sum over the horizon is a per-trajectory accumulator (orc_acc) */
  orc_real x_curr[ORC_MAXN], u_curr[ORC_MAXM], xn[ORC_MAXN];
  orc_real* x_new = (orc_real*)malloc(sizeof(orc_real) * (size_t)(T + 1) * n);
  for (int i = 0; i < n; i++) x_curr[i] = x_new[i] = x0[i];
  for (int t = 0; t < T; t++) {
    for (int a = 0; a < mu; a++) u_curr[a] = u[t * mu + a]; /* :315 */
    if (s->has_gains) { /* :316  u += K[t]*(x_new[t]-xs[t]) */
      for (int a = 0; a < mu; a++) {
        orc_real acc = 0;
        for (int j = 0; j < n; j++) acc += s->K[(size_t)t * mu * n + a + mu * j] * (x_new[t * n + j] - s->xs[t * n + j]);
        u_curr[a] += acc;
      }
    }
    if (referenceFixes & 1) /* opt-in: "the right way", :327-329 */
      for (int a = 0; a < mu; a++) u_curr[a] = (m->u_max[a] < ((u_curr[a] < m->u_min[a]) ? m->u_min[a] : u_curr[a])) ? m->u_max[a] : ((u_curr[a] < m->u_min[a]) ? m->u_min[a] : u_curr[a]);
    for (int a = 0; a < mu; a++) s->us[t * mu + a] = u_curr[a]; /* :323, no clamping */
    total_cost += m->cost(m, x_curr, u_curr);                   /* :324 */
    orc_integrate_dynamics(m, x_curr, u_curr, s->dt, xn);      /* :325 */
    for (int i = 0; i < n; i++) x_curr[i] = x_new[(t + 1) * n + i] = xn[i];
  }
  memcpy(s->xs, x_new, sizeof(orc_real) * (size_t)(T + 1) * n); /* :334 */
  free(x_new);
  total_cost += m->final_cost(m, &s->xs[(size_t)T * n]); /* :335 */
  s->n_rollouts++;
  return total_cost;
}

/* src/ilqr_core.cpp:11-56 */
orc_acc orc_init_traj(const orc_model* m, orc_traj* s, const orc_real* x0, const orc_real* u0) {
  const int n = s->nx, mu = s->nu, T = s->T;
  for (int i = 0; i < n; i++) s->x0[i] = s->xs[i] = x0[i];
  memcpy(s->us, u0, sizeof(orc_real) * (size_t)T * mu);
  s->has_gains = 0; /* K.size()==0 on a fresh object, :316 */
  const orc_acc cost_i = orc_forward_pass(m, s, s->x0, s->us); /* :20 */
  /* :23-48 allocate + zero everything */
  memset(s->fx, 0, sizeof(orc_real) * (size_t)(T + 1) * n * n);
  memset(s->fu, 0, sizeof(orc_real) * (size_t)(T + 1) * n * mu);
  memset(s->cx, 0, sizeof(orc_real) * (size_t)(T + 1) * n);
  memset(s->cu, 0, sizeof(orc_real) * (size_t)(T + 1) * mu);
  memset(s->cxx, 0, sizeof(orc_real) * (size_t)(T + 1) * n * n);
  memset(s->cxu, 0, sizeof(orc_real) * (size_t)(T + 1) * n * mu);
  memset(s->cuu, 0, sizeof(orc_real) * (size_t)(T + 1) * mu * mu);
  memset(s->Vx, 0, sizeof(orc_real) * (size_t)(T + 1) * n);
  memset(s->Vxx, 0, sizeof(orc_real) * (size_t)(T + 1) * n * n);
  memset(s->k, 0, sizeof(orc_real) * (size_t)T * mu);
  memset(s->K, 0, sizeof(orc_real) * (size_t)T * mu * n);
  s->has_gains = 1; /* K.resize(T): K.size()>0 from now on */
  s->cost_s = cost_i;
  return cost_i;
}

/* ------------------------------------------------------------------------------------------ */
/* Finite differences                                                                          */
/* ------------------------------------------------------------------------------------------ */

/* All three sweeps run in orc_fd: the knot (x_t, u_t) is widened once, the model's _fd flavour is
 * evaluated at the perturbed points, differences and quotients are taken in orc_fd and the results are
 * rounded to orc_real when stored.  orc_fd == orc_real in the fp64 build (this is then the reference's
 * arithmetic, operation for operation) and in the fp80 build. */
static void knot_fd(const orc_traj* s, int t, orc_fd* x, orc_fd* u) {
  const int n = s->nx, mu = s->nu;
  for (int j = 0; j < n; j++) x[j] = s->xs[(size_t)t * n + j];
  for (int a = 0; a < mu; a++) u[a] = (t < s->T) ? s->us[(size_t)t * mu + a] : (orc_real)0.0; /* derivatives.cpp:35-38 */
}

/* src/derivatives.cpp:15-26 + include/finite_diff.h:35-47.  F = Euler map of include/model.h:12-15.
 * fx[T], fu[T] are never written (stay zero). */
void orc_get_dynamics_derivatives(const orc_model* m, orc_traj* s) {
  const int n = s->nx, mu = s->nu, T = s->T;
  const orc_fd dt = s->dt;
  orc_fd x[ORC_MAXN], u[ORC_MAXM], plus[ORC_MAXN], minus[ORC_MAXN], fp[ORC_MAXN], fm[ORC_MAXN];
  for (int t = 0; t < T; t++) {
    knot_fd(s, t, x, u);
    orc_real* fx = &s->fx[(size_t)t * n * n];
    orc_real* fu = &s->fu[(size_t)t * n * mu];
    for (int i = 0; i < n; i++) { /* finite_diff_jacobian(dyn_x, x[t]) */
      for (int j = 0; j < n; j++) plus[j] = minus[j] = x[j];
      plus[i] += EPS;
      minus[i] -= EPS;
      integrate_dynamics_fd(m, plus, u, dt, fp);
      integrate_dynamics_fd(m, minus, u, dt, fm);
      for (int r = 0; r < n; r++) fx[r + n * i] = (fp[r] - fm[r]) / (2 * EPS);
    }
    for (int i = 0; i < mu; i++) { /* finite_diff_jacobian(dyn_u, u[t]) */
      for (int j = 0; j < mu; j++) plus[j] = minus[j] = u[j];
      plus[i] += EPS;
      minus[i] -= EPS;
      integrate_dynamics_fd(m, x, plus, dt, fp);
      integrate_dynamics_fd(m, x, minus, dt, fm);
      for (int r = 0; r < n; r++) fu[r + n * i] = (fp[r] - fm[r]) / (2 * EPS);
    }
  }
}

/* src/derivatives.cpp:29-54 + include/finite_diff.h:22-33 */
void orc_get_cost_derivatives(const orc_model* m, orc_traj* s) {
  const int n = s->nx, mu = s->nu, T = s->T;
  orc_fd x[ORC_MAXN], plus[ORC_MAXN], minus[ORC_MAXN], ut[ORC_MAXM];
  for (int t = 0; t < T + 1; t++) {
    knot_fd(s, t, x, ut);
    orc_real* cx = &s->cx[(size_t)t * n];
    orc_real* cu = &s->cu[(size_t)t * mu];
    if (t < T) {
      for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) plus[j] = minus[j] = x[j];
        plus[i] += EPS;
        minus[i] -= EPS;
        cx[i] = (m->cost_fd(m, plus, ut) - m->cost_fd(m, minus, ut)) / (2 * EPS);
      }
      for (int i = 0; i < mu; i++) {
        for (int j = 0; j < mu; j++) plus[j] = minus[j] = ut[j];
        plus[i] += EPS;
        minus[i] -= EPS;
        cu[i] = (m->cost_fd(m, x, plus) - m->cost_fd(m, x, minus)) / (2 * EPS);
      }
    } else { /* :48-52 */
      for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) plus[j] = minus[j] = x[j];
        plus[i] += EPS;
        minus[i] -= EPS;
        cx[i] = (m->final_cost_fd(m, plus) - m->final_cost_fd(m, minus)) / (2 * EPS);
      }
      for (int i = 0; i < mu; i++) cu[i] = 0;
    }
  }
}

/* which: 0 = cost(v, other) [v is x], 1 = cost(other, v) [v is u], 2 = final_cost(v) */
static orc_fd eval_fn(const orc_model* m, const orc_fd* v, const orc_fd* other, int which) {
  if (which == 0) return m->cost_fd(m, v, other);
  if (which == 1) return m->cost_fd(m, other, v);
  return m->final_cost_fd(m, v);
}

/* include/finite_diff.h:67-86.  Diagonal entries use +-2 eps (pp[i] += eps twice) and
 * pm/mp = x+eps-eps, which is not bit-identical to x. */
static void fd_hessian(const orc_model* m, int nd, const orc_fd* x, const orc_fd* other,
                       int which, orc_real* out) {
  orc_fd pp[ORC_MAXN], pm[ORC_MAXN], mp[ORC_MAXN], mm[ORC_MAXN];
  for (int i = 0; i < nd; i++)
    for (int j = i; j < nd; j++) {
      for (int l = 0; l < nd; l++) pp[l] = pm[l] = mp[l] = mm[l] = x[l];
      pp[i] += EPS;
      pp[j] += EPS;
      pm[i] += EPS;
      pm[j] -= EPS;
      mp[i] -= EPS;
      mp[j] += EPS;
      mm[i] -= EPS;
      mm[j] -= EPS;
      const orc_fd v = (eval_fn(m, pp, other, which) - eval_fn(m, mp, other, which) -
                        eval_fn(m, pm, other, which) + eval_fn(m, mm, other, which)) /
                       (4 * EPS * EPS);
      out[i + nd * j] = out[j + nd * i] = (orc_real)v;
    }
}

/* src/derivatives.cpp:57-144 */
void orc_get_cost_2nd_derivatives(const orc_model* m, orc_traj* s) {
  const int n = s->nx, mu = s->nu, T = s->T;
  orc_fd x[ORC_MAXN], ut[ORC_MAXM];
  for (int t = 0; t < T + 1; t++) {
    knot_fd(s, t, x, ut);
    /* calculate_cxx :76-96 */
    fd_hessian(m, n, x, ut, (t < T) ? 0 : 2, &s->cxx[(size_t)t * n * n]);
    /* calculate_cuu :98-112 (also at t = T, with ut = 0) */
    fd_hessian(m, mu, ut, x, 1, &s->cuu[(size_t)t * mu * mu]);
    /* calculate_cxu :114-144 */
    orc_fd px[ORC_MAXN], mx[ORC_MAXN], pu[ORC_MAXM], mu_[ORC_MAXM];
    orc_real* cxu = &s->cxu[(size_t)t * n * mu];
    for (int i = 0; i < n; i++)
      for (int j = 0; j < mu; j++) {
        for (int l = 0; l < n; l++) px[l] = mx[l] = x[l];
        for (int l = 0; l < mu; l++) pu[l] = mu_[l] = ut[l];
        px[i] += EPS;
        mx[i] -= EPS;
        pu[j] += EPS;
        mu_[j] -= EPS;
        if (t < T)
          cxu[i + n * j] = (m->cost_fd(m, px, pu) - m->cost_fd(m, mx, pu) - m->cost_fd(m, px, mu_) + m->cost_fd(m, mx, mu_)) /
                           (4 * (EPS * EPS));
        else /* :140, "TODO this is wrong" in the reference; value unused downstream */
          cxu[i + n * j] = (m->final_cost_fd(m, px) - m->final_cost_fd(m, mx) - m->final_cost_fd(m, px) + m->final_cost_fd(m, mx)) /
                           (4 * (EPS * EPS));
      }
  }
}

void orc_compute_derivatives(const orc_model* m, orc_traj* s) { /* ilqr_core.cpp:115-120 */
  orc_get_dynamics_derivatives(m, s);
  orc_get_cost_derivatives(m, s);
  orc_get_cost_2nd_derivatives(m, s);
}

/* ------------------------------------------------------------------------------------------ */
/* Backward pass                                                                               */
/* ------------------------------------------------------------------------------------------ */

#define BWR orc_real
#define BWN(name) name
#include "orc_bw.inc" /* box-QP + backward pass in the build's arithmetic, under the exported names */
#undef BWR
#undef BWN
#ifdef ORC_BW_IS_DOUBLE /* fp32 build: the mixed backward pass the outer loop uses (orc_bw.inc) */
#define BWR double
#define BWN(name) name##_bw64
#include "orc_bw.inc"
#undef BWR
#undef BWN
#define ORC_BACKWARD_PASS orc_backward_pass_bw64
#else
#define ORC_BACKWARD_PASS orc_backward_pass
#endif

/* src/ilqr_core.cpp:405-412: mean_t max_i |k_i|/(|u_i|+1) */
orc_acc orc_gradient_norm(const orc_traj* s) {
  const int mu = s->nu, T = s->T;
  orc_acc acc = 0;
  for (int t = 0; t < T; t++) {
    orc_real mx = -INFINITY;
    for (int a = 0; a < mu; a++) {
      const orc_real v = fabs(s->k[(size_t)t * mu + a]) / (fabs(s->us[(size_t)t * mu + a]) + 1);
      if (a == 0 || v > mx) mx = v;
    }
    acc += mx;
  }
  return acc / T;
}

/* ------------------------------------------------------------------------------------------ */
/* Outer loop                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* src/ilqr_core.cpp:184-226 */
int orc_line_search(const orc_model* m, orc_traj* s, orc_acc* new_cost_out, orc_acc* dcost_out,
                    orc_acc* expected_out) {
  const int n = s->nx, mu = s->nu, T = s->T;
  orc_real* x_old = (orc_real*)malloc(sizeof(orc_real) * (size_t)(T + 1) * n);
  orc_real* u_old = (orc_real*)malloc(sizeof(orc_real) * (size_t)T * mu);
  orc_real* u_ff = (orc_real*)malloc(sizeof(orc_real) * (size_t)T * mu);
  memcpy(x_old, s->xs, sizeof(orc_real) * (size_t)(T + 1) * n); /* :104 */
  memcpy(u_old, s->us, sizeof(orc_real) * (size_t)T * mu);
  int accepted = -1;
  orc_acc new_cost = 0, dcost = 0, expected = 0, z = 0;
  for (int ai = 0; ai < ORC_NALPHA; ai++) {
    const orc_real alpha = (orc_real)Alpha[ai]; /* what the rollout multiplies k with */
    const orc_acc alpha_s = Alpha[ai];          /* the literal of include/ilqr.h:24 in the scalar test below */
    for (int e = 0; e < T * mu; e++) u_ff[e] = s->us[e] + s->k[e] * alpha; /* :188-190 */
    new_cost = orc_forward_pass(m, s, s->x0, u_ff);                        /* :197 */
    dcost = s->cost_s - new_cost;                                          /* :199 */
    expected = -alpha_s * (s->dV[0] + alpha_s * s->dV[1]);                 /* :200 */
    if (expected > 0)
      z = dcost / expected;
    else
      z = (orc_acc)((0.0 < dcost) - (dcost < 0.0)); /* sgn, include/common.h:52 */
    if (z > zMin) {
      accepted = ai;
      break;
    }
    memcpy(s->xs, x_old, sizeof(orc_real) * (size_t)(T + 1) * n); /* :218-219 */
    memcpy(s->us, u_old, sizeof(orc_real) * (size_t)T * mu);
  }
  free(x_old);
  free(u_old);
  free(u_ff);
  if (new_cost_out) *new_cost_out = new_cost;
  if (dcost_out) *dcost_out = dcost;
  if (expected_out) *expected_out = expected;
  s->last_alpha_idx = accepted;
  return accepted;
}

/* One body of the for-loop of src/ilqr_core.cpp:103-288.  Returns 1 when the loop breaks. */
int orc_iterate_once(const orc_model* m, orc_traj* s, int* flg_change, int fixed_work) {
  /* STEP 1 :115-120 */
  if (*flg_change || fixed_work) {
    orc_compute_derivatives(m, s);
    *flg_change = 0;
  }
  /* STEP 2 :136-150 */
  int backPassDone = 0;
  while (!backPassDone) {
    const int diverge = ORC_BACKWARD_PASS(m, s);
    if (diverge != 0) {
      s->dlambda = fmax(s->dlambda * lambdaFactor, lambdaFactor);
      s->lambda = fmax(s->lambda * s->dlambda, lambdaMin);
      if (s->lambda > lambdaMax) break;
      continue;
    }
    backPassDone = 1;
  }
  /* :153-159 */
  s->gnorm = orc_gradient_norm(s);
  if (s->gnorm < tolGrad && s->lambda < 1e-5 && !fixed_work) {
    s->status = ORC_STATUS_CONVERGED_GRAD;
    return 1;
  }
  /* STEP 3 :175-226 */
  int fwdPassDone = 0;
  orc_acc new_cost = 0, dcost = 0, expected = 0;
  if (backPassDone) {
    fwdPassDone = orc_line_search(m, s, &new_cost, &dcost, &expected) >= 0;
  } else {
    s->last_alpha_idx = -1;
  }
  /* STEP 4 :242-282 */
  if (fwdPassDone) {
    s->dlambda = fmin(s->dlambda / lambdaFactor, 1 / lambdaFactor);
    s->lambda = s->lambda * s->dlambda * (s->lambda > lambdaMin);
    s->cost_s = new_cost;
    *flg_change = 1;
    if (dcost < tolFun && !fixed_work) {
      s->status = ORC_STATUS_CONVERGED_COST;
      return 1;
    }
  } else {
    s->dlambda = fmax(s->dlambda * lambdaFactor, lambdaFactor);
    s->lambda = fmax(s->lambda * s->dlambda, lambdaMin);
    if (s->lambda > lambdaMax && !fixed_work) {
      s->status = ORC_STATUS_LAMBDA_MAX;
      return 1;
    }
  }
  return 0;
}

/* src/ilqr_core.cpp:79-302 */
int orc_generate_trajectory(const orc_model* m, orc_traj* s, int max_iters, int fixed_work,
                            orc_acc* cost_log) {
  int flgChange = 1;
  if (max_iters <= 0 || max_iters > ORC_MAX_ITER) max_iters = ORC_MAX_ITER;
  s->status = ORC_STATUS_RUNNING;
  int iter;
  for (iter = 0; iter < max_iters; iter++) {
    s->iters = iter + 1;
    const int brk = orc_iterate_once(m, s, &flgChange, fixed_work);
    if (cost_log) cost_log[iter] = s->cost_s;
    if (brk) return s->status;
  }
  if (max_iters == ORC_MAX_ITER) s->status = ORC_STATUS_MAX_ITER;
  return s->status;
}

/* ------------------------------------------------------------------------------------------ */
/* Batched drivers                                                                             */
/* ------------------------------------------------------------------------------------------ */

static int pick_threads(int nthreads) {
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  return nthreads;
#else
  (void)nthreads;
  return 1;
#endif
}

int orc_batch_solve(const orc_model* m, int B, int T, orc_f64 dt, const orc_real* x0,
                    const orc_real* u0, int max_iters, int fixed_work, int nthreads,
                    orc_real* xs_out, orc_real* us_out, orc_real* k_out, orc_real* K_out,
                    orc_acc* cost_out, int* iters_out, int* status_out, orc_acc* lambda_out) {
  const int n = m->nx, mu = m->nu;
  const int nt = pick_threads(nthreads);
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
  for (int b = 0; b < B; b++) {
    orc_traj* s = orc_traj_alloc(n, mu, T, dt);
    orc_init_traj(m, s, &x0[(size_t)b * n], &u0[(size_t)b * T * mu]);
    orc_generate_trajectory(m, s, max_iters, fixed_work, 0);
    if (xs_out) memcpy(&xs_out[(size_t)b * (T + 1) * n], s->xs, sizeof(orc_real) * (size_t)(T + 1) * n);
    if (us_out) memcpy(&us_out[(size_t)b * T * mu], s->us, sizeof(orc_real) * (size_t)T * mu);
    if (k_out) memcpy(&k_out[(size_t)b * T * mu], s->k, sizeof(orc_real) * (size_t)T * mu);
    if (K_out) memcpy(&K_out[(size_t)b * T * mu * n], s->K, sizeof(orc_real) * (size_t)T * mu * n);
    if (cost_out) cost_out[b] = s->cost_s;
    if (iters_out) iters_out[b] = s->iters;
    if (status_out) status_out[b] = s->status;
    if (lambda_out) lambda_out[b] = s->lambda;
    orc_traj_free(s);
  }
  return 0;
}

/* n_iters bodies of the outer loop (orc_iterate_once) from a GIVEN state instead of from init_traj:
 * the nominal trajectory (xs, us, cost), the gains (k is the box-QP warm start of t = T-1, K feeds
 * the closed-loop rollouts) and lambda/dlambda.  flgChange starts at 1 (derivatives are recomputed),
 * i.e. the state is what an accepted iteration leaves behind.  This is how the per-iteration
 * teacher-forced parity tests step both sides from the same point. */
int orc_batch_iterate_from(const orc_model* m, int B, int T, orc_f64 dt, const orc_real* x0,
                           const orc_real* xs, const orc_real* us, const orc_real* k, const orc_real* K,
                           const orc_acc* cost, const orc_acc* lambda, const orc_acc* dlambda,
                           int n_iters, int fixed_work, int nthreads, orc_real* xs_out,
                           orc_real* us_out, orc_real* k_out, orc_real* K_out, orc_acc* cost_out,
                           int* iters_out, int* status_out, orc_acc* lambda_out, orc_acc* dlambda_out,
                           int* alpha_out, orc_acc* gnorm_out, orc_acc* dV_out) {
  const int n = m->nx, mu = m->nu;
  const size_t T1 = (size_t)T + 1;
  const int nt = pick_threads(nthreads);
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 1)
  for (int b = 0; b < B; b++) {
    orc_traj* s = orc_traj_alloc(n, mu, T, dt);
    memcpy(s->x0, &x0[(size_t)b * n], sizeof(orc_real) * n);
    memcpy(s->xs, &xs[(size_t)b * T1 * n], sizeof(orc_real) * T1 * n);
    memcpy(s->us, &us[(size_t)b * T * mu], sizeof(orc_real) * (size_t)T * mu);
    if (k) memcpy(s->k, &k[(size_t)b * T * mu], sizeof(orc_real) * (size_t)T * mu);
    if (K) memcpy(s->K, &K[(size_t)b * T * mu * n], sizeof(orc_real) * (size_t)T * mu * n);
    s->has_gains = 1;
    s->cost_s = cost[b];
    s->lambda = lambda[b];
    s->dlambda = dlambda[b];
    s->status = ORC_STATUS_RUNNING;
    int flgChange = 1;
    for (int it = 0; it < n_iters; it++) {
      s->iters = it + 1;
      if (orc_iterate_once(m, s, &flgChange, fixed_work)) break;
    }
    if (xs_out) memcpy(&xs_out[(size_t)b * T1 * n], s->xs, sizeof(orc_real) * T1 * n);
    if (us_out) memcpy(&us_out[(size_t)b * T * mu], s->us, sizeof(orc_real) * (size_t)T * mu);
    if (k_out) memcpy(&k_out[(size_t)b * T * mu], s->k, sizeof(orc_real) * (size_t)T * mu);
    if (K_out) memcpy(&K_out[(size_t)b * T * mu * n], s->K, sizeof(orc_real) * (size_t)T * mu * n);
    if (cost_out) cost_out[b] = s->cost_s;
    if (iters_out) iters_out[b] = s->iters;
    if (status_out) status_out[b] = s->status;
    if (lambda_out) lambda_out[b] = s->lambda;
    if (dlambda_out) dlambda_out[b] = s->dlambda;
    if (alpha_out) alpha_out[b] = s->last_alpha_idx;
    if (gnorm_out) gnorm_out[b] = s->gnorm;
    if (dV_out) {
      dV_out[2 * b] = s->dV[0];
      dV_out[2 * b + 1] = s->dV[1];
    }
    orc_traj_free(s);
  }
  return 0;
}

int orc_batch_rollout(const orc_model* m, int B, int T, orc_f64 dt, const orc_real* x0,
                      const orc_real* u, const orc_real* xs_nom, const orc_real* K, int nthreads,
                      orc_real* xs_out, orc_real* us_out, orc_acc* cost_out) {
  const int n = m->nx, mu = m->nu;
  const int nt = pick_threads(nthreads);
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static)
  for (int b = 0; b < B; b++) {
    orc_traj* s = orc_traj_alloc(n, mu, T, dt);
    if (xs_nom && K) {
      memcpy(s->xs, &xs_nom[(size_t)b * (T + 1) * n], sizeof(orc_real) * (size_t)(T + 1) * n);
      memcpy(s->K, &K[(size_t)b * T * mu * n], sizeof(orc_real) * (size_t)T * mu * n);
      s->has_gains = 1;
    }
    const orc_acc c = orc_forward_pass(m, s, &x0[(size_t)b * n], &u[(size_t)b * T * mu]);
    if (xs_out) memcpy(&xs_out[(size_t)b * (T + 1) * n], s->xs, sizeof(orc_real) * (size_t)(T + 1) * n);
    if (us_out) memcpy(&us_out[(size_t)b * T * mu], s->us, sizeof(orc_real) * (size_t)T * mu);
    if (cost_out) cost_out[b] = c;
    orc_traj_free(s);
  }
  return 0;
}

int orc_batch_derivatives(const orc_model* m, int B, int T, orc_f64 dt, const orc_real* xs,
                          const orc_real* us, int nthreads, orc_real* fx, orc_real* fu, orc_real* cx,
                          orc_real* cu, orc_real* cxx, orc_real* cxu, orc_real* cuu) {
  const int n = m->nx, mu = m->nu;
  const size_t T1 = (size_t)T + 1;
  const int nt = pick_threads(nthreads);
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static)
  for (int b = 0; b < B; b++) {
    orc_traj* s = orc_traj_alloc(n, mu, T, dt);
    memcpy(s->xs, &xs[(size_t)b * T1 * n], sizeof(orc_real) * T1 * n);
    memcpy(s->us, &us[(size_t)b * T * mu], sizeof(orc_real) * (size_t)T * mu);
    orc_compute_derivatives(m, s);
    memcpy(&fx[(size_t)b * T1 * n * n], s->fx, sizeof(orc_real) * T1 * n * n);
    memcpy(&fu[(size_t)b * T1 * n * mu], s->fu, sizeof(orc_real) * T1 * n * mu);
    memcpy(&cx[(size_t)b * T1 * n], s->cx, sizeof(orc_real) * T1 * n);
    memcpy(&cu[(size_t)b * T1 * mu], s->cu, sizeof(orc_real) * T1 * mu);
    memcpy(&cxx[(size_t)b * T1 * n * n], s->cxx, sizeof(orc_real) * T1 * n * n);
    memcpy(&cxu[(size_t)b * T1 * n * mu], s->cxu, sizeof(orc_real) * T1 * n * mu);
    memcpy(&cuu[(size_t)b * T1 * mu * mu], s->cuu, sizeof(orc_real) * T1 * mu * mu);
    orc_traj_free(s);
  }
  return 0;
}

int orc_batch_backward(const orc_model* m, int B, int T, const orc_real* us, const orc_real* fx,
                       const orc_real* fu, const orc_real* cx, const orc_real* cu, const orc_real* cxx,
                       const orc_real* cxu, const orc_real* cuu, const orc_real* k_prev,
                       const orc_acc* lambda, int nthreads, orc_real* k_out, orc_real* K_out,
                       orc_acc* dV_out, int* diverge_out, orc_real* Vx0_out, orc_real* Vxx0_out) {
  const int n = m->nx, mu = m->nu;
  const size_t T1 = (size_t)T + 1;
  const int nt = pick_threads(nthreads);
  (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static)
  for (int b = 0; b < B; b++) {
    orc_traj* s = orc_traj_alloc(n, mu, T, 0.0);
    memcpy(s->us, &us[(size_t)b * T * mu], sizeof(orc_real) * (size_t)T * mu);
    memcpy(s->fx, &fx[(size_t)b * T1 * n * n], sizeof(orc_real) * T1 * n * n);
    memcpy(s->fu, &fu[(size_t)b * T1 * n * mu], sizeof(orc_real) * T1 * n * mu);
    memcpy(s->cx, &cx[(size_t)b * T1 * n], sizeof(orc_real) * T1 * n);
    memcpy(s->cu, &cu[(size_t)b * T1 * mu], sizeof(orc_real) * T1 * mu);
    memcpy(s->cxx, &cxx[(size_t)b * T1 * n * n], sizeof(orc_real) * T1 * n * n);
    memcpy(s->cxu, &cxu[(size_t)b * T1 * n * mu], sizeof(orc_real) * T1 * n * mu);
    memcpy(s->cuu, &cuu[(size_t)b * T1 * mu * mu], sizeof(orc_real) * T1 * mu * mu);
    if (k_prev) memcpy(s->k, &k_prev[(size_t)b * T * mu], sizeof(orc_real) * (size_t)T * mu);
    s->lambda = lambda ? lambda[b] : 1.0;
    const int div = ORC_BACKWARD_PASS(m, s);
    memcpy(&k_out[(size_t)b * T * mu], s->k, sizeof(orc_real) * (size_t)T * mu);
    memcpy(&K_out[(size_t)b * T * mu * n], s->K, sizeof(orc_real) * (size_t)T * mu * n);
    dV_out[2 * b] = s->dV[0];
    dV_out[2 * b + 1] = s->dV[1];
    diverge_out[b] = div;
    if (Vx0_out) memcpy(&Vx0_out[(size_t)b * n], s->Vx, sizeof(orc_real) * n);
    if (Vxx0_out) memcpy(&Vxx0_out[(size_t)b * n * n], s->Vxx, sizeof(orc_real) * n * n);
    orc_traj_free(s);
  }
  return 0;
}
