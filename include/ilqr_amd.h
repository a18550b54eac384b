/*
 * ilqr_amd.h -- C ABI of libilqr_amd.so, the MI355X-native batched iLQR hot path.
 *
 * This is the drop-in boundary for the hot path of kazuotani14/iLQR ("the reference"):
 * forward rollout, finite-difference derivatives, backward Riccati recursion with the
 * per-timestep box-QP, and the line search / lambda schedule that strings them together,
 * for B independent trajectories at once.  Every entry point names the reference interface
 * (file:line, relative to the reference repository) it replaces.  The reference has no FFI
 * of its own (it is one C++ process); INTEGRATION.md shows the few-line binding that routes
 * class iLQR through this library.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; nothing is thrown across the boundary.
 *     Every function returns 0 on success or a negative ilqr_status_code; ilqr_last_error()
 *     gives the message of the calling thread's last failure.
 *   - T is the number of transitions (= u0.size(), src/ilqr_core.cpp:12); state arrays have
 *     T+1 knot points.  "acrobot T=500" of BASELINE.json is T = 499 here (500 knots).
 *   - Host-side array layouts ("canonical"), all double (also for fp32 handles), row-major over the leading indices:
 *         x0 [B][nx]          u0,us,k [B][T][nu]        xs [B][T+1][nx]
 *         K  [B][T][nu*nx]    each K_t column-major nu x nx  (Eigen MatrixXd default)
 *         fx,cxx [B][T+1][nx*nx]   fu,cxu [B][T+1][nx*nu]   cuu [B][T+1][nu*nu]   (column-major)
 *         cx [B][T+1][nx]     cu [B][T+1][nu]
 *     so that a dump of the reference's std::vector<VectorXd/MatrixXd> members compares
 *     element for element.  Device-side storage is private to the handle.
 *   - All work is enqueued on the handle's HIP stream; getters synchronise that stream.
 *   - A handle is independent of every other handle (the reference's file-static
 *     lambda/dlambda, include/ilqr.h:17-18, are per-trajectory state here, reset to (1,1) by
 *     ilqr_init_traj -- i.e. each fresh solve behaves like a fresh reference process).
 *   - There is no CPU fallback: without a HIP device ilqr_create fails with ILQR_ERR_NO_DEVICE.
 */
#ifndef ILQR_AMD_H_
#define ILQR_AMD_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILQR_AMD_ABI_VERSION 5 /* 2: ilqr_desc.dtype; 3: ILQR_MODEL_USER, ilqr_desc.user_params; 4: ilqr_desc.route, assume_cus (the library reads no environment);
                                  5: results into device memory / in one asynchronous call (ilqr_copy_trajectory_to_device, ilqr_copy_gains_to_device,
                                     ilqr_get_results_async, ilqr_host_register); route bit 128 (ILQR_ROUTE_BACKWARD_LDS, round 1's LDS kernel) retired */

typedef struct ilqr_batch ilqr_batch; /* opaque: owns all device memory of one batch */

enum ilqr_status_code {
  ILQR_OK = 0,
  ILQR_ERR_INVALID = -1,   /* bad argument / size mismatch (the reference asserts: boxqp.cpp:29-33, ilqr_core.cpp:66,80-82) */
  ILQR_ERR_NO_DEVICE = -2, /* no HIP device / runtime failure at create */
  ILQR_ERR_HIP = -3,       /* a HIP call failed; see ilqr_last_error() */
  ILQR_ERR_STATE = -4,     /* call order violated (e.g. warm start before any solve, ilqr_core.cpp:66) */
  ILQR_ERR_UNSUPPORTED = -5
};

/* Device models: compile-time twins of the reference's Model subclasses (include/model.h:6-21).
 * A Model whose dynamics/cost exist only as host virtuals uses ILQR_MODEL_HOST: its rollouts and
 * finite differences stay with the caller (the C++ facade does it), the rest runs here. */
enum ilqr_model_id {
  ILQR_MODEL_ACROBOT = 0,           /* include/acrobot.h            nx=4 nu=1 */
  ILQR_MODEL_DOUBLE_INTEGRATOR = 1, /* include/double_integrator.h  nx=4 nu=2 */
  ILQR_MODEL_LQ = 2,                /* synthetic LQ (BASELINE.json configs[4]): xdot = A x + B u, cost .5(x'Qx + u'Ru),
                                       final .5 x'Qf x, nx<=32 nu<=16; device twin, runs end to end (lq_* of the desc) */
  ILQR_MODEL_HOST = 3,              /* a Model that exists only as host code, nx<=32 nu<=16: the caller evaluates its
                                       rollouts and finite differences (ilqr_set_trajectory, ilqr_set_derivatives,
                                       ilqr_accept_candidates), the backward pass / box-QPs / accept logic run on the
                                       device; rollout and finite-difference entry points return ILQR_ERR_UNSUPPORTED */
  ILQR_MODEL_USER = 4               /* the caller's OWN device twin (nx=4, nu in {1,2}), compiled into a build of this
                                       library from a header that is not part of it: -DILQR_USER_MODEL_HEADER='"file.hpp"'
                                       (ilqr_amd/csrc/models.hpp states the contract, INTEGRATION.md 5 the recipe).  Runs
                                       every kernel of the nx = 4 path, both dtypes.  A build without such a header
                                       answers ILQR_ERR_UNSUPPORTED; ilqr_has_user_model() tells which one is loaded. */
};

/* Arithmetic of the device models (BASELINE.json configs[3] asks for fp32; the reference is fp64 only).
 * ILQR_DTYPE_F32 (acrobot, double integrator): every per-knot quantity -- states, controls, gains, derivative
 * records, the Riccati recursion and the box-QP -- is stored and computed in float; the per-trajectory
 * scalars (cost, dV, gradient norm, lambda) stay double, and the finite differences are taken in double
 * from the float knot (eps = 1e-3 second differences of an O(1e3) cost are pure rounding noise in float)
 * and rounded to float when stored.  The ABI's arrays are double in both modes; the conversion happens on
 * the device when they are packed into / unpacked from the handle's layout. */
enum ilqr_dtype { ILQR_DTYPE_F64 = 0, ILQR_DTYPE_F32 = 1 };

/* where a trajectory's outer loop stands (src/ilqr_core.cpp:103-288) */
enum ilqr_traj_status {
  ILQR_RUNNING = 0,
  ILQR_CONVERGED_GRAD = 1, /* "SUCCESS: gradient norm < tolGrad", ilqr_core.cpp:154-159 */
  ILQR_CONVERGED_COST = 2, /* "SUCCESS: cost change < tolFun",   ilqr_core.cpp:257-262 */
  ILQR_LAMBDA_MAX = 3,     /* "EXIT: lambda > lambdaMax",         ilqr_core.cpp:276-281 */
  ILQR_MAX_ITER = 4        /* maxIter iterations done,            ilqr_core.cpp:103 */
};

enum ilqr_flags {
  /* Bench mode: every iteration runs derivatives + backward + all line-search rollouts for
   * every trajectory and the three termination tests are disabled, so B*T*iters is exactly the
   * work done.  Accept/reject and the lambda schedule stay as in the reference. */
  ILQR_FLAG_FIXED_WORK = 1,
  /* Backward-pass kernel of the stage call ilqr_backward_pass and of the two-kernel route (nx = 4 models; see
   * DESIGN.md): the default is four lanes per trajectory; this flag selects one thread per trajectory (the cross-check).
   * (4 was an experiment kernel of round 2 and is ignored.) */
  ILQR_FLAG_BACKWARD_THREAD_PER_TRAJ = 2,
  /* ilqr_iterate / ilqr_solve normally run the derivative sweep and the backward pass of an
   * iteration in one kernel (nx = 4 models: the sweep uses the SIMDs the backward pass leaves
   * idle).  This flag launches them as two kernels, as the stage calls do.  Same results. */
  ILQR_FLAG_UNFUSED = 8,
  /* Opt-in (the reference has nothing like it): the derivative sweep takes the device
   * model's exact derivatives of the Euler map and of the costs instead of eps = 1e-3 central
   * differences.  Results then differ from the reference's by its finite differences' truncation and
   * rounding error (~1e-6 relative on the records), so parity claims are made WITHOUT this flag. */
  ILQR_FLAG_ANALYTIC_DERIVATIVES = 16,
  /* ilqr_iterate / ilqr_solve normally run whole iterations of a 16-trajectory tile in ONE persistent kernel
   * (batches of up to 16 x #CU trajectories of an nx = 4 model: tiles never wait for each other).  This flag
   * launches the stages of every iteration as kernels of their own instead (fused sweep + backward, then
   * rollouts + accept), as larger batches do.  Same results bit for bit. */
  ILQR_FLAG_STAGED = 32,
  /* Opt-in, OFF by default because it changes results: two things the reference's own
   * comments call for.  (1) The rollout clamps every control into [u_min, u_max] and stores / integrates the
   * clamped one -- src/ilqr_core.cpp:327-329, "This is the right way" (the reference adds K (x - xs) to the
   * box-QP's clamped feed-forward without re-clamping, README.md:9 "control-limited part not working").
   * (2) A failed Cholesky factorisation of Quu on the free subspace ends the box-QP with result -1 and the
   * backward pass reports divergence at that step (lambda is raised) -- src/boxqp.cpp:85-88 never looks at
   * info() and goes on with the partial factor.  Every model: the nx = 4 device models, the generic twins, and host-evaluated
   * models (ILQR_MODEL_HOST: (2) on the device; (1) is applied by whoever rolls out -- the C++ facade's host rollouts clamp
   * under this flag).  The CPU oracle has the same switch. */
  ILQR_FLAG_REFERENCE_FIXES = 64,
  /* Opt-in, OFF by default (third part of SURVEY.md 8f-4): lambda regularises the value Hessian instead of Quu --
   * [Tassa 2012] eq. 10a/10b, Quu_reg = cuu + fu'(Vxx' + lambda I) fu, Qux_reg = cxu' + fu'(Vxx' + lambda I) fx --
   * where the reference adds lambda I to Quu and notes "regularization is different" (src/ilqr_core.cpp:365-367).
   * The value update keeps the unregularised Quu, Qux as in the reference.  Every model: the nx = 4 kernels, the
   * tiled kernels of a small twin, and k_backward_w3 on the generic path (n <= 32, m <= 16; host-evaluated models
   * included: the backward pass is what they run on the device) -- not with ILQR_ROUTE_BACKWARD_W2. */
  ILQR_FLAG_REGULARIZE_VXX = 128
};

/* Solver tunables = the compile-time constants of include/ilqr.h:14-24 (defaults shown). */
typedef struct ilqr_params {
  int max_iter;          /* 100   maxIter */
  double tol_fun;        /* 1e-6  tolFun */
  double tol_grad;       /* 1e-6  tolGrad */
  double lambda_init;    /* 1     lambda */
  double dlambda_init;   /* 1     dlambda */
  double lambda_factor;  /* 1.6   lambdaFactor */
  double lambda_max;     /* 1e11  lambdaMax */
  double lambda_min;     /* 1e-8  lambdaMin */
  double z_min;          /* 0     zMin */
} ilqr_params;

typedef struct ilqr_desc {
  int abi_version; /* ILQR_AMD_ABI_VERSION */
  int model;       /* enum ilqr_model_id */
  int nx, nu;      /* Model::x_dims, Model::u_dims (include/model.h:19-20) */
  int T;           /* transitions */
  int B;           /* trajectories in this batch (this rank's shard) */
  double dt;       /* iLQR::dt, include/ilqr.h:30 */
  int device;      /* HIP device ordinal */
  int flags;       /* enum ilqr_flags */
  int dtype;       /* enum ilqr_dtype; 0 = fp64, the reference's arithmetic */
  const double* u_min; /* [nu] Model::u_min (include/model.h:17); NULL = the model's default */
  const double* u_max; /* [nu] */
  const double* goal;  /* [nx] DoubleIntegrator(goal) (double_integrator.h:14); NULL = (1,.5,0,0); acrobot ignores it */
  const double *lq_A, *lq_B, *lq_Q, *lq_R, *lq_Qf; /* ILQR_MODEL_LQ: row-major [nx][nx],[nx][nu],[nx][nx],[nu][nu],[nx][nx] */
  void* stream;             /* hipStream_t to enqueue on; NULL = the library creates one */
  const ilqr_params* params; /* NULL = reference defaults */
  const double* user_params; /* ILQR_MODEL_USER: [n_user_params] handed to UserModelT::set_params (its constructor's arguments) */
  int n_user_params;
  int route;       /* enum ilqr_route bits; 0 = the library picks the kernels by batch size (what every caller wants) */
  int assume_cus;  /* 0 = the device's CU count; > 0: choose routes as if the device had this many (the tests exercise the
                      batch-size thresholds on small batches) */
} ilqr_desc;

/* Which of several equivalent kernels a handle uses.  Every choice leaves the same bits (tests/test_gpu_fused_sweep.py,
 * scripts/soak.py): these exist for A/B measurements and for those tests.  The library reads NO environment variables. */
enum ilqr_route {
  /* What ILQR_ROUTE_AUTO means for the arithmetic: on the nx = 4 path every route computes every element by the same expression in the
   * same order (bit-identical, tests/test_gpu_fused_sweep.py).  On the GENERIC path (n <= 32, m <= 16: LQ model, larger user twins,
   * host-evaluated models) the default backward kernel k_backward_w3 does NOT follow the reference's operation order: the box-QP's
   * inverse comes from a Newton-Schulz refinement of the previous knot's inverse (the literal Cholesky of src/boxqp.cpp:80-119 is its
   * fallback), the upper Vxx tile is the transpose of the lower one, matrix-vector products are per-lane sums.  Its gains equal the
   * reference-order kernel's and the oracle's to rounding (1e-9 on well-conditioned steps; the 1e-6 per-knot tolerance is what is
   * tested), and where fp64 itself does not determine a pass (an indefinite Quu) only its discrete outcome.  The kernel that keeps
   * the reference's order of operations, bit for bit what round 1's LDS kernel computed, is opt-in: ILQR_ROUTE_BACKWARD_W2. */
  ILQR_ROUTE_AUTO = 0,
  ILQR_ROUTE_TILE_PER_CU = 1,       /* ilqr_iterate: persistent 16-trajectory tiles, one per CU (k_solve_hex for m = 1 without opt-in fixes, else k_solve_tile<..,1>) */
  ILQR_ROUTE_TWO_TILES_PER_CU = 2,  /* ... two per CU (k_solve_tile<..,2>; with ILQR_FLAG_STAGED the one-producer k_sweep_backward) */
  ILQR_ROUTE_WIDE_TILES = 3,        /* ... 64-trajectory wide tiles (k_solve_wide / k_solve_wide2; m <= 2 without opt-in fixes, else as 2) */
  ILQR_ROUTE_WIDE_ONE_PER_CU = 4,   /* wide tiles: one per CU whatever the batch size */
  ILQR_ROUTE_WIDE_TWO_PER_CU = 8,   /* wide tiles (m = 1, k_solve_wide): two per CU whatever the batch size.  The m = 2 wide tiles (k_solve_wide2) always run
                                       one per CU: ilqr_create answers ILQR_ERR_UNSUPPORTED to this bit on an nx = 4, nu = 2 handle */
  ILQR_ROUTE_NO_COMPACTION = 16,    /* ilqr_generate_trajectory without re-packing running trajectories between chunks */
  ILQR_ROUTE_FULL_RECORDS = 32,     /* LQ model, exact derivatives: whole per-knot records instead of one shared copy of the constant blocks */
  ILQR_ROUTE_LQ_THREAD_ROLLOUT = 64,/* LQ model: thread-per-rollout k_rollout_g instead of the matrix-core k_rollout_lq */
  /* (128 was ILQR_ROUTE_BACKWARD_LDS, round 1's LDS kernel k_backward_w: retired in ABI 5 -- ILQR_ROUTE_BACKWARD_W2 gives the same bits;
   *  ilqr_create answers ILQR_ERR_UNSUPPORTED to the bit) */
  ILQR_ROUTE_QUAD_CHAIN = 256,      /* one tile per CU: the 4-lane DPP chain (k_solve_tile<..,1>) also where the matrix-core chains (k_solve_hex: m = 1, no opt-in fixes) would run */
  ILQR_ROUTE_LQ_RECOMMIT = 512,     /* LQ model: no candidate buffers (11 x the nominal trajectory); the accepted rollout is run again to commit it */
  ILQR_ROUTE_LQ_DENSE_FD = 2048,    /* LQ model, finite differences: every perturbed point's quadratic forms evaluated densely on the matrix cores
                                       (k_derivatives_g) instead of by what moved (k_derivatives_lq: Q p = Q x + delta_i Q[:,i] + delta_j Q[:,j]) */
  ILQR_ROUTE_WAVE_PER_TRAJECTORY = 4096, /* a small user twin (even nx <= 8, nu <= 4): the generic wavefront-per-trajectory kernels instead of the tiled
                                            thread-per-trajectory ones it runs in by default (cross-check; fp64 only) */
  ILQR_ROUTE_BACKWARD_W2 = 1024     /* generic path: round 2's register kernel k_backward_w2 (literal Cholesky in every box-QP, per-knot cx / cu records)
                                       instead of k_backward_w3 (matrix-core refinement of the previous knot's inverse; LQ model with exact
                                       derivatives: no record array at all) */
};

const char* ilqr_last_error(void);
int ilqr_abi_version(void);
int ilqr_has_user_model(void); /* 1 if this build carries a user device model (ILQR_MODEL_USER), else 0 */
void ilqr_default_params(ilqr_params* p);

/* iLQR::iLQR(Model*, double), include/ilqr.h:30-44 */
int ilqr_create(const ilqr_desc* desc, ilqr_batch** out);
void ilqr_destroy(ilqr_batch* h);
int ilqr_set_stream(ilqr_batch* h, void* hip_stream);
int ilqr_synchronize(ilqr_batch* h);

/* ---- whole-solve entry points ------------------------------------------------------------ */
/* iLQR::init_traj(x0,u0), src/ilqr_core.cpp:11-56: open-loop rollout, zero all derivative and
 * gain arrays, lambda=dlambda=initial.  cost_out [B] may be NULL. */
int ilqr_init_traj(ilqr_batch* h, const double* x0, const double* u0, double* cost_out);
/* iLQR::generate_trajectory(), src/ilqr_core.cpp:79-302, on the state left by init_traj /
 * previous calls.  Runs until every trajectory has left its loop (max_iter each). */
int ilqr_generate_trajectory(ilqr_batch* h);
/* iLQR::generate_trajectory(x0,u0), src/ilqr_core.cpp:59-62 (= BASELINE.json's "iLQR::solve()") */
int ilqr_solve(ilqr_batch* h, const double* x0, const double* u0);
/* iLQR::generate_trajectory(x0) (warm start), src/ilqr_core.cpp:65-76: re-roll the stored
 * controls with the stored gains from new x0 [B][nx]; lambda/dlambda persist as in the reference. */
int ilqr_warm_start(ilqr_batch* h, const double* x0);
/* n_iters bodies of the outer for-loop (src/ilqr_core.cpp:103-288) for every trajectory that
 * is still running; asynchronous (no host synchronisation inside). */
int ilqr_iterate(ilqr_batch* h, int n_iters);

/* ---- single stages (teacher-forced parity tests, host-model fallback) --------------------- */
/* STEP 1, src/ilqr_core.cpp:115-120 = src/derivatives.cpp:15-144 over t = 0..T, all trajectories */
int ilqr_compute_derivatives(ilqr_batch* h);
/* one iLQR::backward_pass(), src/ilqr_core.cpp:350-401, at the current lambda (no retry).
 * diverge_out [B] (may be NULL) receives its return value. Also refreshes dV. */
int ilqr_backward_pass(ilqr_batch* h, int* diverge_out);
/* STEP 2 incl. the lambda-increase retry loop and the gradient-norm test, ilqr_core.cpp:136-159 */
int ilqr_backward_step(ilqr_batch* h);
/* iLQR::forward_pass(x0,u) for u = us + alpha*k, src/ilqr_core.cpp:305-337 with :188-190:
 * all 11 alphas of include/ilqr.h:24 are rolled out concurrently; cost_out [B][11] may be NULL. */
int ilqr_rollout_candidates(ilqr_batch* h, double* cost_out);
/* STEP 3 + STEP 4, src/ilqr_core.cpp:175-282: candidates, first-accept selection in the
 * reference's serial order, lambda update, termination tests, commit of the accepted one. */
int ilqr_line_search(ilqr_batch* h);
/* Host-evaluated models (ILQR_MODEL_HOST): the caller rolled the 11 candidates out itself
 * (forward_pass needs Model::dynamics/cost, host virtuals) and hands over their costs
 * cost_c [B][11]; the device applies STEP 3/4 of src/ilqr_core.cpp:184-282 to them -- first
 * accepted alpha in the reference's serial order, cost_s, lambda schedule, termination tests,
 * iteration count -- and reports the accepted alpha index per trajectory (accepted [B], -1 =
 * none).  The caller then stores the accepted rollout with ilqr_set_trajectory. */
int ilqr_accept_candidates(ilqr_batch* h, const double* cost_c, int* accepted);
/* For callers that produce the first rollout themselves (host-evaluated models) and pass it in
 * with ilqr_set_trajectory.
 *   warm = 0: the non-rollout part of iLQR::init_traj (src/ilqr_core.cpp:23-48 and the statics of
 *             include/ilqr.h:17-18): zero the derivative and gain arrays, lambda = dlambda =
 *             initial values, every trajectory running with zero iterations.
 *   warm = 1: a new outer loop on the stored solution (warm start, src/ilqr_core.cpp:65-76):
 *             status, iteration count and flgChange restart; lambda, dlambda, gains and
 *             derivatives persist. */
int ilqr_reset_state(ilqr_batch* h, int warm);

/* ---- state exchange (canonical host layouts, see top) -------------------------------------- */
int ilqr_set_trajectory(ilqr_batch* h, const double* x0, const double* xs, const double* us, const double* cost);
int ilqr_set_gains(ilqr_batch* h, const double* k, const double* K);
int ilqr_set_derivatives(ilqr_batch* h, const double* fx, const double* fu, const double* cx,
                         const double* cu, const double* cxx, const double* cxu, const double* cuu);
int ilqr_set_lambda(ilqr_batch* h, const double* lambda, const double* dlambda); /* [B] each, NULL = keep */

int ilqr_get_trajectory(ilqr_batch* h, double* xs, double* us); /* either may be NULL */
int ilqr_get_gains(ilqr_batch* h, double* k, double* K);
/* The derivative records.  WHICH trajectory they describe depends on the route, as the reference's members do on where its loop stands
 * (src/ilqr_core.cpp:115-120 refreshes fx ... cuu at the top of an iteration, :210-213 accepts a new trajectory at its end):
 *   - after a stage call (ilqr_compute_derivatives, ilqr_set_derivatives): the records that call left;
 *   - nx = 4 handles after ilqr_iterate: the persistent kernels keep no records in memory; the getter computes those of the CURRENT
 *     nominal trajectory on demand (zeros after ilqr_init_traj, as ilqr_core.cpp:39-45 leaves them);
 *   - LQ model with exact derivatives on the default route: likewise recomputed for the current nominal trajectory (k_analytic_lq);
 *   - generic handles on the record-keeping routes (finite differences, ILQR_ROUTE_BACKWARD_W2, ILQR_ROUTE_FULL_RECORDS, user twins,
 *     host-evaluated models): what the LAST sweep left, i.e. the records of the trajectory the last iteration STARTED from -- one
 *     accepted step behind the nominal trajectory, exactly as the reference's members are after generate_trajectory().
 * A caller that needs one definite meaning calls ilqr_compute_derivatives first: then every route returns the current trajectory's. */
int ilqr_get_derivatives(ilqr_batch* h, double* fx, double* fu, double* cx, double* cu,
                         double* cxx, double* cxu, double* cuu);
int ilqr_get_cost(ilqr_batch* h, double* cost);                     /* [B] cost_s */
int ilqr_get_lambda(ilqr_batch* h, double* lambda, double* dlambda); /* [B] */
int ilqr_get_dV(ilqr_batch* h, double* dV);                         /* [B][2] */
int ilqr_get_gnorm(ilqr_batch* h, double* gnorm);                   /* [B] */
int ilqr_get_status(ilqr_batch* h, int* status, int* iters, int* alpha_idx); /* [B] each, any NULL */
/* one alpha's rollout of the LAST line search (ilqr_rollout_candidates / ilqr_line_search / ilqr_iterate).  The candidate
 * buffers are scratch of the line search (the reference keeps x_new/u_new as locals of its loop, src/ilqr_core.cpp:184-186):
 * ILQR_ERR_STATE when none has been rolled out yet or when ilqr_generate_trajectory re-packed running trajectories
 * (compaction), which leaves the buffers behind. */
int ilqr_get_candidate(ilqr_batch* h, int alpha_idx, double* xs, double* us);
int ilqr_count_running(ilqr_batch* h, int* n_running);
/* device-to-device copy of the per-trajectory costs [B] into caller-owned device memory
 * (the payload of the multi-GPU gather).  Enqueued on the handle's stream:
 * ilqr_synchronize before handing the buffer to work on another stream (a collective). */
int ilqr_copy_cost_to_device(ilqr_batch* h, void* dst_device);
/* ---- results without a host round trip per array (ABI 5) ------------------------------------------------------------------------
 * The reference hands its caller xs / us / K / k as members (include/ilqr.h:48-56); a batched caller of the getters above pays one
 * unpack kernel + one device-to-host copy + one stream synchronisation per array, and for an 11 ms job (B = 4096, 20 iterations)
 * 164 MB of results are 3-10 ms of that.  Two ways around it:
 * (1) Callers that stay on the GPU (MPC: the next warm start, src/ilqr_core.cpp:65-76, a learned policy's loss, ...): the canonical
 *     layouts of the top of this file as double, written into CALLER-OWNED DEVICE memory on this handle's device.  Enqueued on the handle's
 *     stream, not waited for (ilqr_synchronize, or stream order, before another stream reads them).  Any pointer may be NULL. */
int ilqr_copy_trajectory_to_device(ilqr_batch* h, void* xs_device, void* us_device);
int ilqr_copy_gains_to_device(ilqr_batch* h, void* k_device, void* K_device);
/* (2) Host callers: every requested array (NULL = skip) in ONE call -- the unpack kernels and the device-to-host copies follow each
 *     other on the handle's stream with NO synchronisation; the arrays are valid after ilqr_synchronize.  With page-locked destinations
 *     (ilqr_host_register once per buffer a caller reuses, or hipHostMalloc) the copies are DMA at PCIe rate and the call returns at
 *     once; with pageable memory the runtime stages them and the call may block, the result is the same. */
int ilqr_get_results_async(ilqr_batch* h, double* xs, double* us, double* k, double* K, double* cost);
/* hipHostRegister / hipHostUnregister for callers that do not link the HIP runtime themselves */
int ilqr_host_register(void* ptr, size_t bytes);
int ilqr_host_unregister(void* ptr);

/* ---- several shards of one batch, one process ----------------------------------
 * The path partitions by trajectory: shard i = a handle of its own (any device, ilqr_desc.device) holding the contiguous block
 * [offset_i, offset_i + B_i) of the global batch; there is no data-path collective.  A group names the shards of one job and
 * performs its ONE exchange, the gather of the per-trajectory costs in global order:
 *   - shards on n > 1 DISTINCT devices: one RCCL communicator per device (ncclCommInitAll, one process) and one
 *     ncclAllGather of the padded shards over xGMI -- every device ends up with the whole vector (librccl.so is loaded when the
 *     first such group is created, not by ilqr_create);
 *   - shards that share a device (N logical shards on one GPU): device-to-host copies, shard by shard.
 * The reference has no counterpart (one trajectory per process, src/run_ilqr.cpp:27-59); a caller of it who solves many
 * problems loops over them -- this is that loop's gather. */
typedef struct ilqr_group ilqr_group;
/* flags: 1 = use RCCL even for a single device / a single shard (tests) */
int ilqr_group_create(ilqr_batch* const* shards, int n_shards, int flags, ilqr_group** out);
void ilqr_group_destroy(ilqr_group* g);
/* cost_out: host [sum of the shards' B], global order.  Synchronises every shard. */
int ilqr_group_gather_costs(ilqr_group* g, double* cost_out);
/* 1 if the group's gather runs over RCCL, 0 if it copies; *n_ranks = communicators in use (0 without RCCL) */
int ilqr_group_uses_rccl(ilqr_group* g, int* n_ranks);

/* ---- measurement --------------------------------------------------------------------------- */
enum ilqr_stage { ILQR_STAGE_DERIVATIVES = 0, ILQR_STAGE_BACKWARD = 1, ILQR_STAGE_ROLLOUT = 2,
                  ILQR_STAGE_ACCEPT = 3, ILQR_STAGE_SOLVE = 4, ILQR_NUM_STAGES = 5 };
/* HIP-event timing of every kernel launch of a stage, accumulated on the handle's stream.  ILQR_STAGE_SOLVE is the
 * persistent per-tile kernel of ilqr_iterate (one launch = all the iterations of the call); while it runs, the
 * BACKWARD and ROLLOUT slots receive the kernel's own per-phase clocks (mean over tiles, one "launch" per
 * iteration) so that the split stays visible. */
int ilqr_profile_enable(ilqr_batch* h, int enable);
int ilqr_profile_reset(ilqr_batch* h);
/* total milliseconds and launch count per stage since the last reset (synchronises) */
int ilqr_profile_read(ilqr_batch* h, double ms_out[ILQR_NUM_STAGES], int launches_out[ILQR_NUM_STAGES]);
/* The clock the CUs ran at during the persistent kernel's launches since the last reset, in MHz: shader cycles (s_memtime)
 * over constant-rate ticks, summed over tiles.  MI355X lowers its clock with the number of busy SIMDs, so issue-rate
 * figures derived from a launch duration need the clock of THAT launch (bench.py: roofline_issue). */
int ilqr_profile_shader_clock(ilqr_batch* h, double* mhz_out);
/* name of the kernel a stage launches (as rocprofv3 reports it), for bench.py's roofline line */
const char* ilqr_stage_kernel_name(ilqr_batch* h, int stage);

#ifdef __cplusplus
}
#endif
#endif /* ILQR_AMD_H_ */
