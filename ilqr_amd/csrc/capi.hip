// capi.hip -- the entry points of the C ABI of include/ilqr_amd.h (creation, whole solves, stage calls, state exchange) on top of the
// HIP kernels.  The handle and its helpers: handle.hpp; kernel launchers and route logic: launch.hpp; shard groups (RCCL):
// group.hpp; measurement: profile.hpp -- one translation unit (every kernel template is instantiated where it is launched).
// No CPU compute path exists: every entry point either launches kernels or moves bytes.
#include "../../include/ilqr_amd.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "backward_wave.hpp"
#include "backward_wave2.hpp"
#include "backward_wave3.hpp"
#include "generic.hpp"
#include "kernels_wide.hpp"
#include "kernels_wide2.hpp"
#include "kernels.hpp"

using namespace ilqr;

#include "handle.hpp"
#include "launch.hpp"

// ------------------------------------------------------------------------------------------
// public API
// ------------------------------------------------------------------------------------------
extern "C" {

const char* ilqr_last_error(void) { return g_err; }
int ilqr_abi_version(void) { return ILQR_AMD_ABI_VERSION; }
int ilqr_has_user_model(void) {
#ifdef ILQR_HAVE_USER_MODEL
  return 1;
#else
  return 0;
#endif
}

void ilqr_default_params(ilqr_params* p) {  // include/ilqr.h:14-24
  p->max_iter = 100;
  p->tol_fun = 1e-6;
  p->tol_grad = 1e-6;
  p->lambda_init = 1;
  p->dlambda_init = 1;
  p->lambda_factor = 1.6;
  p->lambda_max = 1e11;
  p->lambda_min = 1e-8;
  p->z_min = 0;
}

void ilqr_destroy(ilqr_batch* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
#ifdef ILQR_W2_TIMING
  if (h->aos) {  // experiment build: what the first wavefront of every k_backward_w2 launch spent where (backward_wave.hpp)
    long long st[8], qp[8], cnt[4];
    if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_w2_cycles), sizeof(st)) == hipSuccess && hipMemcpyFromSymbol(qp, HIP_SYMBOL(g_q_cycles), sizeof(qp)) == hipSuccess &&
        hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_q_counts), sizeof(cnt)) == hipSuccess && cnt[0] > 0) {
      const double q = (double)cnt[0];  // (literal box-QPs; on the k_backward_w3 route the step sections are per LITERAL QP too: scale by the counts below)
      fprintf(stderr, "[k_backward_w2, wavefront 0] shader cycles per step: load+Qx/Qu %.0f  Vxx'fx,Vxx'fu %.0f  Qxx/Qux/Quu %.0f  box-QP %.0f  K %.0f  dV+T1+Vx %.0f  Vn+symmetrise+stores %.0f  (loop top %.0f)\n",
              st[0] / q, st[1] / q, st[2] / q, st[3] / q, st[4] / q, st[5] / q, st[6] / q, st[7] / q);
      fprintf(stderr, "[box-QP] per QP: %.2f iterations, %.2f factorisations, %.2f Armijo trips beyond the first; cycles: gradient+clamp set %.0f  Cholesky %.0f  inverse+R^-1R^-T %.0f  direction %.0f  line search %.0f  rest %.0f\n",
              cnt[1] / q, cnt[2] / q, cnt[3] / q, qp[0] / q, qp[1] / q, qp[2] / q, qp[3] / q, qp[4] / q, qp[5] / q);
    }
    long long w3[4];
    if (hipMemcpyFromSymbol(w3, HIP_SYMBOL(g_w3_counts), sizeof(w3)) == hipSuccess && w3[0] + w3[1] > 0)
      fprintf(stderr, "[k_backward_w3, wavefront 0] box-QPs on the matrix-core path %lld, handed to the literal path %lld, Newton-Schulz iterations per refinement %.2f\n",
              w3[0], w3[1], (double)w3[2] / (double)(w3[0] > 0 ? w3[0] : 1));
  }
#endif
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->staging) (void)hipFree(h->staging);
  if (h->d_perm) (void)hipFree(h->d_perm);
  if (h->perm_scratch) (void)hipFree(h->perm_scratch);
  (void)timers_drain(h);  // (every event back into the pool, each once)
  for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

static int create_impl(const ilqr_desc* d, ilqr_batch* h) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(ILQR_ERR_NO_DEVICE, "no HIP device visible: libilqr_amd has no CPU path");
  if (d->device < 0 || d->device >= ndev) return fail(ILQR_ERR_NO_DEVICE, "device %d out of range (%d visible)", d->device, ndev);
  HIPCHK(hipSetDevice(d->device));
  HIPCHK(hipDeviceGetAttribute(&h->num_cus, hipDeviceAttributeMultiprocessorCount, d->device));
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, d->device) == hipSuccess && khz > 0) h->wall_clock_khz = khz;
  }
  // route choices come with the descriptor (ilqr_desc.route, include/ilqr_amd.h): the library reads no environment
  h->route.staged = false;
  h->route.unfused = false;
  h->route.backward_w2 = (d->route & ILQR_ROUTE_BACKWARD_W2) != 0;
  h->route.lq_dense_fd = (d->route & ILQR_ROUTE_LQ_DENSE_FD) != 0;
  h->route.lq_thread_rollout = (d->route & ILQR_ROUTE_LQ_THREAD_ROLLOUT) != 0;
  h->route.full_records = (d->route & ILQR_ROUTE_FULL_RECORDS) != 0;
  h->route.no_compaction = (d->route & ILQR_ROUTE_NO_COMPACTION) != 0;
  h->route.quad_chain = (d->route & ILQR_ROUTE_QUAD_CHAIN) != 0;
  h->route.fused = d->route & 3;
  h->route.wide_occ = (d->route & ILQR_ROUTE_WIDE_ONE_PER_CU) ? 1 : (d->route & ILQR_ROUTE_WIDE_TWO_PER_CU) ? 2 : 0;
  if (d->assume_cus > 0) h->num_cus = d->assume_cus;
  h->device = d->device;
  if (d->stream) {
    h->stream = (hipStream_t)d->stream;
  } else {
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  h->model = d->model;
  h->dtype = d->dtype;
  if (d->dtype != ILQR_DTYPE_F64 && d->dtype != ILQR_DTYPE_F32) return fail(ILQR_ERR_INVALID, "dtype %d: ILQR_DTYPE_F64 or ILQR_DTYPE_F32", d->dtype);
  if (d->dtype == ILQR_DTYPE_F32 && d->model != ILQR_MODEL_ACROBOT && d->model != ILQR_MODEL_DOUBLE_INTEGRATOR && d->model != ILQR_MODEL_USER)
    return fail(ILQR_ERR_UNSUPPORTED, "fp32 is available for the nx = 4 device models (acrobot, double integrator); the generic nx <= 32 path is fp64");
  h->nx = d->nx;
  h->nu = d->nu;
  h->T = d->T;
  h->B = d->B;
  h->dt = d->dt;
  h->flags = d->flags;
  h->Bp = ((d->B + 63) / 64) * 64;
  h->ntiles = h->Bp / TW;
  if (d->params)
    h->params = *d->params;
  else
    ilqr_default_params(&h->params);

  // model parameters (the constructor bodies of acrobot.h:14-40, double_integrator.h:14-27)
  if (d->model == ILQR_MODEL_ACROBOT) {
    REQUIRE(d->nx == 4 && d->nu == 1, "acrobot is nx=4 nu=1 (include/acrobot.h:27-28), got %d/%d", d->nx, d->nu);
    AcrobotModel& m = h->acrobot;
    m.goal[0] = 3.1415;
    m.goal[1] = m.goal[2] = m.goal[3] = 0;
    m.u_min[0] = d->u_min ? d->u_min[0] : -5.0;
    m.u_max[0] = d->u_max ? d->u_max[0] : 5.0;
    if (h->dtype == ILQR_DTYPE_F32) {  // the float model, and its parameters' float values in the double twin
      AcrobotModelT<float>& f = h->acrobot_f;
      for (int i = 0; i < 4; i++) m.goal[i] = (double)(f.goal[i] = (float)m.goal[i]);
      m.u_min[0] = (double)(f.u_min[0] = (float)m.u_min[0]);
      m.u_max[0] = (double)(f.u_max[0] = (float)m.u_max[0]);
    }
  } else if (d->model == ILQR_MODEL_DOUBLE_INTEGRATOR) {
    REQUIRE(d->nx == 4 && d->nu == 2, "double integrator is nx=4 nu=2 (include/double_integrator.h:16-17), got %d/%d", d->nx, d->nu);
    DoubleIntegratorModel& m = h->dint;
    const double g0[4] = {1.0, 0.5, 0.0, 0.0};
    for (int i = 0; i < 4; i++) m.goal[i] = d->goal ? d->goal[i] : g0[i];
    for (int j = 0; j < 2; j++) {
      m.u_min[j] = d->u_min ? d->u_min[j] : -0.5;
      m.u_max[j] = d->u_max ? d->u_max[j] : 0.5;
    }
    if (h->dtype == ILQR_DTYPE_F32) {
      DoubleIntegratorModelT<float>& f = h->dint_f;
      for (int i = 0; i < 4; i++) m.goal[i] = (double)(f.goal[i] = (float)m.goal[i]);
      for (int j = 0; j < 2; j++) {
        m.u_min[j] = (double)(f.u_min[j] = (float)m.u_min[j]);
        m.u_max[j] = (double)(f.u_max[j] = (float)m.u_max[j]);
      }
    }
#ifdef ILQR_HAVE_USER_MODEL
  } else if (d->model == ILQR_MODEL_USER) {
    using UM = UserModelT<double>;
    REQUIRE(d->nx == UM::NX && d->nu == UM::NU, "this build's user model is nx=%d nu=%d, got %d/%d", UM::NX, UM::NU, d->nx, d->nu);
    REQUIRE(d->u_min && d->u_max, "ILQR_MODEL_USER needs u_min/u_max (Model::u_min/u_max, include/model.h:17)");
    REQUIRE(!(d->flags & ILQR_FLAG_ANALYTIC_DERIVATIVES) || has_analytic_record<UM>::value, "this user model has no analytic_record()");
    // not a tiled shape -- or a small one asked to take the generic route: the generic kernels (fp64), trajectory-contiguous layout like the LQ model's
    if (!kUserTiled || (kUserSmall && (d->route & ILQR_ROUTE_WAVE_PER_TRAJECTORY))) {
      REQUIRE(d->dtype == ILQR_DTYPE_F64, "the generic nx <= 32 path is fp64");
      h->aos = true;
    }
    REQUIRE(d->n_user_params >= 0 && (d->n_user_params == 0 || d->user_params), "ILQR_MODEL_USER: n_user_params = %d with user_params = %p", d->n_user_params, (const void*)d->user_params);
    h->user_f.set_params(d->user_params, d->n_user_params);
    if (h->dtype == ILQR_DTYPE_F32) {  // the twin the finite differences are taken in: built from the parameters' FLOAT values, like the shipped models'
      std::vector<double> p32(d->user_params, d->user_params + (d->user_params ? d->n_user_params : 0));
      for (double& q : p32) q = (double)(float)q;
      h->user.set_params(p32.data(), (int)p32.size());
    } else {
      h->user.set_params(d->user_params, d->n_user_params);
    }
    for (int j = 0; j < UM::NU; j++) {
      h->user_f.u_min[j] = (float)d->u_min[j];
      h->user_f.u_max[j] = (float)d->u_max[j];
      // fp32 handle: the double twin carries the float model's limits (its other parameters are whatever set_params made of them)
      h->user.u_min[j] = (h->dtype == ILQR_DTYPE_F32) ? (double)h->user_f.u_min[j] : d->u_min[j];
      h->user.u_max[j] = (h->dtype == ILQR_DTYPE_F32) ? (double)h->user_f.u_max[j] : d->u_max[j];
    }
    static_cast<UM&>(h->user_g) = h->user;  // the generic kernels' copy: parameters AND limits (a model's cost may read its own u_min / u_max)
#endif
  } else if (d->model == ILQR_MODEL_HOST || d->model == ILQR_MODEL_LQ) {
    // Generic dimensions: trajectory-contiguous layout, one wavefront per trajectory in the backward
    // pass.  Host-evaluated models receive their derivatives through ilqr_set_derivatives; the LQ
    // model has a device twin (generic.hpp) and runs end to end.
    REQUIRE(d->nx <= WN && d->nu <= WM, "generic kernels: nx <= %d, nu <= %d", WN, WM);
    REQUIRE(d->u_min && d->u_max, "generic handles need u_min/u_max (Model::u_min/u_max, include/model.h:17)");
    if (d->model == ILQR_MODEL_LQ)
      REQUIRE(d->lq_A && d->lq_B && d->lq_Q && d->lq_R && d->lq_Qf, "ILQR_MODEL_LQ needs lq_A, lq_B, lq_Q, lq_R, lq_Qf");
    h->aos = true;
  } else {
    return fail(ILQR_ERR_UNSUPPORTED, "model id %d is not available in this build", d->model);
  }

  const size_t nt = h->ntiles, T = h->T, T1 = h->T + 1, nx = h->nx, nu = h->nu, REC = rec_of(h), Bp = h->Bp;
  BatchView& v = h->v;
  v.B = h->B;
  v.Bp = h->Bp;
  v.ntiles = h->ntiles;
  v.T = h->T;
  v.dt = h->dt;
  v.analytic = (h->flags & ILQR_FLAG_ANALYTIC_DERIVATIVES) ? 1 : 0;
  int rc = 0;
  if (h->aos) {
    const size_t Bn = h->B;
    v.nch = 0;
    rc |= dev_alloc(h, &v.x0, Bn * nx);
    rc |= dev_alloc(h, &v.xs, Bn * T1 * nx);
    rc |= dev_alloc(h, &v.us, Bn * T * nu);
    rc |= dev_alloc(h, &v.kff, Bn * T * nu);
    rc |= dev_alloc(h, &v.Kfb, Bn * T * nu * nx);
    v.D = nullptr;  // on first use (ensure_records): the fused LQ route never needs it
    rc |= dev_alloc(h, &h->const_rec, 2 * REC);
    rc |= dev_alloc(h, &h->d_umin, nu);
    rc |= dev_alloc(h, &h->d_umax, nu);
    v.cand_u = nullptr;
    v.cand_x = nullptr;
    if (d->model == ILQR_MODEL_LQ && !h->route.lq_thread_rollout && !(d->route & ILQR_ROUTE_LQ_RECOMMIT)) {
      // the eleven rollouts of the matrix-core search, whole ([b][alpha][t][row]): the commit is then a copy, not a twelfth rollout
      // (11 x the nominal trajectory, ~7 GB at configs[4]: if the device cannot spare them the handle works without -- the ILQR_ROUTE_LQ_RECOMMIT route)
      void *cx = nullptr, *cu = nullptr;
      if (hipMalloc(&cx, Bn * NALPHA * T1 * nx * sizeof(double)) == hipSuccess && hipMalloc(&cu, Bn * NALPHA * T * nu * sizeof(double)) == hipSuccess) {
        h->allocs.push_back(cx);
        h->allocs.push_back(cu);
        v.cand_x = (double*)cx;
        v.cand_u = (double*)cu;
      } else {
        if (cx) (void)hipFree(cx);
        (void)hipGetLastError();  // (clears the allocation error)
      }
    }
    rc |= dev_alloc(h, &v.cost_c, (size_t)NALPHA * Bp);  // the 11 candidate costs (device or caller-evaluated)
    if (d->model == ILQR_MODEL_LQ) {
      // zero-padded copies of the model matrices at the kernels' maximum dimensions
      double* pad = nullptr;
      const size_t nA = GN * GN, nB = GN * GM, nR = GM * GM, tot = 3 * nA + nB + nR;
      rc |= dev_alloc(h, &pad, tot);
      if (!rc) {
        std::vector<double> hp(tot, 0.0);
        double *pA = hp.data(), *pB = pA + nA, *pQ = pB + nB, *pR = pQ + nA, *pQf = pR + nR;
        for (size_t i = 0; i < nx; i++) {
          for (size_t j = 0; j < nx; j++) {
            pA[i * GN + j] = d->lq_A[i * nx + j];
            pQ[i * GN + j] = d->lq_Q[i * nx + j];
            pQf[i * GN + j] = d->lq_Qf[i * nx + j];
          }
          for (size_t j = 0; j < nu; j++) pB[i * GM + j] = d->lq_B[i * nu + j];
        }
        for (size_t i = 0; i < nu; i++)
          for (size_t j = 0; j < nu; j++) pR[i * GM + j] = d->lq_R[i * nu + j];
        // on the handle's stream, behind dev_alloc's zero fill of the same buffer (a copy on the null
        // stream could be overtaken by it: the stream is non-blocking); hp must outlive the copy
        if (hipMemcpyAsync(pad, hp.data(), tot * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess)
          rc = 1;
        h->lq.nx = (int)nx;
        h->lq.nu = (int)nu;
        h->lq.A = pad;
        h->lq.Bm = pad + nA;
        h->lq.Q = pad + nA + nB;
        h->lq.R = pad + 2 * nA + nB;
        h->lq.Qf = pad + 2 * nA + nB + nR;
        h->lq.umin = h->d_umin;
        h->lq.umax = h->d_umax;
      }
    }
    if (!rc) {
      if (hipMemcpyAsync(h->d_umin, d->u_min, nu * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = 1;
      if (hipMemcpyAsync(h->d_umax, d->u_max, nu * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess) rc = 1;
    }
  } else {
  rc |= dev_alloc_real(h, &v.x0, nt * nx * TW);
  // (+ kRolloutFetchSlack rows: the shared-row rollouts prefetch up to that many steps past T - 1 without clamping the row index -- rollout.hpp --
  //  which for the last tile is past the array's end; what those loads return is never used)
  const size_t slack = (size_t)kRolloutFetchSlack * nu * nx * TW;
  rc |= dev_alloc_real(h, &v.xs, nt * T1 * nx * TW + slack);
  rc |= dev_alloc_real(h, &v.us, nt * T * nu * TW + slack);
  rc |= dev_alloc_real(h, &v.kff, nt * T * nu * TW + slack);
  rc |= dev_alloc_real(h, &v.Kfb, nt * T * nu * nx * TW + slack);
  v.D = nullptr;  // on first use (ensure_records)
  v.nch = h->T / CT + 1;
  // (one plane more than there are alphas: where the rollout lanes without a rollout of their own put their stores, rollout.hpp)
  // (the matrix-core kernel keeps its candidates in groups of CG controls, a plane row padded to whole chunks past T - 1: rollout.hpp, cand_g_u)
  rc |= dev_alloc_real(h, &v.cand_u, (size_t)(NALPHA + 1) * nt * (size_t)(cand_groups(T) * CG) * nu * TW);
  rc |= dev_alloc_real(h, &v.cand_x, (size_t)(NALPHA + 1) * nt * v.nch * nx * TW);
  rc |= dev_alloc(h, &v.cost_c, (size_t)NALPHA * Bp);
  }
  rc |= dev_alloc(h, &v.cost, Bp);
  rc |= dev_alloc(h, &v.lambda, Bp);
  rc |= dev_alloc(h, &v.dlambda, Bp);
  rc |= dev_alloc(h, &v.dV, 2 * Bp);
  rc |= dev_alloc(h, &v.gnorm, Bp);
  rc |= dev_alloc(h, &v.status, Bp);
  rc |= dev_alloc(h, &v.iters, Bp);
  rc |= dev_alloc(h, &v.flg_change, Bp);
  rc |= dev_alloc(h, &v.alpha_idx, Bp);
  rc |= dev_alloc(h, &v.diverge, Bp);
  rc |= dev_alloc(h, &v.backpass_done, Bp);
  rc |= dev_alloc(h, &v.n_running, 1);
  rc |= dev_alloc(h, &h->commit_idx, Bp);
  rc |= dev_alloc(h, &h->phase_ticks, 5 * (size_t)h->ntiles);
  if (!rc && hipMemsetAsync(h->commit_idx, 0xFF, Bp * sizeof(int), h->stream) != hipSuccess) rc = 1;
  if (rc) return ILQR_ERR_HIP;
  sync_float_view(h);

  h->sp.max_iter = h->params.max_iter;
  h->sp.tol_fun = h->params.tol_fun;
  h->sp.tol_grad = h->params.tol_grad;
  h->sp.lambda_factor = h->params.lambda_factor;
  h->sp.lambda_max = h->params.lambda_max;
  h->sp.lambda_min = h->params.lambda_min;
  h->sp.z_min = h->params.z_min;
  h->sp.fixed_work = (h->flags & ILQR_FLAG_FIXED_WORK) ? 1 : 0;
  h->sp.fixes = ((h->flags & ILQR_FLAG_REFERENCE_FIXES) ? 3 : 0) | ((h->flags & ILQR_FLAG_REGULARIZE_VXX) ? 4 : 0);
  // generic handles: ILQR_FLAG_REFERENCE_FIXES -- models with a device twin: their rollouts clamp, their box-QP reports a failed factorisation; the
  // host-evaluated route: the box-QP likewise, the rollouts belong to the caller (the facade clamps).  ILQR_FLAG_REGULARIZE_VXX is the backward pass's alone (k_backward_w3<.., REGV>), on any model
  if ((h->sp.fixes & 4) && h->aos && h->route.backward_w2)
    return fail(ILQR_ERR_UNSUPPORTED, "ILQR_FLAG_REGULARIZE_VXX on the generic path is implemented in k_backward_w3: drop ILQR_ROUTE_BACKWARD_W2");
  // (a host-evaluated model under ILQR_FLAG_REFERENCE_FIXES: part (2), the failed factorisation that ends the box-QP, is the device's -- k_backward_w3 /
  //  k_backward_w2 honour sp.fixes & 2 --; part (1), the clamped rollout, belongs to whoever rolls out: the C++ facade's host_forward does it)

  hipLaunchKernelGGL(k_reset_state<double>, dim3((h->Bp + 255) / 256), dim3(256), 0, h->stream, h->v, h->params.lambda_init,
                     h->params.dlambda_init);
  HIPCHK(hipGetLastError());
  h->lq_fused = h->model == ILQR_MODEL_LQ && v.analytic && !h->route.full_records && !h->route.backward_w2 && !(h->sp.fixes & 4);
  if (h->lq_fused) {  // both constant records, once (what = 3)
    hipLaunchKernelGGL(k_analytic_lq, dim3(1), dim3(64), 0, h->stream, h->v, h->lq, 1, 3, h->const_rec, kAnalyticChunk);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

int ilqr_create(const ilqr_desc* d, ilqr_batch** out) {
  if (!d || !out) return fail(ILQR_ERR_INVALID, "null argument");
  *out = nullptr;
  REQUIRE(d->abi_version == ILQR_AMD_ABI_VERSION, "ABI version %d, library is %d", d->abi_version, ILQR_AMD_ABI_VERSION);
  REQUIRE(d->B >= 1 && d->T >= 1 && d->nx >= 1 && d->nu >= 1, "B, T, nx, nu must be positive");
  REQUIRE(d->nx <= MAXN && d->nu <= MAXM, "nx <= %d and nu <= %d", MAXN, MAXM);
  REQUIRE(d->dt > 0, "dt must be positive");
  if (d->route & 128)  // (ILQR_ROUTE_BACKWARD_LDS of ABI <= 4)
    return fail(ILQR_ERR_UNSUPPORTED, "route bit 128 (round 1's LDS kernel k_backward_w) was retired in ABI 5: ILQR_ROUTE_BACKWARD_W2 gives the same bits");
  if ((d->route & ILQR_ROUTE_WIDE_TWO_PER_CU) && d->nu == 2 && d->nx == 4)
    return fail(ILQR_ERR_UNSUPPORTED, "ILQR_ROUTE_WIDE_TWO_PER_CU: the m = 2 wide tiles (k_solve_wide2) run one tile per CU (four wavefronts at <= 512 registers); the bit applies to k_solve_wide (m = 1)");
  ilqr_batch* h = new ilqr_batch();
  const int rc = create_impl(d, h);
  if (rc) {
    char keep[sizeof(g_err)];
    memcpy(keep, g_err, sizeof(keep));
    if (rc != ILQR_ERR_NO_DEVICE) ilqr_destroy(h); else delete h;
    memcpy(g_err, keep, sizeof(keep));
    return rc;
  }
  *out = h;
  return 0;
}

int ilqr_set_stream(ilqr_batch* h, void* s) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->own_stream) {
    HIPCHK(hipStreamDestroy(h->stream));
    h->own_stream = false;
  }
  if (s) {
    h->stream = (hipStream_t)s;
  } else {
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  return 0;
}

int ilqr_synchronize(ilqr_batch* h) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// ---- whole-solve entry points --------------------------------------------------------------
int ilqr_init_traj(ilqr_batch* h, const double* x0, const double* u0, double* cost_out) {
  if (!h || !x0 || !u0) return fail(ILQR_ERR_INVALID, "null argument");
  if (host_model(h)) return no_device_model();
  HIPCHK(hipSetDevice(h->device));
  if (int rc = upload(h, x0, h->v.x0, 1, h->nx)) return rc;
  if (int rc = upload(h, u0, h->v.us, h->T, h->nu)) return rc;  // us = u_0, ilqr_core.cpp:17
  // ilqr_core.cpp:23-48: zero derivative/gain arrays; statics lambda/dlambda as for a fresh process
  const size_t T = h->T, T1 = h->T + 1;
  if (h->v.D) HIPCHK(hipMemsetAsync(h->v.D, 0, dev_elems(h, T1, rec_of(h)) * elem_size(h), h->stream));
  h->recs = ilqr_batch::REC_ZERO;
  HIPCHK(hipMemsetAsync(h->v.kff, 0, dev_elems(h, T, h->nu) * elem_size(h), h->stream));
  HIPCHK(hipMemsetAsync(h->v.Kfb, 0, dev_elems(h, T, h->nu * h->nx) * elem_size(h), h->stream));
  if (int rc = forget_pending(h)) return rc;
  hipLaunchKernelGGL(k_reset_state<double>, dim3((h->Bp + 255) / 256), dim3(256), 0, h->stream, h->v, h->params.lambda_init,
                     h->params.dlambda_init);
  HIPCHK(hipGetLastError());
  // ilqr_core.cpp:20: open-loop rollout (K is empty); writes xs, us, cost in place
  AlphaSet al = line_search_alphas();
  if (int rc = launch_rollout(h, false, false, al, 1, h->v.cost, 0)) return rc;
  h->initialised = true;
  if (cost_out) return scalars_to_host(h, h->v.cost, cost_out);
  return 0;
}

int ilqr_iterate(ilqr_batch* h, int n_iters) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  if (host_model(h)) return no_device_model();
  if (!h->initialised) return fail(ILQR_ERR_STATE, "ilqr_iterate before ilqr_init_traj/ilqr_set_trajectory");
  HIPCHK(hipSetDevice(h->device));
  struct Chain {  // stage timers share their boundary events for the duration of this call
    ilqr_batch* h;
    explicit Chain(ilqr_batch* hh) : h(hh) { h->chain_timers = true; h->chain_event = nullptr; }
    ~Chain() { h->chain_timers = false; h->chain_event = nullptr; }
  } chain(h);
  if (n_iters > 0) h->cands_valid = !(h->active_tiles > 0 && h->active_tiles < h->ntiles);  // (a compacted chunk rolls out the leading tiles only)
  if (use_persistent(h) && n_iters > 0) {
    if (int rc = launch_solve_tiles(h, n_iters)) return rc;
    return flush_commit(h);
  }
  for (int it = 0; it < n_iters; it++) {
    if (use_fused_sweep(h)) {
      if (int rc = launch_sweep_backward(h, 1, h->sp.fixed_work)) return rc;  // STEP 1 + STEP 2
    } else {
      if (int rc = launch_derivatives(h, h->sp.fixed_work)) return rc;  // STEP 1
      if (int rc = launch_backward(h, 1)) return rc;                    // STEP 2
    }
    if (!h->aos) {  // STEP 3 + STEP 3/4 in one launch: the rollout block of a tile also accepts for it
      if (int rc = launch_rollout(h, true, true, line_search_alphas(), NALPHA, h->v.cost_c, 1, true)) return rc;
      h->commit_pending = true;
    } else if (lq_search_accepts(h)) {  // STEP 3 + STEP 3/4 in one launch (k_rollout_lq<RG_SEARCH, true>); the commit is a copy (k_commit_lq)
      if (int rc = launch_rollout(h, true, true, line_search_alphas(), NALPHA, h->v.cost_c, 1, true)) return rc;
      h->cands_valid = true;
      h->commit_pending = true;
    } else {
      if (int rc = do_rollout_candidates(h, 1)) return rc;              // STEP 3
      if (int rc = launch_accept(h)) return rc;                         // STEP 3/4
    }
  }
  if (!h->aos && n_iters > 0) h->recs = ilqr_batch::REC_STALE;  // (D lags the nominal after an accept, on every route)
  return flush_commit(h);  // the last iteration's accepted trajectories
}

int ilqr_count_running(ilqr_batch* h, int* n) {
  if (!h || !n) return fail(ILQR_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->device));
  std::vector<int> st(h->B);
  if (int rc = scalars_to_host(h, h->v.status, st.data())) return rc;
  int c = 0;
  for (int s : st) c += (s == 0);
  *n = c;
  return 0;
}

// Slot j <- slot perm[j] for every per-trajectory array a running solve carries (tiled: x0, xs, us, k, K; scalars: cost,
// lambda, dlambda, dV, gnorm, status, iters, flgChange, alpha index, diverge, backpass_done).  Candidates and derivative
// records are not moved: no accept is pending between ilqr_iterate calls, the records are recomputed when asked for, and
// the candidates are marked as nobody's (ilqr_get_candidate then fails with ILQR_ERR_STATE until the next rollout).
static int apply_permutation(ilqr_batch* h, const std::vector<int>& perm) {
  const size_t es = elem_size(h), Bp = (size_t)h->Bp;
  const size_t biggest = std::max<size_t>((size_t)h->ntiles * (h->T + 1) * h->nx * TW, (size_t)h->ntiles * h->T * h->nu * h->nx * TW) * es;
  const size_t need = std::max<size_t>(biggest, 2 * Bp * sizeof(double));
  if (h->perm_scratch_bytes < need) {
    if (h->perm_scratch) HIPCHK(hipFree(h->perm_scratch));
    h->perm_scratch = nullptr;
    HIPCHK(hipMalloc(&h->perm_scratch, need));
    h->perm_scratch_bytes = need;
  }
  if (!h->d_perm) HIPCHK(hipMalloc((void**)&h->d_perm, Bp * sizeof(int)));
  HIPCHK(hipMemcpyAsync(h->d_perm, perm.data(), Bp * sizeof(int), hipMemcpyHostToDevice, h->stream));
  auto tiled = [&](void* arr, int S, int E) -> int {
    const size_t n = (size_t)h->ntiles * S * E * TW;
    if (h->dtype == ILQR_DTYPE_F32)
      hipLaunchKernelGGL(k_permute_tiled<float>, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const float*)arr, (float*)h->perm_scratch, h->d_perm, h->ntiles, S, E);
    else
      hipLaunchKernelGGL(k_permute_tiled<double>, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const double*)arr, (double*)h->perm_scratch, h->d_perm, h->ntiles, S, E);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(arr, h->perm_scratch, n * es, hipMemcpyDeviceToDevice, h->stream));
    return 0;
  };
  auto scalar = [&](auto* arr, size_t n) -> int {
    using T = std::remove_pointer_t<decltype(arr)>;
    hipLaunchKernelGGL(k_permute_scalar<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const T*)arr, (T*)h->perm_scratch, h->d_perm, (int)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(arr, h->perm_scratch, n * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    return 0;
  };
  BatchView& v = h->v;
  int rc = 0;
  rc |= tiled(v.x0, 1, h->nx);
  rc |= tiled(v.xs, h->T + 1, h->nx);
  rc |= tiled(v.us, h->T, h->nu);
  rc |= tiled(v.kff, h->T, h->nu);
  rc |= tiled(v.Kfb, h->T, h->nu * h->nx);
  rc |= scalar(v.cost, Bp);
  rc |= scalar(v.lambda, Bp);
  rc |= scalar(v.dlambda, Bp);
  rc |= scalar(v.dV, Bp);
  rc |= scalar(v.dV + Bp, Bp);
  rc |= scalar(v.gnorm, Bp);
  rc |= scalar(v.status, Bp);
  rc |= scalar(v.iters, Bp);
  rc |= scalar(v.flg_change, Bp);
  rc |= scalar(v.alpha_idx, Bp);
  rc |= scalar(v.diverge, Bp);
  rc |= scalar(v.backpass_done, Bp);
  if (rc) return rc;
  h->recs = ilqr_batch::REC_STALE;
  h->cands_valid = false;  // the candidates stayed where they were
  return 0;
}

int ilqr_generate_trajectory(ilqr_batch* h) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  if (!h->initialised) return fail(ILQR_ERR_STATE, "generate_trajectory needs x0/xs/us (asserts of ilqr_core.cpp:80-82)");
  // Intra-tile compaction (the reference's own TODO, notes.md:16): a tile costs what its slowest trajectory costs, and with
  // more tiles than the device holds at once (ntiles > #CU) every tile that still has ONE running trajectory takes a slot.
  // So a big batch is solved in chunks of iterations; after a chunk, if the running trajectories would fit into at most
  // half of the tiles that are still launched, they are re-packed into the leading tiles (one permutation pass over
  // the per-trajectory arrays, ~0.1 ms per 4096 trajectories), and only those tiles are launched from then on.  The
  // original order is restored before returning.  Trajectories never interact and no kernel's arithmetic depends on a
  // trajectory's slot: statuses, iteration counts and costs are bit-identical (tests/test_gpu_full_solves.py).
  const bool persistent = use_persistent(h);
  const bool compacting = persistent && h->ntiles > h->num_cus && !(h->sp.fixed_work) && !h->route.no_compaction;
  int done_iters = 0;
  const int chunk = persistent ? (compacting ? std::min(std::max(1, h->params.max_iter), 8) : std::max(1, h->params.max_iter)) : 10;  // (a persistent tile stops by itself)
  std::vector<int> slot_orig;  // slot j currently holds original trajectory slot_orig[j] (empty: identity)
  h->active_tiles = h->ntiles;
  // Every exit of the chunk loop -- also a failing HIP call -- goes through the restore below: the handle is never left
  // with permuted slots or a subset of tiles behind the caller's back.
  auto chunks = [&]() -> int {
  while (done_iters < h->params.max_iter) {
    const int n = std::min(chunk, h->params.max_iter - done_iters);
    if (int rc = ilqr_iterate(h, n)) return rc;
    done_iters += n;
    int running = 0;
    HIPCHK(hipMemcpyAsync(&running, h->v.n_running, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (running == 0) break;
    if (compacting && done_iters < h->params.max_iter && 2 * ((running + TW - 1) / TW) <= h->active_tiles && h->active_tiles > 1) {
      std::vector<int> st(h->Bp);
      HIPCHK(hipMemcpyAsync(st.data(), h->v.status, (size_t)h->Bp * sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      if (slot_orig.empty()) {
        slot_orig.resize(h->Bp);
        for (int j = 0; j < h->Bp; j++) slot_orig[j] = j;
      }
      std::vector<int> perm;
      perm.reserve(h->Bp);
      for (int j = 0; j < h->Bp; j++)
        if (st[j] == 0) perm.push_back(j);
      const int n_run = (int)perm.size();
      for (int j = 0; j < h->Bp; j++)
        if (st[j] != 0) perm.push_back(j);
      if (int rc = apply_permutation(h, perm)) {
        h->initialised = false;  // a half-applied permutation: the arrays no longer describe one batch
        return rc;
      }
      std::vector<int> so(h->Bp);
      for (int j = 0; j < h->Bp; j++) so[j] = slot_orig[perm[j]];
      slot_orig.swap(so);
      h->active_tiles = std::max(1, (n_run + TW - 1) / TW);
    }
  }
  return 0;
  };
  const int rc_out = chunks();
  h->active_tiles = h->ntiles;
  if (!slot_orig.empty() && h->initialised) {  // back to the caller's order: slot o <- the slot that holds original trajectory o
    std::vector<int> back(h->Bp);
    for (int j = 0; j < h->Bp; j++) back[slot_orig[j]] = j;
    if (int rc = apply_permutation(h, back)) {
      h->initialised = false;  // (stage calls and getters of trajectories then fail with ILQR_ERR_STATE instead of reporting the wrong order)
      return rc_out ? rc_out : rc;
    }
  }
  return rc_out;
}

int ilqr_solve(ilqr_batch* h, const double* x0, const double* u0) {
  if (int rc = ilqr_init_traj(h, x0, u0, nullptr)) return rc;
  return ilqr_generate_trajectory(h);
}

int ilqr_warm_start(ilqr_batch* h, const double* x0) {
  if (!h || !x0) return fail(ILQR_ERR_INVALID, "null argument");
  if (host_model(h)) return no_device_model();
  if (!h->initialised) return fail(ILQR_ERR_STATE, "warm start needs a previous solve (assert us.size()>0, ilqr_core.cpp:66)");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = upload(h, x0, h->v.x0, 1, h->nx)) return rc;
  // forward_pass(x_0, us) with the stored gains: u = us[t] + K[t](x - xs[t])  (alpha*k term = 0)
  AlphaSet al;
  for (int i = 0; i < NALPHA; i++) al.a[i] = 0.0;
  if (generic_twin(h)) {  // generic path: the rollout itself overwrites xs/us (slot 0 of `al` for everyone)
    HIPCHK(hipMemsetAsync(h->commit_idx, 0, (size_t)h->Bp * sizeof(int), h->stream));
    if (int rc = launch_rollout(h, true, true, al, 1, h->v.cost, 0)) return rc;
  } else {
    if (int rc = launch_rollout(h, true, true, al, 1, h->v.cost, 0)) return rc;
    HIPCHK(hipMemsetAsync(h->commit_idx, 0, (size_t)h->Bp * sizeof(int), h->stream));  // slot 0 for everyone
    if (int rc = launch_commit(h)) return rc;
  }
  HIPCHK(hipMemsetAsync(h->commit_idx, 0xFF, (size_t)h->Bp * sizeof(int), h->stream));
  // a new outer loop starts: status/iters/flgChange reset, lambda & dlambda persist (file statics)
  std::vector<double> lam(h->B), dlam(h->B);
  if (int rc = scalars_to_host(h, h->v.lambda, lam.data())) return rc;
  if (int rc = scalars_to_host(h, h->v.dlambda, dlam.data())) return rc;
  hipLaunchKernelGGL(k_reset_state<double>, dim3((h->Bp + 255) / 256), dim3(256), 0, h->stream, h->v, 1.0, 1.0);
  HIPCHK(hipGetLastError());
  if (int rc = scalars_to_dev(h, lam.data(), h->v.lambda)) return rc;
  if (int rc = scalars_to_dev(h, dlam.data(), h->v.dlambda)) return rc;
  return ilqr_generate_trajectory(h);
}

// ---- stages --------------------------------------------------------------------------------
int ilqr_compute_derivatives(ilqr_batch* h) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  if (host_model(h)) return no_device_model();
  HIPCHK(hipSetDevice(h->device));
  if (!h->initialised) return fail(ILQR_ERR_STATE, "stage call before ilqr_init_traj/ilqr_set_trajectory (the reference asserts, ilqr_core.cpp:80-82)");
  return launch_derivatives(h, 1);
}

int ilqr_backward_pass(ilqr_batch* h, int* diverge_out) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (!h->initialised) return fail(ILQR_ERR_STATE, "stage call before ilqr_init_traj/ilqr_set_trajectory (the reference asserts, ilqr_core.cpp:80-82)");
  if (int rc = launch_backward(h, 0)) return rc;
  if (diverge_out) return scalars_to_host(h, h->v.diverge, diverge_out);
  return 0;
}

int ilqr_backward_step(ilqr_batch* h) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (!h->initialised) return fail(ILQR_ERR_STATE, "stage call before ilqr_init_traj/ilqr_set_trajectory (the reference asserts, ilqr_core.cpp:80-82)");
  return launch_backward(h, 1);
}

int ilqr_rollout_candidates(ilqr_batch* h, double* cost_out) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  if (host_model(h)) return no_device_model();
  HIPCHK(hipSetDevice(h->device));
  if (!h->initialised) return fail(ILQR_ERR_STATE, "stage call before ilqr_init_traj/ilqr_set_trajectory (the reference asserts, ilqr_core.cpp:80-82)");
  if (int rc = do_rollout_candidates(h, 0)) return rc;
  if (cost_out) {
    std::vector<double> tmp((size_t)NALPHA * h->Bp);
    HIPCHK(hipMemcpyAsync(tmp.data(), h->v.cost_c, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int b = 0; b < h->B; b++)
      for (int a = 0; a < NALPHA; a++) cost_out[(size_t)b * NALPHA + a] = tmp[(size_t)a * h->Bp + b];
  }
  return 0;
}

int ilqr_line_search(ilqr_batch* h) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  if (host_model(h)) return no_device_model();
  HIPCHK(hipSetDevice(h->device));
  if (!h->initialised) return fail(ILQR_ERR_STATE, "stage call before ilqr_init_traj/ilqr_set_trajectory (the reference asserts, ilqr_core.cpp:80-82)");
  if (int rc = flush_commit(h)) return rc;
  if (int rc = do_rollout_candidates(h, 1)) return rc;
  if (int rc = launch_accept(h)) return rc;
  return flush_commit(h);
}

int ilqr_accept_candidates(ilqr_batch* h, const double* cost_c, int* accepted) {
  if (!h || !cost_c || !accepted) return fail(ILQR_ERR_INVALID, "null argument");
  if (!h->v.cost_c) return fail(ILQR_ERR_UNSUPPORTED, "this handle has no candidate-cost buffer");
  HIPCHK(hipSetDevice(h->device));
  if (!h->initialised) return fail(ILQR_ERR_STATE, "stage call before ilqr_init_traj/ilqr_set_trajectory (the reference asserts, ilqr_core.cpp:80-82)");
  if (int rc = flush_commit(h)) return rc;
  std::vector<double> tmp((size_t)NALPHA * h->Bp, 0.0);
  for (int b = 0; b < h->B; b++)
    for (int a = 0; a < NALPHA; a++) tmp[(size_t)a * h->Bp + b] = cost_c[(size_t)b * NALPHA + a];
  HIPCHK(hipMemcpyAsync(h->v.cost_c, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemsetAsync(h->v.n_running, 0, sizeof(int), h->stream));  // (the device sweeps do this for k_accept)
  if (int rc = launch_accept(h)) return rc;
  std::vector<int> ci(h->Bp);
  HIPCHK(hipMemcpyAsync(ci.data(), h->commit_idx, (size_t)h->Bp * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemsetAsync(h->commit_idx, 0xFF, (size_t)h->Bp * sizeof(int), h->stream));  // the caller commits
  HIPCHK(hipStreamSynchronize(h->stream));
  h->commit_pending = false;
  for (int b = 0; b < h->B; b++) accepted[b] = ci[b];
  return 0;
}

int ilqr_reset_state(ilqr_batch* h, int warm) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = flush_commit(h)) return rc;
  std::vector<double> lam, dlam;
  if (warm) {
    lam.resize(h->B);
    dlam.resize(h->B);
    if (int rc = scalars_to_host(h, h->v.lambda, lam.data())) return rc;
    if (int rc = scalars_to_host(h, h->v.dlambda, dlam.data())) return rc;
  } else {
    const size_t T = h->T, T1 = h->T + 1;
    if (h->v.D) HIPCHK(hipMemsetAsync(h->v.D, 0, dev_elems(h, T1, rec_of(h)) * elem_size(h), h->stream));
    h->recs = ilqr_batch::REC_ZERO;
    HIPCHK(hipMemsetAsync(h->v.kff, 0, dev_elems(h, T, h->nu) * elem_size(h), h->stream));
    HIPCHK(hipMemsetAsync(h->v.Kfb, 0, dev_elems(h, T, h->nu * h->nx) * elem_size(h), h->stream));
    if (int rc = forget_pending(h)) return rc;
  }
  hipLaunchKernelGGL(k_reset_state<double>, dim3((h->Bp + 255) / 256), dim3(256), 0, h->stream, h->v, h->params.lambda_init,
                     h->params.dlambda_init);
  HIPCHK(hipGetLastError());
  if (warm) {
    if (int rc = scalars_to_dev(h, lam.data(), h->v.lambda)) return rc;
    if (int rc = scalars_to_dev(h, dlam.data(), h->v.dlambda)) return rc;
  }
  return 0;
}

// ---- state exchange --------------------------------------------------------------------------
int ilqr_set_trajectory(ilqr_batch* h, const double* x0, const double* xs, const double* us, const double* cost) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (x0) if (int rc = upload(h, x0, h->v.x0, 1, h->nx)) return rc;
  if (xs) if (int rc = upload(h, xs, h->v.xs, h->T + 1, h->nx)) return rc;
  if (us) if (int rc = upload(h, us, h->v.us, h->T, h->nu)) return rc;
  if (cost) if (int rc = scalars_to_dev(h, cost, h->v.cost)) return rc;
  h->initialised = true;
  return 0;
}
int ilqr_set_gains(ilqr_batch* h, const double* k, const double* K) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (k) if (int rc = upload(h, k, h->v.kff, h->T, h->nu)) return rc;
  if (K) if (int rc = upload(h, K, h->v.Kfb, h->T, h->nu * h->nx)) return rc;
  return 0;
}
int ilqr_set_derivatives(ilqr_batch* h, const double* fx, const double* fu, const double* cx, const double* cu,
                         const double* cxx, const double* cxu, const double* cuu) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  int off[7], len[7];
  rec_offsets(h->nx, h->nu, off, len);
  const double* srcs[7] = {fx, fu, cx, cu, cxx, cxu, cuu};
  for (int i = 0; i < 7; i++)
    if (srcs[i]) if (int rc = upload_rec(h, srcs[i], off[i], len[i])) return rc;
  return 0;
}
int ilqr_set_lambda(ilqr_batch* h, const double* lambda, const double* dlambda) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (lambda) if (int rc = scalars_to_dev(h, lambda, h->v.lambda)) return rc;
  if (dlambda) if (int rc = scalars_to_dev(h, dlambda, h->v.dlambda)) return rc;
  return 0;
}

int ilqr_get_trajectory(ilqr_batch* h, double* xs, double* us) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (xs) if (int rc = download(h, h->v.xs, xs, h->T + 1, h->nx)) return rc;
  if (us) if (int rc = download(h, h->v.us, us, h->T, h->nu)) return rc;
  return 0;
}
int ilqr_get_gains(ilqr_batch* h, double* k, double* K) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (k) if (int rc = download(h, h->v.kff, k, h->T, h->nu)) return rc;
  if (K) if (int rc = download(h, h->v.Kfb, K, h->T, h->nu * h->nx)) return rc;
  return 0;
}
int ilqr_get_derivatives(ilqr_batch* h, double* fx, double* fu, double* cx, double* cu, double* cxx, double* cxu,
                         double* cuu) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  int off[7], len[7];
  rec_offsets(h->nx, h->nu, off, len);
  double* dsts[7] = {fx, fu, cx, cu, cxx, cxu, cuu};
  if (int rc = materialise_records(h)) return rc;
  for (int i = 0; i < 7; i++)
    if (dsts[i]) if (int rc = download_rec(h, dsts[i], off[i], len[i])) return rc;
  return 0;
}
int ilqr_get_cost(ilqr_batch* h, double* cost) {
  if (!h || !cost) return fail(ILQR_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->device));
  return scalars_to_host(h, h->v.cost, cost);
}
int ilqr_get_lambda(ilqr_batch* h, double* lambda, double* dlambda) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (lambda) if (int rc = scalars_to_host(h, h->v.lambda, lambda)) return rc;
  if (dlambda) if (int rc = scalars_to_host(h, h->v.dlambda, dlambda)) return rc;
  return 0;
}
int ilqr_get_dV(ilqr_batch* h, double* dV) {
  if (!h || !dV) return fail(ILQR_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->device));
  std::vector<double> tmp(2 * (size_t)h->Bp);
  HIPCHK(hipMemcpyAsync(tmp.data(), h->v.dV, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->B; b++) {
    dV[2 * b] = tmp[b];
    dV[2 * b + 1] = tmp[h->Bp + b];
  }
  return 0;
}
int ilqr_get_gnorm(ilqr_batch* h, double* g) {
  if (!h || !g) return fail(ILQR_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->device));
  return scalars_to_host(h, h->v.gnorm, g);
}
int ilqr_get_status(ilqr_batch* h, int* status, int* iters, int* alpha_idx) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (status) if (int rc = scalars_to_host(h, h->v.status, status)) return rc;
  if (iters) if (int rc = scalars_to_host(h, h->v.iters, iters)) return rc;
  if (alpha_idx) if (int rc = scalars_to_host(h, h->v.alpha_idx, alpha_idx)) return rc;
  return 0;
}
int ilqr_get_candidate(ilqr_batch* h, int a, double* xs, double* us) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  if (host_model(h)) return no_device_model();
  REQUIRE(a >= 0 && a < NALPHA, "alpha index %d out of range", a);
  if (!h->cands_valid)
    return fail(ILQR_ERR_STATE, "no candidates: none rolled out yet, or the solve re-packed running trajectories (compaction) and left the "
                                "candidate buffers behind -- call ilqr_rollout_candidates / ilqr_iterate first");
  HIPCHK(hipSetDevice(h->device));
  if (h->aos) {  // generic handles keep candidates only on the LQ matrix-core route: [b][alpha][t][row], whole trajectories
    if (!(h->model == ILQR_MODEL_LQ && h->lq_cands_kept && h->v.cand_x))
      return fail(ILQR_ERR_UNSUPPORTED, "this handle's line search keeps no candidate trajectories (only their costs: ilqr_rollout_candidates)");
    const size_t wx = (size_t)(h->T + 1) * h->nx * sizeof(double), wu = (size_t)h->T * h->nu * sizeof(double);
    if (xs) HIPCHK(hipMemcpy2DAsync(xs, wx, (const char*)h->v.cand_x + (size_t)a * wx, (size_t)NALPHA * wx, wx, h->B, hipMemcpyDeviceToHost, h->stream));
    if (us) HIPCHK(hipMemcpy2DAsync(us, wu, (const char*)h->v.cand_u + (size_t)a * wu, (size_t)NALPHA * wu, wu, h->B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
  }
  const size_t nx_el = (size_t)h->B * (h->T + 1) * h->nx, nu_el = (size_t)h->B * h->T * h->nu;
  if (int rc = ensure_staging(h, nx_el + nu_el)) return rc;
  double* dxs = h->staging;
  double* dus = h->staging + nx_el;
  const dim3 grid(grid_for((size_t)h->B * (h->T + 1), 256)), block(256);
  if (int rc = with_model(h, [&](auto& v, auto& m, auto&) {
        using MM = std::decay_t<decltype(m)>;
        if constexpr (MM::NX == 4 && MM::NU == 1) {
          if (h->cands_grouped) {
            hipLaunchKernelGGL((k_unpack_cand<MM, true>), grid, block, 0, h->stream, v, m, a, dxs, dus);
            return 0;
          }
        }
        hipLaunchKernelGGL((k_unpack_cand<MM>), grid, block, 0, h->stream, v, m, a, dxs, dus);
        return 0;
      }))
    return rc;
  HIPCHK(hipGetLastError());
  if (xs) HIPCHK(hipMemcpyAsync(xs, dxs, nx_el * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (us) HIPCHK(hipMemcpyAsync(us, dus, nu_el * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
int ilqr_copy_cost_to_device(ilqr_batch* h, void* dst) {
  if (!h || !dst) return fail(ILQR_ERR_INVALID, "null argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(dst, h->v.cost, (size_t)h->B * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return 0;
}
// the handle's (tiled, possibly float) array -> canonical double [B][S][E] in device memory `dst`; enqueued, not waited for
static int unpack_to_device(ilqr_batch* h, const void* src, double* dst, int S, int E) {
  const size_t n = (size_t)h->B * S * E;
  if (h->aos) {  // generic handles: the canonical layout IS the device layout
    HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return 0;
  }
  if (h->dtype == ILQR_DTYPE_F32)
    hipLaunchKernelGGL(k_unpack<float>, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const float*)src, dst, h->B, S, E);
  else
    hipLaunchKernelGGL(k_unpack<double>, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const double*)src, dst, h->B, S, E);
  HIPCHK(hipGetLastError());
  return 0;
}
int ilqr_copy_trajectory_to_device(ilqr_batch* h, void* xs_dev, void* us_dev) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (xs_dev) if (int rc = unpack_to_device(h, h->v.xs, (double*)xs_dev, h->T + 1, h->nx)) return rc;
  if (us_dev) if (int rc = unpack_to_device(h, h->v.us, (double*)us_dev, h->T, h->nu)) return rc;
  return 0;
}
int ilqr_copy_gains_to_device(ilqr_batch* h, void* k_dev, void* K_dev) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (k_dev) if (int rc = unpack_to_device(h, h->v.kff, (double*)k_dev, h->T, h->nu)) return rc;
  if (K_dev) if (int rc = unpack_to_device(h, h->v.Kfb, (double*)K_dev, h->T, h->nu * h->nx)) return rc;
  return 0;
}
int ilqr_get_results_async(ilqr_batch* h, double* xs, double* us, double* k, double* K, double* cost) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  struct Item { const void* src; double* dst; int S, E; };
  const Item items[4] = {{h->v.xs, xs, h->T + 1, h->nx}, {h->v.us, us, h->T, h->nu}, {h->v.kff, k, h->T, h->nu}, {h->v.Kfb, K, h->T, h->nu * h->nx}};
  if (h->aos) {
    for (const Item& it : items)
      if (it.dst) HIPCHK(hipMemcpyAsync(it.dst, it.src, (size_t)h->B * it.S * it.E * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  } else {
    // every array gets its own stretch of the staging buffer: the unpack kernels and the copies follow each other on the stream without a
    // host synchronisation in between (the staging buffer is only ever touched by work enqueued on this stream: a later upload is ordered
    // behind these copies)
    size_t total = 0;
    for (const Item& it : items)
      if (it.dst) total += (size_t)h->B * it.S * it.E;
    if (total)
      if (int rc = ensure_staging(h, total)) return rc;
    size_t off = 0;
    for (const Item& it : items) {
      if (!it.dst) continue;
      const size_t n = (size_t)h->B * it.S * it.E;
      if (int rc = unpack_to_device(h, it.src, h->staging + off, it.S, it.E)) return rc;
      HIPCHK(hipMemcpyAsync(it.dst, h->staging + off, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      off += n;
    }
  }
  if (cost) HIPCHK(hipMemcpyAsync(cost, h->v.cost, (size_t)h->B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return 0;
}
int ilqr_host_register(void* ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(ILQR_ERR_INVALID, "null argument");
  HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return 0;
}
int ilqr_host_unregister(void* ptr) {
  if (!ptr) return fail(ILQR_ERR_INVALID, "null argument");
  HIPCHK(hipHostUnregister(ptr));
  return 0;
}

}  // extern "C"

#include "group.hpp"
#include "profile.hpp"
