// capi.hip -- implementation of the C ABI of include/ilqr_amd.h on top of the HIP kernels.
//
// One opaque handle (ilqr_batch) owns all device memory of a batch in the tiled layout of
// common.hpp, a HIP stream, and the per-stage HIP-event timers.  No CPU compute path exists:
// every entry point either launches kernels or moves bytes.
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
#include "kernels.hpp"

using namespace ilqr;

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIPCHK(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(ILQR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define REQUIRE(cond, ...)                              \
  do {                                                  \
    if (!(cond)) return fail(ILQR_ERR_INVALID, __VA_ARGS__); \
  } while (0)

// ------------------------------------------------------------------------------------------
// the handle
// ------------------------------------------------------------------------------------------
struct StageTimer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;  // (begin, end); an event may end one stage and begin the next
  double ms = 0;
  int launches = 0;
};

struct ilqr_batch {
  int model, nx, nu, T, B, Bp, ntiles, device, flags;
  int dtype = ILQR_DTYPE_F64;   // arithmetic of the nx = 4 device models (ilqr_desc.dtype)
  double dt;
  ilqr_params params;
  // fp64 handle: its models.  fp32 handle: the double-precision TWINS the finite differences are taken in
  // (kernels.hpp, derivatives_of_knot), built from the float models' own parameter values
  AcrobotModel acrobot;
  DoubleIntegratorModel dint;
  AcrobotModelT<float> acrobot_f;          // fp32 handle: what the rollouts integrate
  DoubleIntegratorModelT<float> dint_f;
  LqModel lq;                   // ILQR_MODEL_LQ: padded matrices on the device
#ifdef ILQR_HAVE_USER_MODEL
  UserModelT<double> user;      // ILQR_MODEL_USER: the build's user device twin (fp32 handle: the twin the finite differences are taken in)
  GenericModelOf<UserModelT<double>> user_g;  // ... as the generic kernels take it (any NX <= 32, NU <= 16 that is not a tiled nx = 4 shape)
  UserModelT<float> user_f;
#endif
  // v is the view every entry point addresses arrays through; for an fp32 handle its trajectory pointers hold
  // the addresses of FLOAT arrays (never dereferenced as double: kernels get vf, the same addresses typed float*)
  BatchView v;
  BatchViewT<float> vf;
  SolverParams sp;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int* commit_idx = nullptr;
  long long* phase_ticks = nullptr;  // [ntiles][5] per-tile clocks of k_solve_tile: sweep+backward, rollouts+accept, iterations, shader cycles, wall ticks
  double wall_clock_khz = 100000.0;
  double* staging = nullptr;  // device scratch for canonical <-> tiled conversion
  // LQ model with exact derivatives: the sweep writes one copy of the constant matrices (const_rec) and
  // per knot only cx, cu; records_partial says that D holds no matrices for t < T right now
  double* const_rec = nullptr;   // [2][REC]: the constant blocks of every knot t < T, then knot T's record
  bool records_partial = false;
  // ... on the k_backward_w3 route (lq_fused) no sweep runs at all: the backward pass forms cx = cxx x_t, cu = cuu u_t from the knot, and the
  // record array D is allocated only if somebody asks for records (getters, ilqr_set_derivatives, the finite-difference mode)
  bool lq_fused = false;          // the handle can take that route (LQ model, exact derivatives, k_backward_w3, no ILQR_ROUTE_FULL_RECORDS)
  bool lq_fused_stale = false;    // fused iterations have run since D was last written: a getter gets the records of the current nominal computed
  bool lq_caller_records = false; // ilqr_set_derivatives replaced the model's blocks: the next backward pass reads D, not the model
  // nx = 4 device models: D (0.75 GB per 4096 acrobot trajectories) is allocated the first time somebody wants
  // records in HBM -- the stage calls, the two-kernel route, the getters.  ilqr_iterate's fused kernel keeps them
  // in LDS (kernels.hpp) and leaves D as it was: recs says what D holds.
  //   REC_ZERO  what init_traj leaves (ilqr_core.cpp:39-45): zeros        REC_VALID  the records of the nominal
  //   REC_STALE iterations have run since: whoever asks gets them computed from the current nominal
  enum { REC_ZERO, REC_VALID, REC_STALE } recs = REC_ZERO;
  size_t staging_elems = 0;
  std::vector<void*> allocs;
  bool initialised = false;  // init_traj / set_trajectory has run
  bool commit_pending = false;  // an accepted candidate is not yet copied into xs/us
  bool lq_cands_kept = false;   // LQ model: the last search rollout (k_rollout_lq<RG_SEARCH>) stored its candidates in v.cand_x / v.cand_u
  // cand_u / cand_x / cost_c hold, slot for slot, the last rollouts of the trajectories now in those slots.  Compaction
  // (ilqr_generate_trajectory) moves trajectories without moving their candidates: after it they belong to nobody.
  bool cands_valid = false;
  bool aos = false;             // host-model / generic handles: trajectory-contiguous layout, wave-per-trajectory backward
  double* d_umin = nullptr;     // [nu] device copies of the limits (generic kernel)
  double* d_umax = nullptr;
  bool profile = false;
  int num_cus = 256;
  // full solves of batches with more tiles than CUs: running trajectories are re-packed into the leading tiles between
  // chunks of iterations (ilqr_generate_trajectory); active_tiles = how many tiles the persistent kernel is launched for
  int active_tiles = 0;
  int* d_perm = nullptr;       // [Bp]
  void* perm_scratch = nullptr;  // as large as the largest per-knot array
  size_t perm_scratch_bytes = 0;
  // Route choices for A/B runs and the bit-identity tests: ilqr_desc.route, fixed at ilqr_create -- a handle never changes
  // kernels between calls, and nothing is read from the environment
  struct {
    bool staged = false, unfused = false, backward_w1 = false, backward_w2 = false, lq_thread_rollout = false, full_records = false, no_compaction = false, quad_chain = false;
    int fused = 0;  // 0 = by batch size
    int wide_occ = 0;  // wide tiles per CU: 0 = by batch size
  } env;
  StageTimer timers[ILQR_NUM_STAGES];
  std::vector<hipEvent_t> event_pool;
  // inside ilqr_iterate nothing is enqueued between the end of one stage and the begin of the next: the
  // end event serves as the next begin (one event record per kernel boundary instead of two; the
  // records cost ~2.5 us each on the queue)
  bool chain_timers = false;
  hipEvent_t chain_event = nullptr;
};

static int rec_of(const ilqr_batch* h) { return rec_size(h->nx, h->nu); }
static size_t elem_size(const ilqr_batch* h) { return h->dtype == ILQR_DTYPE_F32 ? sizeof(float) : sizeof(double); }
// the float view of an fp32 handle: same addresses as v, typed
static void sync_float_view(ilqr_batch* h) {
  const BatchView& v = h->v;
  BatchViewT<float>& f = h->vf;
  f.B = v.B; f.Bp = v.Bp; f.ntiles = v.ntiles; f.T = v.T; f.dt = v.dt;
  f.x0 = (float*)v.x0; f.xs = (float*)v.xs; f.us = (float*)v.us; f.kff = (float*)v.kff; f.Kfb = (float*)v.Kfb;
  f.D = (float*)v.D; f.cand_u = (float*)v.cand_u; f.cand_x = (float*)v.cand_x; f.nch = v.nch;
  f.cost_c = v.cost_c; f.cost = v.cost; f.lambda = v.lambda; f.dlambda = v.dlambda; f.dV = v.dV; f.gnorm = v.gnorm;
  f.status = v.status; f.iters = v.iters; f.flg_change = v.flg_change; f.alpha_idx = v.alpha_idx; f.diverge = v.diverge;
  f.backpass_done = v.backpass_done; f.n_running = v.n_running; f.analytic = v.analytic;
}
// f(view, model, model the finite differences are taken in) for the handle's device model and arithmetic
template <class F>
static int with_model(ilqr_batch* h, F&& f) {
  if (h->dtype == ILQR_DTYPE_F32) {
    switch (h->model) {
      case ILQR_MODEL_ACROBOT: return f(h->vf, h->acrobot_f, h->acrobot);
      case ILQR_MODEL_DOUBLE_INTEGRATOR: return f(h->vf, h->dint_f, h->dint);
#ifdef ILQR_HAVE_USER_MODEL
      case ILQR_MODEL_USER:
        if constexpr (kUserTiled) return f(h->vf, h->user_f, h->user);
        break;
#endif
      default: break;
    }
  } else {
    switch (h->model) {
      case ILQR_MODEL_ACROBOT: return f(h->v, h->acrobot, h->acrobot);
      case ILQR_MODEL_DOUBLE_INTEGRATOR: return f(h->v, h->dint, h->dint);
#ifdef ILQR_HAVE_USER_MODEL
      case ILQR_MODEL_USER:
        if constexpr (kUserTiled) return f(h->v, h->user, h->user);
        break;
#endif
      default: break;
    }
  }
  return fail(ILQR_ERR_UNSUPPORTED, "model %d has no device kernels of this kind", h->model);
}
// f(view) for the handle's arithmetic
template <class F>
static int with_view(ilqr_batch* h, F&& f) {
  return h->dtype == ILQR_DTYPE_F32 ? f(h->vf) : f(h->v);
}
// ILQR_MODEL_HOST: the model exists only as host code; nothing but the backward pass runs here
static bool host_model(const ilqr_batch* h) { return h->model == ILQR_MODEL_HOST; }
static int no_device_model();
// elements of a per-knot array with S time slots of E doubles, in this handle's device layout
static size_t dev_elems(const ilqr_batch* h, size_t S, size_t E) {
  return h->aos ? (size_t)h->B * S * E : (size_t)h->ntiles * S * E * TW;
}

template <class T>
static int dev_alloc(ilqr_batch* h, T** p, size_t n) {
  void* q = nullptr;
  HIPCHK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
  HIPCHK(hipMemsetAsync(q, 0, std::max<size_t>(n, 1) * sizeof(T), h->stream));
  h->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

// trajectory arrays: n elements of the handle's arithmetic (the pointer keeps the view's nominal double* type)
static int dev_alloc_real(ilqr_batch* h, double** p, size_t n) {
  void* q = nullptr;
  const size_t bytes = std::max<size_t>(n, 1) * elem_size(h);
  HIPCHK(hipMalloc(&q, bytes));
  HIPCHK(hipMemsetAsync(q, 0, bytes, h->stream));
  h->allocs.push_back(q);
  *p = (double*)q;
  return 0;
}

static int grid_for(size_t n, int block) { return (int)std::min<size_t>((n + block - 1) / block, 65535u * 16u); }

static int no_device_model() {
  return fail(ILQR_ERR_UNSUPPORTED, "host-evaluated model: rollouts and finite differences stay on the host; only the backward pass (ilqr_set_derivatives + ilqr_backward_pass/_step) runs on the device");
}

// stage timing -------------------------------------------------------------------------------
static int timer_event(ilqr_batch* h, hipEvent_t* e) {
  if (!h->event_pool.empty()) {
    *e = h->event_pool.back();
    h->event_pool.pop_back();
    return 0;
  }
  HIPCHK(hipEventCreate(e));
  return 0;
}
static int timer_begin(ilqr_batch* h, int stage, std::pair<hipEvent_t, hipEvent_t>* ev) {
  if (!h->profile) return 0;
  (void)stage;
  if (h->chain_timers && h->chain_event) {
    ev->first = h->chain_event;
  } else {
    if (int rc = timer_event(h, &ev->first)) return rc;
    HIPCHK(hipEventRecord(ev->first, h->stream));
  }
  h->chain_event = nullptr;
  return timer_event(h, &ev->second);
}
static int timer_end(ilqr_batch* h, int stage, const std::pair<hipEvent_t, hipEvent_t>& ev) {
  if (!h->profile) return 0;
  HIPCHK(hipEventRecord(ev.second, h->stream));
  h->timers[stage].pending.push_back(ev);
  h->timers[stage].launches++;
  h->chain_event = h->chain_timers ? ev.second : nullptr;
  return 0;
}
static int timers_drain(ilqr_batch* h) {
  std::vector<hipEvent_t> used;
  for (int s = 0; s < ILQR_NUM_STAGES; s++) {
    StageTimer& t = h->timers[s];
    for (auto& ev : t.pending) {
      float ms = 0;
      HIPCHK(hipEventSynchronize(ev.second));
      HIPCHK(hipEventElapsedTime(&ms, ev.first, ev.second));
      t.ms += ms;
      used.push_back(ev.first);
      used.push_back(ev.second);
    }
    t.pending.clear();
  }
  std::sort(used.begin(), used.end());
  used.erase(std::unique(used.begin(), used.end()), used.end());
  h->event_pool.insert(h->event_pool.end(), used.begin(), used.end());
  h->chain_event = nullptr;
  return 0;
}

// host <-> device helpers -----------------------------------------------------------------------
static int ensure_staging(ilqr_batch* h, size_t elems) {
  if (elems <= h->staging_elems) return 0;
  if (h->staging) {
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipFree(h->staging));
    h->staging = nullptr;
    h->staging_elems = 0;
  }
  HIPCHK(hipMalloc((void**)&h->staging, elems * sizeof(double)));
  h->staging_elems = elems;
  return 0;
}
// canonical host [B][S][E] -> tiled device  (AoS handles: the canonical layout IS the device layout)
static int upload(ilqr_batch* h, const double* src, void* dst_tiled, int S, int E) {
  const size_t n = (size_t)h->B * S * E;
  if (h->aos) {
    HIPCHK(hipMemcpyAsync(dst_tiled, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
  }
  if (int rc = ensure_staging(h, n)) return rc;
  HIPCHK(hipMemcpyAsync(h->staging, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const size_t nt = (size_t)h->ntiles * S * E * TW;
  if (h->dtype == ILQR_DTYPE_F32)
    hipLaunchKernelGGL(k_pack<float>, dim3(grid_for(nt, 256)), dim3(256), 0, h->stream, h->staging, (float*)dst_tiled, h->B, h->ntiles, S, E);
  else
    hipLaunchKernelGGL(k_pack<double>, dim3(grid_for(nt, 256)), dim3(256), 0, h->stream, h->staging, (double*)dst_tiled, h->B, h->ntiles, S, E);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));  // staging is reused by the next call
  return 0;
}
static int download(ilqr_batch* h, const void* src_tiled, double* dst, int S, int E) {
  const size_t n = (size_t)h->B * S * E;
  if (h->aos) {
    HIPCHK(hipMemcpyAsync(dst, src_tiled, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
  }
  if (int rc = ensure_staging(h, n)) return rc;
  if (h->dtype == ILQR_DTYPE_F32)
    hipLaunchKernelGGL(k_unpack<float>, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const float*)src_tiled, h->staging, h->B, S, E);
  else
    hipLaunchKernelGGL(k_unpack<double>, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, (const double*)src_tiled, h->staging, h->B, S, E);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(dst, h->staging, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
static int launch_derivatives(ilqr_batch* h, int force);
// the record array of a tiled handle, allocated (zero-filled) on first use
static int ensure_records(ilqr_batch* h) {
  if (h->v.D) return 0;
  if (h->aos) return dev_alloc(h, &h->v.D, (size_t)h->B * (h->T + 1) * rec_of(h));  // (generic handles: on first use as well -- 44 GB at configs[4])
  if (int rc = dev_alloc_real(h, &h->v.D, (size_t)h->ntiles * (h->T + 1) * rec_of(h) * TW)) return rc;
  sync_float_view(h);
  return 0;
}
// D as the getters, the stage calls and ilqr_set_derivatives expect it.  LQ handles: fill in the constant
// matrices the partial sweep skipped.  nx = 4 handles: have the sweep compute the records of the current nominal
// trajectory if iterations have run since D was last written.
static int materialise_records(ilqr_batch* h) {
  if (int rc = ensure_records(h)) return rc;
  if (!h->aos) {
    if (h->recs == ilqr_batch::REC_STALE) return launch_derivatives(h, 1);
    return 0;
  }
  if (h->lq_fused_stale) {  // the fused LQ route never wrote D: whole exact records of the current nominal, now
    h->lq_fused_stale = false;
    h->records_partial = false;
    const int nchunk = (h->T + 1 + kAnalyticChunk - 1) / kAnalyticChunk;
    hipLaunchKernelGGL(k_analytic_lq, dim3(h->B * nchunk), dim3(64), 0, h->stream, h->v, h->lq, 1, 0, h->const_rec, kAnalyticChunk);
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (!h->records_partial) return 0;
  const int nchunk = (h->T + 1 + kAnalyticChunk - 1) / kAnalyticChunk;
  hipLaunchKernelGGL(k_analytic_lq, dim3(h->B * nchunk), dim3(64), 0, h->stream, h->v, h->lq, 1, 2, h->const_rec, kAnalyticChunk);
  HIPCHK(hipGetLastError());
  return 0;
}
static int upload_rec(ilqr_batch* h, const double* src, int off, int E) {
  if (int rc = materialise_records(h)) return rc;
  h->records_partial = false;  // the caller's blocks replace the model's: every knot reads its own record again
  h->lq_caller_records = true;
  h->recs = ilqr_batch::REC_VALID;

  const int S = h->T + 1;
  const size_t n = (size_t)h->B * S * E;
  if (int rc = ensure_staging(h, n)) return rc;
  HIPCHK(hipMemcpyAsync(h->staging, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (h->aos) {
    hipLaunchKernelGGL(k_rec_aos, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, h->v.D, h->staging, h->B, S, rec_of(h), off, E, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
  }
  const size_t nt = (size_t)h->ntiles * S * E * TW;
  if (h->dtype == ILQR_DTYPE_F32)
    hipLaunchKernelGGL(k_pack_rec<float>, dim3(grid_for(nt, 256)), dim3(256), 0, h->stream, h->staging, h->vf.D, h->B, h->ntiles, S, rec_of(h), off, E);
  else
    hipLaunchKernelGGL(k_pack_rec<double>, dim3(grid_for(nt, 256)), dim3(256), 0, h->stream, h->staging, h->v.D, h->B, h->ntiles, S, rec_of(h), off, E);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
static int download_rec(ilqr_batch* h, double* dst, int off, int E) {
  const int S = h->T + 1;
  const size_t n = (size_t)h->B * S * E;
  if (int rc = ensure_staging(h, n)) return rc;
  if (h->aos) {
    hipLaunchKernelGGL(k_rec_aos, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, h->v.D, h->staging, h->B, S, rec_of(h), off, E, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dst, h->staging, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
  }
  if (h->dtype == ILQR_DTYPE_F32)
    hipLaunchKernelGGL(k_unpack_rec<float>, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, h->vf.D, h->staging, h->B, S, rec_of(h), off, E);
  else
    hipLaunchKernelGGL(k_unpack_rec<double>, dim3(grid_for(n, 256)), dim3(256), 0, h->stream, h->v.D, h->staging, h->B, S, rec_of(h), off, E);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(dst, h->staging, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
// per-trajectory scalar arrays [Bp] on device <-> [B] host
template <class T>
static int scalars_to_host(ilqr_batch* h, const T* dev, T* host) {
  HIPCHK(hipMemcpyAsync(host, dev, (size_t)h->B * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
template <class T>
static int scalars_to_dev(ilqr_batch* h, const T* host, T* dev) {
  HIPCHK(hipMemcpyAsync(dev, host, (size_t)h->B * sizeof(T), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

// ------------------------------------------------------------------------------------------
// kernel launchers (dispatch on the device model)
// ------------------------------------------------------------------------------------------
// cand = true: controls + checkpoint states go to the candidate buffers; false: straight into xs/us (init)
template <class V, class M>
static int launch_rollout_t(ilqr_batch* h, const V& v, const M& m, bool gains, bool cand, const AlphaSet& al, int n_alpha,
                            double* cost_out, int mode, bool with_accept) {
  const int aw = (n_alpha + 3) / 4;  // wavefronts per tile: 4 alphas each
  dim3 grid(h->ntiles), block(64 * aw);
  const bool deep = h->ntiles <= h->num_cus;  // one block per CU: deep prefetch (see k_rollout)
  if (gains && cand && with_accept && deep)
    hipLaunchKernelGGL((k_rollout<M, true, true, kDeepPrefetch<M>, true>), grid, block, 0, h->stream, v, m, al, n_alpha, cost_out, mode, h->sp, h->commit_idx);
  else if (gains && cand && with_accept)
    hipLaunchKernelGGL((k_rollout<M, true, true, 4, true>), grid, block, 0, h->stream, v, m, al, n_alpha, cost_out, mode, h->sp, h->commit_idx);
  else if (gains && cand && deep)
    hipLaunchKernelGGL((k_rollout<M, true, true, kDeepPrefetch<M>>), grid, block, 0, h->stream, v, m, al, n_alpha, cost_out, mode, h->sp, nullptr);
  else if (gains && cand)
    hipLaunchKernelGGL((k_rollout<M, true, true, 4>), grid, block, 0, h->stream, v, m, al, n_alpha, cost_out, mode, h->sp, nullptr);
  else if (!gains && !cand)
    hipLaunchKernelGGL((k_rollout<M, false, false>), grid, block, 0, h->stream, v, m, al, n_alpha, cost_out, mode, h->sp, nullptr);
  else
    return fail(ILQR_ERR_INVALID, "unsupported rollout variant");
  HIPCHK(hipGetLastError());
  return 0;
}
// Does the handle's model have a device twin in the GENERIC kernels (generic.hpp)?  The shipped LQ model, or the build's user
// model when its dimensions are not a tiled nx = 4 shape.
static bool generic_twin(const ilqr_batch* h) {
  return h->model == ILQR_MODEL_LQ || (h->model == ILQR_MODEL_USER && h->aos);
}
// f(model) for the handle's generic device twin
template <class F>
static int with_generic_model(ilqr_batch* h, F&& f) {
  if (h->model == ILQR_MODEL_LQ) return f(h->lq);
#ifdef ILQR_HAVE_USER_MODEL
  if constexpr (!kUserTiled)
    if (h->model == ILQR_MODEL_USER) return f(h->user_g);
#endif
  return fail(ILQR_ERR_UNSUPPORTED, "model %d has no generic device kernels", h->model);
}
// generic path (generic.hpp): what = RG_INIT / RG_SEARCH / RG_COMMIT.  The LQ model rolls out on the
// matrix cores (k_rollout_lq, one wavefront per trajectory); ILQR_ROUTE_LQ_THREAD_ROLLOUT selects the
// generic thread-per-rollout kernel (same results bit for bit; kept as the cross-check and as the
// template for device models without matrix structure).
// Does the handle's search kernel also accept and commit (k_rollout_lq<RG_SEARCH, true>)?  The LQ model's matrix-core rollout with candidate buffers.
static bool lq_search_accepts(const ilqr_batch* h) { return h->model == ILQR_MODEL_LQ && !h->env.lq_thread_rollout && h->v.cand_x != nullptr; }
template <class M>
static int launch_rollout_g(ilqr_batch* h, const M& m, int what, const AlphaSet& al, double* cost_out, int mode, int write_cost, bool with_accept = false) {
  if constexpr (std::is_same<M, LqModel>::value)
  if (!h->env.lq_thread_rollout) {
    const dim3 grid(h->B), block(64);
    if (what == RG_SEARCH && with_accept && h->v.cand_x) {
      hipLaunchKernelGGL((k_rollout_lq<RG_SEARCH, true>), grid, block, 0, h->stream, h->v, m, al, cost_out, h->commit_idx, mode, 0, h->sp);
      h->lq_cands_kept = true;
    } else if (what == RG_SEARCH) {
      hipLaunchKernelGGL((k_rollout_lq<RG_SEARCH>), grid, block, 0, h->stream, h->v, m, al, cost_out, nullptr, mode, 0, h->sp);
      h->lq_cands_kept = h->v.cand_x != nullptr;  // the commit of what the next accept chooses is a copy (launch_commit)
    } else if (what == RG_INIT)
      hipLaunchKernelGGL((k_rollout_lq<RG_INIT>), grid, block, 0, h->stream, h->v, m, al, cost_out, nullptr, 0, 1, h->sp);
    else
      hipLaunchKernelGGL((k_rollout_lq<RG_COMMIT>), grid, block, 0, h->stream, h->v, m, al, cost_out, h->commit_idx, 0, write_cost, h->sp);
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (what == RG_SEARCH)
    hipLaunchKernelGGL((k_rollout_g<M, RG_SEARCH>), dim3((h->B + kSearchTraj - 1) / kSearchTraj), dim3(64), 0, h->stream, h->v, m, al,
                       cost_out, nullptr, mode, 0, h->sp.fixes);
  else if (what == RG_INIT)
    hipLaunchKernelGGL((k_rollout_g<M, RG_INIT>), dim3((h->B + 63) / 64), dim3(64), 0, h->stream, h->v, m, al, cost_out, nullptr, 0, 1, h->sp.fixes);
  else
    hipLaunchKernelGGL((k_rollout_g<M, RG_COMMIT>), dim3((h->B + 63) / 64), dim3(64), 0, h->stream, h->v, m, al, cost_out,
                       h->commit_idx, 0, write_cost, h->sp.fixes);
  HIPCHK(hipGetLastError());
  return 0;
}

// with_accept (tiled models, 11-alpha search): the rollout kernel also does STEP 3/4 for its tile
static int launch_rollout(ilqr_batch* h, bool gains, bool cand, const AlphaSet& al, int n_alpha, double* cost_out, int mode,
                          bool with_accept = false) {
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (int rc = timer_begin(h, ILQR_STAGE_ROLLOUT, &ev)) return rc;
  int rc;
  if (generic_twin(h)) {
    rc = with_generic_model(h, [&](auto& m) {
      if (!gains) return launch_rollout_g(h, m, RG_INIT, al, cost_out, 0, 1);
      if (n_alpha == NALPHA) return launch_rollout_g(h, m, RG_SEARCH, al, cost_out, mode, 0, with_accept);
      return launch_rollout_g(h, m, RG_COMMIT, al, cost_out, 0, 1);  // a single closed-loop rollout written in place (warm start): slot commit_idx of `al`
    });
    if (rc) return rc;
    return timer_end(h, ILQR_STAGE_ROLLOUT, ev);
  }
  rc = with_model(h, [&](auto& v, auto& m, auto&) { return launch_rollout_t(h, v, m, gains, cand, al, n_alpha, cost_out, mode, with_accept); });
  if (rc) return rc;
  return timer_end(h, ILQR_STAGE_ROLLOUT, ev);
}

static AlphaSet line_search_alphas();
static int launch_commit(ilqr_batch* h) {
  if (generic_twin(h)) {
    if (h->model == ILQR_MODEL_LQ && h->lq_cands_kept) {  // the matrix-core search kept its eleven rollouts: copy the accepted one
      hipLaunchKernelGGL(k_commit_lq, dim3(h->B), dim3(256), 0, h->stream, h->v, h->nx, h->nu, h->commit_idx);
      HIPCHK(hipGetLastError());
      return 0;
    }
    // no stored candidates otherwise on the generic path: re-run the accepted rollout in place
    return with_generic_model(h, [&](auto& m) { return launch_rollout_g(h, m, RG_COMMIT, line_search_alphas(), h->v.cost, 0, 0); });
  }
  dim3 grid((h->T + 1 + 15) / 16, h->ntiles), block(256);
  if (int rc = with_model(h, [&](auto& v, auto& m, auto&) {
        hipLaunchKernelGGL((k_commit<std::decay_t<decltype(m)>>), grid, block, 0, h->stream, v, m, h->commit_idx);
        return 0;
      }))
    return rc;
  HIPCHK(hipGetLastError());
  return 0;
}

// copy an accepted-but-not-yet-copied candidate into the nominal trajectory now
static int flush_commit(ilqr_batch* h) {
  if (!h->commit_pending) return 0;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (int rc = timer_begin(h, ILQR_STAGE_ACCEPT, &ev)) return rc;
  if (int rc = launch_commit(h)) return rc;
  h->commit_pending = false;
  HIPCHK(hipMemsetAsync(h->commit_idx, 0xFF, (size_t)h->Bp * sizeof(int), h->stream));  // all -1
  return timer_end(h, ILQR_STAGE_ACCEPT, ev);
}

// A fresh solve starts: nothing of an earlier one may leak into it -- neither an accepted candidate whose
// copy is still pending (an ilqr_iterate that returned early on an error leaves one), nor the "records hold
// no matrices" state of an exact-derivative LQ sweep (init_traj promises zeroed records, ilqr_core.cpp:39-45).
static int forget_pending(ilqr_batch* h) {
  h->records_partial = false;
  h->lq_fused_stale = false;
  h->lq_caller_records = false;
  h->cands_valid = false;  // (candidates of an earlier solve are nobody's)
  h->commit_pending = false;
  HIPCHK(hipMemsetAsync(h->commit_idx, 0xFF, (size_t)h->Bp * sizeof(int), h->stream));  // all -1
  return 0;
}

static int launch_derivatives(ilqr_batch* h, int force) {
  if (generic_twin(h))  // the generic sweep has no fused commit: rebuild the accepted rollout first
    if (int rc = flush_commit(h)) return rc;
  if (h->lq_fused) {  // k_backward_w3<.., LQF> forms cx, cu from the knot itself: no sweep, no record array
    h->lq_fused_stale = true;
    h->lq_caller_records = false;
    return 0;
  }
  if (int rc = ensure_records(h)) return rc;
  h->recs = ilqr_batch::REC_VALID;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (int rc = timer_begin(h, ILQR_STAGE_DERIVATIVES, &ev)) return rc;
  dim3 grid((h->T + 1 + 15) / 16, h->ntiles), block(256);
  const int* ci = h->commit_pending ? h->commit_idx : nullptr;
  if (generic_twin(h)) {
    if (h->v.analytic && h->model == ILQR_MODEL_LQ) {
      const int what = h->env.full_records ? 0 : 1;  // (A/B runs and the bit-identity test)
      const int chunk = (what == 1) ? 4 * kAnalyticChunk : kAnalyticChunk;
      const int nchunk = (h->T + 1 + chunk - 1) / chunk;
      hipLaunchKernelGGL(k_analytic_lq, dim3(h->B * nchunk), dim3(64), 0, h->stream, h->v, h->lq, force, what, h->const_rec, chunk);
      h->records_partial = (what == 1);
    } else {
      if (int rc = with_generic_model(h, [&](auto& m) {
            hipLaunchKernelGGL((k_derivatives_g<std::decay_t<decltype(m)>>), dim3(h->B * (h->T + 1)), dim3(64), 0, h->stream, h->v, m, force);
            return 0;
          }))
        return rc;
    }
    HIPCHK(hipGetLastError());
    return timer_end(h, ILQR_STAGE_DERIVATIVES, ev);
  }
  if (int rc = with_model(h, [&](auto& v, auto& m, auto& fdm) {
        hipLaunchKernelGGL((k_derivatives<std::decay_t<decltype(m)>, std::decay_t<decltype(fdm)>>), grid, block, 0, h->stream, v, m, fdm, force, ci);
        return 0;
      }))
    return rc;
  HIPCHK(hipGetLastError());
  // the kernel above performed the copy on the way; commit_idx is rewritten for every trajectory
  // by the next k_accept and only read while commit_pending is set, so it needs no reset here
  h->commit_pending = false;
  return timer_end(h, ILQR_STAGE_DERIVATIVES, ev);
}

static bool use_quad_backward(const ilqr_batch* h) {
  if (h->nx != 4) return false;
  if (h->flags & ILQR_FLAG_BACKWARD_THREAD_PER_TRAJ) return false;
  return true;
}

static int launch_backward(ilqr_batch* h, int mode) {
  if (!h->aos)
    if (int rc = materialise_records(h)) return rc;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (int rc = timer_begin(h, ILQR_STAGE_BACKWARD, &ev)) return rc;
  if (h->aos) {
    // the register-resident kernel, two (nx > 16) or more (nx <= 16) wavefronts per SIMD; ILQR_ROUTE_BACKWARD_LDS forces round
    // 1's LDS kernel -- the two give bit-identical results (tests/test_gpu_generic_backward.py)
    const bool fused = h->lq_fused && !h->lq_caller_records;  // cx, cu from the knot, the matrices from const_rec: D untouched
    if (!fused)
      if (int rc = ensure_records(h)) return rc;
    const double* crec = (fused || h->records_partial) ? h->const_rec : nullptr;
    const dim3 grid(h->B), block(64);
    const bool full = h->nu == WM && (h->nx == 16 || h->nx == 32);
#define ILQR_W3(NT_, FULL_, LQF_) hipLaunchKernelGGL((k_backward_w3<NT_, FULL_, LQF_>), grid, block, 0, h->stream, h->v, h->nx, h->nu, h->d_umin, h->d_umax, h->sp, mode, crec)
    if (h->env.backward_w1)
      hipLaunchKernelGGL(k_backward_w, grid, block, 0, h->stream, h->v, h->nx, h->nu, h->d_umin, h->d_umax, h->sp, mode, crec);
    else if (h->env.backward_w2 && h->nx > 16)
      hipLaunchKernelGGL(k_backward_w2<2>, grid, block, 0, h->stream, h->v, h->nx, h->nu, h->d_umin, h->d_umax, h->sp, mode, crec);
    else if (h->env.backward_w2)
      hipLaunchKernelGGL(k_backward_w2<1>, grid, block, 0, h->stream, h->v, h->nx, h->nu, h->d_umin, h->d_umax, h->sp, mode, crec);
    else if (h->nx > 16) {
      if (fused) { if (full) ILQR_W3(2, true, true); else ILQR_W3(2, false, true); }
      else { if (full) ILQR_W3(2, true, false); else ILQR_W3(2, false, false); }
    } else {
      if (fused) ILQR_W3(1, false, true); else ILQR_W3(1, false, false);
    }
#undef ILQR_W3
  } else if (use_quad_backward(h)) {
    dim3 grid(h->ntiles), block(64);  // one wavefront = one tile of 16 trajectories x 4 lanes
    if (int rc = with_model(h, [&](auto& v, auto& m, auto&) {
          hipLaunchKernelGGL((k_backward_q<std::decay_t<decltype(m)>>), grid, block, 0, h->stream, v, m, h->sp, mode);
          return 0;
        }))
      return rc;
  } else {
    dim3 grid(h->Bp / 64), block(64);
    if (int rc = with_model(h, [&](auto& v, auto& m, auto&) {
          hipLaunchKernelGGL((k_backward_t<std::decay_t<decltype(m)>>), grid, block, 0, h->stream, v, m, h->sp, mode);
          return 0;
        }))
      return rc;
  }
  HIPCHK(hipGetLastError());
  return timer_end(h, ILQR_STAGE_BACKWARD, ev);
}

// Which route ilqr_iterate takes.  All of them leave the same bits (tests/test_gpu_fused_sweep.py):
//   ntiles <= #CU, m = 1, no fixes     one persistent tile per CU, its backward pass as four matrix-core chains   k_solve_hex
//   ntiles <= #CU otherwise            one persistent 16-trajectory tile per CU            k_solve_tile<.., 1>
//   m = 1, no opt-in fixes, > 2 tiles per CU    64-trajectory wide tiles, one or two per CU   k_solve_wide
//   anything larger otherwise          persistent 16-trajectory tiles, two per CU (the dispatcher hands a CU its next
//                                      tile when one is through)                           k_solve_tile<.., 2>
//   ILQR_FLAG_STAGED                   one launch per stage: k_sweep_backward (records in the LDS ring, one block per CU
//                                      or the one-producer variant, two per CU) up to two tiles per CU, beyond that
//                                      k_derivatives + k_backward_q with the records in HBM
//   ILQR_FLAG_UNFUSED, AoS (generic) models   always the two-kernel route
// ilqr_desc.route (ILQR_ROUTE_TILE_PER_CU / TWO_TILES_PER_CU / WIDE_TILES) forces a variant for A/B runs and the bit-identity tests.
static int fused_variant(const ilqr_batch* h) {  // 0: two kernels, 1: one tile per CU, 2: two tiles per CU, 3: wide tiles (64 trajectories, one per CU), 4: one tile per CU, matrix-core chains
  if (!use_quad_backward(h) || h->aos || (h->flags & ILQR_FLAG_UNFUSED) || h->env.unfused) return 0;
  const bool staged = (h->flags & ILQR_FLAG_STAGED) || h->env.staged;
  const bool wide_ok = !staged && h->nu == 1 && h->sp.fixes == 0;  // wide tiles (kernels_wide.hpp): persistent route, m = 1, no opt-in fixes
  const int one_per_cu = (wide_ok && !h->env.quad_chain) ? 4 : 1;  // k_solve_hex (backward_hex.hpp) shares the wide tiles' conditions
  if (h->env.fused) return (h->env.fused == 3 && !wide_ok) ? 2 : (h->env.fused == 1 ? one_per_cu : h->env.fused);
  if (h->ntiles <= h->num_cus) return one_per_cu;
  // beyond two 16-trajectory tiles per CU: 64-trajectory wide tiles, the thread-per-trajectory chain (one per CU up to 64 #CU
  // trajectories -- a third tile per CU would be a second round of the two-per-CU kernel: 1.49 against 1.16-1.27 ms at
  // B = 8448 .. 14336 --, two per CU beyond)
  if (wide_ok && h->ntiles > 2 * h->num_cus) return 3;
  if (!staged) return 2;  // persistent tiles, two per CU, for ANY larger batch: the dispatcher hands a CU its next tile when one is through
  return (h->ntiles <= 2 * h->num_cus) ? 2 : 0;
}
static bool use_fused_sweep(const ilqr_batch* h) { return fused_variant(h) != 0; }
constexpr int kRingKbTwoBlocks = 60;
template <class V, class M, class MFD>
static void launch_sweep_backward_t(ilqr_batch* h, const V& v, const M& m, const MFD& fdm, int variant, int mode, int force, const int* ci) {
  if (variant == 2)
    hipLaunchKernelGGL((k_sweep_backward<M, 1, kRingKbTwoBlocks, MFD>), dim3(h->ntiles), dim3(64 * 2), 0, h->stream, v, m, fdm, h->sp, mode, force, ci);
  else
    hipLaunchKernelGGL((k_sweep_backward<M, kProducers, ILQR_RING_KB, MFD>), dim3(h->ntiles), dim3(64 * (1 + kProducers)), 0, h->stream, v, m, fdm, h->sp, mode, force, ci);
}
static int launch_sweep_backward(ilqr_batch* h, int mode, int force) {
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (int rc = timer_begin(h, ILQR_STAGE_BACKWARD, &ev)) return rc;
  const int variant = fused_variant(h);
  const int* ci = h->commit_pending ? h->commit_idx : nullptr;
  if (int rc = with_model(h, [&](auto& v, auto& m, auto& fdm) {
        launch_sweep_backward_t(h, v, m, fdm, variant, mode, force, ci);
        return 0;
      }))
    return rc;
  HIPCHK(hipGetLastError());
  h->commit_pending = false;  // the producers performed the copy on the way (see launch_derivatives)
  h->recs = ilqr_batch::REC_STALE;  // the records lived in LDS only
  return timer_end(h, ILQR_STAGE_BACKWARD, ev);
}

// selection + lambda schedule + termination; the copy of the accepted candidate is left pending
// (fused into the next derivative sweep, or flushed by flush_commit)
static int launch_accept(ilqr_batch* h) {
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (int rc = timer_begin(h, ILQR_STAGE_ACCEPT, &ev)) return rc;
  hipLaunchKernelGGL(k_accept<double>, dim3((h->Bp + 255) / 256), dim3(256), 0, h->stream, h->v, h->sp, h->commit_idx);  // (scalars only)
  HIPCHK(hipGetLastError());
  h->commit_pending = true;
  return timer_end(h, ILQR_STAGE_ACCEPT, ev);
}

// Whole iterations per tile in one persistent kernel (k_solve_tile): the one-block-per-CU regime of the fused
// kernel.  ILQR_FLAG_STAGED: per-stage launches instead (A/B runs, the bit-identity tests).
static bool use_persistent(const ilqr_batch* h) {
  if (h->aos || (h->flags & ILQR_FLAG_STAGED) || h->env.staged) return false;
  return fused_variant(h) != 0;
}
static AlphaSet line_search_alphas();
static int launch_solve_tiles(ilqr_batch* h, int n_iters) {
  std::pair<hipEvent_t, hipEvent_t> ev;
  HIPCHK(hipMemsetAsync(h->v.n_running, 0, sizeof(int), h->stream));
  if (int rc = timer_begin(h, ILQR_STAGE_SOLVE, &ev)) return rc;
  const AlphaSet al = line_search_alphas();
  const int pending = h->commit_pending ? 1 : 0;
  long long* ticks = h->profile ? h->phase_ticks : nullptr;
  const int occ = fused_variant(h);
  const int grid_tiles = (h->active_tiles > 0 && h->active_tiles < h->ntiles) ? h->active_tiles : h->ntiles;  // (the rest hold finished trajectories only)
  if (int rc = with_model(h, [&](auto& v, auto& m, auto& fdm) {
        using MM = std::decay_t<decltype(m)>;
        using MF = std::decay_t<decltype(fdm)>;
        if (occ == 3) {
          if constexpr (MM::NU == 1)
          {
            if (h->env.wide_occ == 1 || (h->env.wide_occ == 0 && (grid_tiles + 3) / 4 <= h->num_cus))
              hipLaunchKernelGGL((k_solve_wide<MM, MF, 1>), dim3((grid_tiles + 3) / 4), dim3(512), 0, h->stream, v, m, fdm, al, h->sp, n_iters, h->sp.fixed_work, h->commit_idx, pending, ticks);
            else
              hipLaunchKernelGGL((k_solve_wide<MM, MF, 2>), dim3((grid_tiles + 3) / 4), dim3(256), 0, h->stream, v, m, fdm, al, h->sp, n_iters, h->sp.fixed_work, h->commit_idx, pending, ticks);
          }
        } else if (occ == 4) {
          if constexpr (MM::NU == 1)
            hipLaunchKernelGGL((k_solve_hex<MM, MF>), dim3(grid_tiles), dim3(512), 0, h->stream, v, m, fdm, al, h->sp, n_iters, h->sp.fixed_work, h->commit_idx, pending, ticks);
        } else if (occ == 1)
          hipLaunchKernelGGL((k_solve_tile<MM, MF, 1>), dim3(grid_tiles), dim3(256), 0, h->stream, v, m, fdm, al, h->sp, n_iters, h->sp.fixed_work, h->commit_idx, pending, ticks);
        else
          hipLaunchKernelGGL((k_solve_tile<MM, MF, 2>), dim3(grid_tiles), dim3(256), 0, h->stream, v, m, fdm, al, h->sp, n_iters, h->sp.fixed_work, h->commit_idx, pending, ticks);
        return 0;
      }))
    return rc;
  HIPCHK(hipGetLastError());
  h->commit_pending = (occ != 4);   // the last iteration's accepts (flushed by the caller); k_solve_hex commits every iteration's itself
  h->recs = ilqr_batch::REC_STALE;
  return timer_end(h, ILQR_STAGE_SOLVE, ev);
}

static AlphaSet line_search_alphas() {
  AlphaSet a;
  for (int i = 0; i < NALPHA; i++) a.a[i] = kAlphaHost[i];
  return a;
}

static int do_rollout_candidates(ilqr_batch* h, int mode) {
  if (int rc = launch_rollout(h, true, true, line_search_alphas(), NALPHA, h->v.cost_c, mode)) return rc;
  h->cands_valid = true;
  return 0;
}

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
  h->env.staged = false;
  h->env.unfused = false;
  h->env.backward_w1 = (d->route & ILQR_ROUTE_BACKWARD_LDS) != 0;
  h->env.backward_w2 = (d->route & ILQR_ROUTE_BACKWARD_W2) != 0;
  h->env.lq_thread_rollout = (d->route & ILQR_ROUTE_LQ_THREAD_ROLLOUT) != 0;
  h->env.full_records = (d->route & ILQR_ROUTE_FULL_RECORDS) != 0;
  h->env.no_compaction = (d->route & ILQR_ROUTE_NO_COMPACTION) != 0;
  h->env.quad_chain = (d->route & ILQR_ROUTE_QUAD_CHAIN) != 0;
  h->env.fused = d->route & 3;
  h->env.wide_occ = (d->route & ILQR_ROUTE_WIDE_ONE_PER_CU) ? 1 : (d->route & ILQR_ROUTE_WIDE_TWO_PER_CU) ? 2 : 0;
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
    if (!kUserTiled) {  // not a tiled nx = 4 shape: the generic kernels (fp64), trajectory-contiguous layout like the LQ model's
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
    if (d->model == ILQR_MODEL_LQ && !h->env.lq_thread_rollout && !(d->route & ILQR_ROUTE_LQ_RECOMMIT)) {
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
  rc |= dev_alloc_real(h, &v.xs, nt * T1 * nx * TW);
  rc |= dev_alloc_real(h, &v.us, nt * T * nu * TW);
  rc |= dev_alloc_real(h, &v.kff, nt * T * nu * TW);
  rc |= dev_alloc_real(h, &v.Kfb, nt * T * nu * nx * TW);
  v.D = nullptr;  // on first use (ensure_records)
  v.nch = h->T / CT + 1;
  // (one plane more than there are alphas: where the rollout lanes without a rollout of their own put their stores, rollout.hpp)
  rc |= dev_alloc_real(h, &v.cand_u, (size_t)(NALPHA + 1) * nt * T * nu * TW);
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
  // generic handles: ILQR_FLAG_REFERENCE_FIXES on the models with a device twin (their rollouts clamp, their box-QP reports a failed factorisation); the
  // host-evaluated route's rollouts belong to the caller, and lambda on Vxx would be two more products per step of the matrix-core kernels
  if ((h->sp.fixes & 4) && h->aos) return fail(ILQR_ERR_UNSUPPORTED, "ILQR_FLAG_REGULARIZE_VXX is implemented for the nx = 4 device models");
  if (h->sp.fixes && h->model == ILQR_MODEL_HOST) return fail(ILQR_ERR_UNSUPPORTED, "ILQR_FLAG_REFERENCE_FIXES on a host-evaluated model: its rollouts are the caller's (clamp there); the flag is implemented for the models with a device twin");

  hipLaunchKernelGGL(k_reset_state<double>, dim3((h->Bp + 255) / 256), dim3(256), 0, h->stream, h->v, h->params.lambda_init,
                     h->params.dlambda_init);
  HIPCHK(hipGetLastError());
  h->lq_fused = h->model == ILQR_MODEL_LQ && v.analytic && !h->env.full_records && !h->env.backward_w1 && !h->env.backward_w2;
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
  const bool compacting = persistent && h->ntiles > h->num_cus && !(h->sp.fixed_work) && !h->env.no_compaction;
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
        hipLaunchKernelGGL((k_unpack_cand<std::decay_t<decltype(m)>>), grid, block, 0, h->stream, v, m, a, dxs, dus);
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

// ---- shard groups (include/ilqr_amd.h) -------------------------------------------------------
}  // extern "C"
#include <dlfcn.h>
// RCCL is loaded at run time (dlopen below) and only when a group spans devices, so its header must not be a build dependency:
// the real declarations where the header exists, otherwise the six entry points and three types this file uses (nccl.h's ABI)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;  // ncclFloat64
#endif
namespace {
struct RcclApi {  // librccl.so, loaded on first use: a single-GPU user of the library never maps it
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load() {
    if (lib) return true;
    lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return false;
    CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd && GetErrorString;
  }
};
RcclApi g_rccl;
}  // namespace
struct ilqr_group {
  std::vector<ilqr_batch*> shards;
  bool rccl = false;
  int per = 0;                       // padded shard length of the all-gather (the largest B)
  std::vector<ncclComm_t> comms;     // one per shard (= per device), in shard order
  std::vector<double*> send, recv;   // per device: [per], [n_shards * per]
};
#define NCCLCHK(call)                                                                              \
  do {                                                                                             \
    ncclResult_t r_ = (call);                                                                      \
    if (r_ != ncclSuccess) return fail(ILQR_ERR_HIP, "%s: %s", #call, g_rccl.GetErrorString(r_)); \
  } while (0)
extern "C" {
int ilqr_group_create(ilqr_batch* const* shards, int n_shards, int flags, ilqr_group** out) {
  if (!shards || !out || n_shards < 1) return fail(ILQR_ERR_INVALID, "ilqr_group_create: null argument / no shards");
  for (int i = 0; i < n_shards; i++)
    if (!shards[i]) return fail(ILQR_ERR_INVALID, "ilqr_group_create: shard %d is null", i);
  ilqr_group* g = new ilqr_group();
  g->shards.assign(shards, shards + n_shards);
  bool distinct = true;
  for (int i = 0; i < n_shards; i++)
    for (int j = 0; j < i; j++) distinct = distinct && shards[i]->device != shards[j]->device;
  for (int i = 0; i < n_shards; i++) g->per = std::max(g->per, shards[i]->B);
  g->rccl = distinct && (n_shards > 1 || (flags & 1));
  if (g->rccl) {
    if (!g_rccl.load()) {
      delete g;
      return fail(ILQR_ERR_UNSUPPORTED, "shards on %d devices need librccl.so for their gather: %s", n_shards, dlerror());
    }
    std::vector<int> devs(n_shards);
    for (int i = 0; i < n_shards; i++) devs[i] = shards[i]->device;
    g->comms.resize(n_shards);
    ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n_shards, devs.data());
    if (r != ncclSuccess) {
      g->comms.clear();
      delete g;
      return fail(ILQR_ERR_HIP, "ncclCommInitAll over %d devices: %s", n_shards, g_rccl.GetErrorString(r));
    }
    g->send.assign(n_shards, nullptr);
    g->recv.assign(n_shards, nullptr);
    for (int i = 0; i < n_shards; i++) {
      if (hipSetDevice(devs[i]) != hipSuccess || hipMalloc((void**)&g->send[i], (size_t)g->per * sizeof(double)) != hipSuccess ||
          hipMalloc((void**)&g->recv[i], (size_t)n_shards * g->per * sizeof(double)) != hipSuccess ||
          hipMemset(g->send[i], 0, (size_t)g->per * sizeof(double)) != hipSuccess) {
        ilqr_group_destroy(g);
        return fail(ILQR_ERR_HIP, "ilqr_group_create: device buffers of shard %d", i);
      }
    }
  }
  *out = g;
  return 0;
}
void ilqr_group_destroy(ilqr_group* g) {
  if (!g) return;
  for (size_t i = 0; i < g->comms.size(); i++) {
    (void)hipSetDevice(g->shards[i]->device);
    if (i < g->send.size() && g->send[i]) (void)hipFree(g->send[i]);
    if (i < g->recv.size() && g->recv[i]) (void)hipFree(g->recv[i]);
    if (g->comms[i]) (void)g_rccl.CommDestroy(g->comms[i]);
  }
  delete g;
}
int ilqr_group_uses_rccl(ilqr_group* g, int* n_ranks) {
  if (!g) return 0;
  if (n_ranks) *n_ranks = (int)g->comms.size();
  return g->rccl ? 1 : 0;
}
int ilqr_group_gather_costs(ilqr_group* g, double* cost_out) {
  if (!g || !cost_out) return fail(ILQR_ERR_INVALID, "null argument");
  const int n = (int)g->shards.size();
  if (!g->rccl) {  // shards share a device: plain copies, shard by shard
    size_t off = 0;
    for (int i = 0; i < n; i++) {
      if (int rc = ilqr_get_cost(g->shards[i], cost_out + off)) return rc;
      off += (size_t)g->shards[i]->B;
    }
    return 0;
  }
  // every shard's costs into its device's send buffer (on the shard's stream), then ONE all-gather over the devices' links
  for (int i = 0; i < n; i++)
    if (int rc = ilqr_copy_cost_to_device(g->shards[i], g->send[i])) return rc;
  NCCLCHK(g_rccl.GroupStart());
  for (int i = 0; i < n; i++) {  // (a failure inside the group closes it before returning: the calling thread must not be left inside an open NCCL group)
    const hipError_t he = hipSetDevice(g->shards[i]->device);
    const ncclResult_t nr = (he == hipSuccess) ? g_rccl.AllGather(g->send[i], g->recv[i], (size_t)g->per, ncclDouble, g->comms[i], g->shards[i]->stream)  // (stream order: after the copy)
                                               : ncclSuccess;
    if (he != hipSuccess || nr != ncclSuccess) {
      (void)g_rccl.GroupEnd();
      return he != hipSuccess ? fail(ILQR_ERR_HIP, "hipSetDevice(%d) inside the gather: %s", g->shards[i]->device, hipGetErrorString(he))
                              : fail(ILQR_ERR_HIP, "ncclAllGather of shard %d: %s", i, g_rccl.GetErrorString(nr));
    }
  }
  NCCLCHK(g_rccl.GroupEnd());
  std::vector<double> all((size_t)n * g->per);
  ilqr_batch* h0 = g->shards[0];
  HIPCHK(hipSetDevice(h0->device));
  HIPCHK(hipMemcpyAsync(all.data(), g->recv[0], all.size() * sizeof(double), hipMemcpyDeviceToHost, h0->stream));
  for (int i = 0; i < n; i++) {
    HIPCHK(hipSetDevice(g->shards[i]->device));
    HIPCHK(hipStreamSynchronize(g->shards[i]->stream));
  }
  size_t off = 0;
  for (int i = 0; i < n; i++) {  // drop the padding of ragged shards
    std::copy(all.begin() + (size_t)i * g->per, all.begin() + (size_t)i * g->per + g->shards[i]->B, cost_out + off);
    off += (size_t)g->shards[i]->B;
  }
  return 0;
}

// ---- measurement -----------------------------------------------------------------------------
int ilqr_profile_enable(ilqr_batch* h, int enable) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  h->profile = enable != 0;
  return 0;
}
int ilqr_profile_reset(ilqr_batch* h) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  if (int rc = timers_drain(h)) return rc;
  for (auto& t : h->timers) {
    t.ms = 0;
    t.launches = 0;
  }
  HIPCHK(hipMemsetAsync(h->phase_ticks, 0, 5 * (size_t)h->ntiles * sizeof(long long), h->stream));
  return 0;
}
int ilqr_profile_read(ilqr_batch* h, double ms_out[ILQR_NUM_STAGES], int launches_out[ILQR_NUM_STAGES]) {
  if (!h) return fail(ILQR_ERR_INVALID, "null handle");
  if (int rc = timers_drain(h)) return rc;
  double ms[ILQR_NUM_STAGES];
  int ln[ILQR_NUM_STAGES];
  for (int s = 0; s < ILQR_NUM_STAGES; s++) {
    ms[s] = h->timers[s].ms;
    ln[s] = h->timers[s].launches;
  }
  if (h->timers[ILQR_STAGE_SOLVE].launches > 0) {  // the persistent kernel's own phase clocks: mean over tiles
    std::vector<long long> tk(5 * (size_t)h->ntiles);
    HIPCHK(hipMemcpyAsync(tk.data(), h->phase_ticks, tk.size() * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double sweep = 0, roll = 0, its = 0;
    for (int t = 0; t < h->ntiles; t++) {
      sweep += (double)tk[5 * t];
      roll += (double)tk[5 * t + 1];
      its += (double)tk[5 * t + 2];
    }
    const double to_ms = 1.0 / h->wall_clock_khz / h->ntiles;  // ticks -> ms, mean over tiles
    ms[ILQR_STAGE_BACKWARD] += sweep * to_ms;
    ms[ILQR_STAGE_ROLLOUT] += roll * to_ms;
    ln[ILQR_STAGE_BACKWARD] += (int)(its / h->ntiles + 0.5);
    ln[ILQR_STAGE_ROLLOUT] += (int)(its / h->ntiles + 0.5);
  }
  for (int s = 0; s < ILQR_NUM_STAGES; s++) {
    if (ms_out) ms_out[s] = ms[s];
    if (launches_out) launches_out[s] = ln[s];
  }
  return 0;
}
int ilqr_profile_shader_clock(ilqr_batch* h, double* mhz_out) {
  if (!h || !mhz_out) return fail(ILQR_ERR_INVALID, "null argument");
  std::vector<long long> tk(5 * (size_t)h->ntiles);
  HIPCHK(hipMemcpyAsync(tk.data(), h->phase_ticks, tk.size() * sizeof(long long), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  double cyc = 0, wall = 0;
  for (int t = 0; t < h->ntiles; t++) {
    cyc += (double)tk[5 * t + 3];
    wall += (double)tk[5 * t + 4];
  }
  *mhz_out = (wall > 0) ? cyc / wall * h->wall_clock_khz * 1e-3 : 0.0;  // cycles per tick x ticks per ms / 1000
  return 0;
}
const char* ilqr_stage_kernel_name(ilqr_batch* h, int stage) {
  switch (stage) {
    case ILQR_STAGE_DERIVATIVES: return (h && h->aos) ? (h->lq_fused ? "" : (h->v.analytic && h->model == ILQR_MODEL_LQ) ? "k_analytic_lq" : "k_derivatives_g") : "k_derivatives";
    case ILQR_STAGE_BACKWARD:
      if (h && h->aos) return h->env.backward_w1 ? "k_backward_w" : h->env.backward_w2 ? "k_backward_w2" : "k_backward_w3";
      if (h && use_fused_sweep(h)) return "k_sweep_backward";  // what ilqr_iterate launches
      return (h && use_quad_backward(h)) ? "k_backward_q" : "k_backward_t";
    case ILQR_STAGE_ROLLOUT: return (h && h->aos) ? ((h->env.lq_thread_rollout || h->model != ILQR_MODEL_LQ) ? "k_rollout_g" : "k_rollout_lq") : "k_rollout";
    case ILQR_STAGE_ACCEPT: return "k_accept";
    case ILQR_STAGE_SOLVE: return (h && use_persistent(h)) ? (fused_variant(h) == 1 ? "k_solve_tile" : fused_variant(h) == 3 ? "k_solve_wide" : fused_variant(h) == 4 ? "k_solve_hex" : "k_solve_tile<2>") : "";
    default: return "";
  }
}

}  // extern "C"
