// handle.hpp -- the opaque handle behind the C ABI (ilqr_batch: device memory of one batch, its stream, its route choices, its stage
// timers) and the helpers every entry point uses: error plumbing, allocation, host <-> device layout conversion, the record
// array's states.  Included once, by capi.hip.
#pragma once
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
  bool cands_grouped = false;  // ... and lie in k_solve_hex's grouped layout (rollout.hpp: CANDT) instead of one element per trajectory
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
    bool staged = false, unfused = false, backward_w2 = false, lq_dense_fd = false, lq_thread_rollout = false, full_records = false, no_compaction = false, quad_chain = false;
    int fused = 0;  // 0 = by batch size
    int wide_occ = 0;  // wide tiles per CU: 0 = by batch size
  } route;
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

