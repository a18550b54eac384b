// profile.hpp -- the measurement entry points of the C ABI: per-stage HIP-event timers, the persistent kernels' own phase clocks, the
// kernel a stage launches on this handle.  Included once, by capi.hip.
#pragma once
#include "launch.hpp"

// ---- measurement -----------------------------------------------------------------------------
extern "C" {
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
    case ILQR_STAGE_DERIVATIVES: return (h && h->aos) ? (h->lq_fused ? "" : (h->v.analytic && h->model == ILQR_MODEL_LQ) ? "k_analytic_lq" : (h->model == ILQR_MODEL_LQ && !h->route.lq_dense_fd) ? "k_derivatives_lq" : "k_derivatives_g") : "k_derivatives";
    case ILQR_STAGE_BACKWARD:
      if (h && h->aos) return h->route.backward_w2 ? "k_backward_w2" : "k_backward_w3";
      if (h && use_fused_sweep(h)) return "k_sweep_backward";  // what ilqr_iterate launches
      return (h && use_quad_backward(h)) ? "k_backward_q" : "k_backward_t";
    case ILQR_STAGE_ROLLOUT: return (h && h->aos) ? ((h->route.lq_thread_rollout || h->model != ILQR_MODEL_LQ) ? "k_rollout_g" : "k_rollout_lq") : "k_rollout";
    case ILQR_STAGE_ACCEPT: return "k_accept";
    case ILQR_STAGE_SOLVE: return (h && use_persistent(h)) ? (fused_variant(h) == 1 ? "k_solve_tile" : fused_variant(h) == 3 ? (h->nu == 2 ? "k_solve_wide2" : "k_solve_wide") : fused_variant(h) == 4 ? "k_solve_hex" : "k_solve_tile<2>") : "";
    default: return "";
  }
}

}  // extern "C"

