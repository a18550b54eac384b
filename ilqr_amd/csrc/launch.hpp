// launch.hpp -- kernel launchers of the C ABI: dispatch on the handle's device model and arithmetic, and the ROUTE logic (which of
// several equivalent kernels a handle's batch size / flags / ilqr_desc.route select: DESIGN.md 3.2).  Included once, by capi.hip.
#pragma once
#include "handle.hpp"

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
  if constexpr (kUserGeneric)
    if (h->model == ILQR_MODEL_USER) return f(h->user_g);
#endif
  return fail(ILQR_ERR_UNSUPPORTED, "model %d has no generic device kernels", h->model);
}
// generic path (generic.hpp): what = RG_INIT / RG_SEARCH / RG_COMMIT.  The LQ model rolls out on the
// matrix cores (k_rollout_lq, one wavefront per trajectory); ILQR_ROUTE_LQ_THREAD_ROLLOUT selects the
// generic thread-per-rollout kernel (same results bit for bit; kept as the cross-check and as the
// template for device models without matrix structure).
// Does the handle's search kernel also accept and commit (k_rollout_lq<RG_SEARCH, true>)?  The LQ model's matrix-core rollout with candidate buffers.
static bool lq_search_accepts(const ilqr_batch* h) { return h->model == ILQR_MODEL_LQ && !h->route.lq_thread_rollout && h->v.cand_x != nullptr; }
template <class M>
static int launch_rollout_g(ilqr_batch* h, const M& m, int what, const AlphaSet& al, double* cost_out, int mode, int write_cost, bool with_accept = false) {
  if constexpr (std::is_same<M, LqModel>::value)
  if (!h->route.lq_thread_rollout) {
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
  if (cand) h->cands_grouped = false;  // (the stage kernels write the alpha planes)
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
      const int what = h->route.full_records ? 0 : 1;  // (A/B runs and the bit-identity test)
      const int chunk = (what == 1) ? 4 * kAnalyticChunk : kAnalyticChunk;
      const int nchunk = (h->T + 1 + chunk - 1) / chunk;
      hipLaunchKernelGGL(k_analytic_lq, dim3(h->B * nchunk), dim3(64), 0, h->stream, h->v, h->lq, force, what, h->const_rec, chunk);
      h->records_partial = (what == 1);
    } else {
      if (h->model == ILQR_MODEL_LQ && !h->route.lq_dense_fd) {
        // the LQ twin: every perturbed point of the knots t < T evaluated by what moved (k_derivatives_lq), knot T by the generic sweep
        const int nchunk = (h->T + kLqKnotsPerWave - 1) / kLqKnotsPerWave;
        hipLaunchKernelGGL(k_derivatives_lq, dim3(h->B * nchunk), dim3(64), 0, h->stream, h->v, h->lq, force);
        hipLaunchKernelGGL((k_derivatives_g<LqModel>), dim3(h->B), dim3(64), 0, h->stream, h->v, h->lq, force, h->T);
      } else if (int rc = with_generic_model(h, [&](auto& m) {
            hipLaunchKernelGGL((k_derivatives_g<std::decay_t<decltype(m)>>), dim3(h->B * (h->T + 1)), dim3(64), 0, h->stream, h->v, m, force, -1);
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
    // the register-resident kernels, two (nx > 16) or more (nx <= 16) wavefronts per SIMD: k_backward_w3, or with ILQR_ROUTE_BACKWARD_W2 the
    // literal-order k_backward_w2 (round 1's LDS kernel k_backward_w, whose bits k_backward_w2 reproduces, was retired in ABI 5)
    const bool fused = h->lq_fused && !h->lq_caller_records;  // cx, cu from the knot, the matrices from const_rec: D untouched
    if (!fused)
      if (int rc = ensure_records(h)) return rc;
    const double* crec = (fused || h->records_partial) ? h->const_rec : nullptr;
    const dim3 grid(h->B), block(64);
    const bool full = h->nu == WM && (h->nx == 16 || h->nx == 32);
#define ILQR_W3(NT_, FULL_, LQF_) hipLaunchKernelGGL((k_backward_w3<NT_, FULL_, LQF_>), grid, block, 0, h->stream, h->v, h->nx, h->nu, h->d_umin, h->d_umax, h->sp, mode, crec)
    if (h->route.backward_w2 && h->nx > 16)
      hipLaunchKernelGGL(k_backward_w2<2>, grid, block, 0, h->stream, h->v, h->nx, h->nu, h->d_umin, h->d_umax, h->sp, mode, crec);
    else if (h->route.backward_w2)
      hipLaunchKernelGGL(k_backward_w2<1>, grid, block, 0, h->stream, h->v, h->nx, h->nu, h->d_umin, h->d_umax, h->sp, mode, crec);
    else if (h->sp.fixes & 4) {  // ILQR_FLAG_REGULARIZE_VXX: the bounds-checked instantiations on whole records (ilqr_create keeps lq_fused off)
      if (h->nx > 16)
        hipLaunchKernelGGL((k_backward_w3<2, false, false, true>), grid, block, 0, h->stream, h->v, h->nx, h->nu, h->d_umin, h->d_umax, h->sp, mode, crec);
      else
        hipLaunchKernelGGL((k_backward_w3<1, false, false, true>), grid, block, 0, h->stream, h->v, h->nx, h->nu, h->d_umin, h->d_umax, h->sp, mode, crec);
    } else if (h->nx > 16) {
      if (fused) { if (full) ILQR_W3(2, true, true); else ILQR_W3(2, false, true); }
      else { if (full) ILQR_W3(2, true, false); else ILQR_W3(2, false, false); }
    } else {
      if (fused) ILQR_W3(1, false, true); else ILQR_W3(1, false, false);
    }
#undef ILQR_W3
  } else if (use_quad_backward(h)) {
    dim3 grid(h->ntiles), block(64);  // one wavefront = one tile of 16 trajectories x 4 lanes
    if (int rc = with_model(h, [&](auto& v, auto& m, auto&) {
          if constexpr (std::decay_t<decltype(m)>::NX == 4)
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
  if (!use_quad_backward(h) || h->aos || (h->flags & ILQR_FLAG_UNFUSED) || h->route.unfused) return 0;
  const bool staged = (h->flags & ILQR_FLAG_STAGED) || h->route.staged;
  const bool wide_ok = !staged && h->nu <= 2 && h->sp.fixes == 0;  // wide tiles (kernels_wide.hpp, kernels_wide2.hpp): persistent route, m <= 2, no opt-in fixes
  const int one_per_cu = (wide_ok && h->nu == 1 && !h->route.quad_chain) ? 4 : 1;  // k_solve_hex (backward_hex.hpp): the wide tiles' conditions and m = 1
  if (h->route.fused) return (h->route.fused == 3 && !wide_ok) ? 2 : (h->route.fused == 1 ? one_per_cu : h->route.fused);
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
        if constexpr (std::decay_t<decltype(m)>::NX == 4) launch_sweep_backward_t(h, v, m, fdm, variant, mode, force, ci);
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
  if (h->aos || (h->flags & ILQR_FLAG_STAGED) || h->route.staged) return false;
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
        if constexpr (MM::NX != 4) {
          return fail(ILQR_ERR_STATE, "persistent tiles are nx = 4 kernels");
        } else
        if (occ == 3) {
          if constexpr (MM::NU == 1)
          {
            if (h->route.wide_occ == 1 || (h->route.wide_occ == 0 && (grid_tiles + 3) / 4 <= h->num_cus))
              hipLaunchKernelGGL((k_solve_wide<MM, MF, 1>), dim3((grid_tiles + 3) / 4), dim3(512), 0, h->stream, v, m, fdm, al, h->sp, n_iters, h->sp.fixed_work, h->commit_idx, pending, ticks);
            else
              hipLaunchKernelGGL((k_solve_wide<MM, MF, 2>), dim3((grid_tiles + 3) / 4), dim3(256), 0, h->stream, v, m, fdm, al, h->sp, n_iters, h->sp.fixed_work, h->commit_idx, pending, ticks);
          }
          else if constexpr (MM::NU == 2)
            hipLaunchKernelGGL((k_solve_wide2<MM, MF>), dim3((grid_tiles + 3) / 4), dim3(256), 0, h->stream, v, m, fdm, al, h->sp, n_iters, h->sp.fixed_work, h->commit_idx, pending, ticks);
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
  h->cands_grouped = (occ == 4) && kHexCandT;  // (what ilqr_get_candidate finds in the buffers)
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

