// solve_tile.hpp -- the LDS ring between producer wavefronts and the backward wavefront, the fused sweep + backward pass of a
// tile (k_sweep_backward), whole iterations of a tile in one persistent kernel (k_solve_tile, one or two tiles per CU), and
// the commit of an accepted candidate.
#pragma once
#include "backward_quad.hpp"

namespace ilqr {

#ifndef ILQR_PRODUCERS
#define ILQR_PRODUCERS 3
#endif
constexpr int kProducers = ILQR_PRODUCERS;
#ifndef ILQR_LEAD_ROUNDS
#define ILQR_LEAD_ROUNDS 1
#endif
// LDS of one tile's sweep + backward pass
template <class real, int NX, int NU, int kProd, int RING_KB, int PAD = 0>
struct SweepShared {
  using RS = RingSlot<NX, NU, real, RING_KB, PAD>;
  double steps[104];                  // backtracking step sizes, in the chain's arithmetic (double for every handle) (per-lane indexed -> LDS, not constant cache)
  alignas(16) real ring[RS::SLOTS * RS::ELEMS];   // knot with running index G lives in slot G % SLOTS
  int rounds_done[kProd];             // rounds (counted across passes) whose records are in the ring
  int consumer_at;                    // running index of the knot the backward pass waits for (everything below is consumed)
  int passes_started;                 // backward passes begun; -1 once the tile's backward wavefront is through
  unsigned long long pass_lanes;      // exec mask of the backward wavefront in the current pass (bit 4 l = trajectory l)
};

// The consumer side of the ring (see NoGate).  Knots are numbered by a RUNNING index G = pass * N + (T - t),
// N = knots per pass rounded up to whole producer rounds: passes follow each other seamlessly in the ring.
// KPP: knots per producer wavefront and round (4 x 16 trajectories for a tile; 1 x 64 for a wide tile, kernels_wide.hpp)
template <class SH, int kProd, int KPP = 4>
struct RingGate {
  static constexpr bool kRing = true;
  static constexpr int kKnotsPerRound = KPP * kProd;
  SH& sh;
  const int T, nrounds, N;
  int pass = -1, have = 0;
  // wait(t) is called for t = T, T-1, T-2, ... within a pass (the backward pass prefetches in that order), so the
  // running index of the knot and its ring slot are carried along instead of recomputed (a multiply-high modulo and
  // half a dozen scalar instructions per step on the backward wavefront's chain)
  int g_next = 0, slot_next = 0, slot_cur = 0;
  __device__ __forceinline__ RingGate(SH& s, int T_) : sh(s), T(T_), nrounds((T_ + 1 + kKnotsPerRound - 1) / kKnotsPerRound), N(nrounds * kKnotsPerRound) {}
  __device__ __forceinline__ void begin_pass() {  // wave-uniform among the lanes still in the pass loop
    pass++;
    have = pass * N;
    g_next = pass * N;
    slot_next = g_next % SH::RS::SLOTS;
    __hip_atomic_store(&sh.pass_lanes, __ballot(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&sh.passes_started, pass + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ int slot(int) const { return slot_cur; }  // of the knot last waited for
  __device__ __forceinline__ void wait(int) {
    const int G = g_next++;
    slot_cur = slot_next;
    slot_next = (slot_next + 1 == SH::RS::SLOTS) ? 0 : slot_next + 1;
    if (G < have) return;
    const int j = G - pass * N;
    const int round = pass * nrounds + j / kKnotsPerRound, w = (j % kKnotsPerRound) / KPP;
    __hip_atomic_store(&sh.consumer_at, G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(&sh.rounds_done[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= round) {
      __builtin_amdgcn_s_sleep(2);
    }
    have = pass * N + (j / kKnotsPerRound) * kKnotsPerRound + (w + 1) * KPP;
  }
  __device__ __forceinline__ void finish() {  // releases the producers (also from a pass abandoned half way)
    __hip_atomic_store(&sh.consumer_at, 0x3fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&sh.passes_started, -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
};

// STEP 1 + STEP 2 of one iteration for ONE tile, run by a block of 1 + kProd wavefronts (k_sweep_backward is
// this and nothing else).  The quad backward pass keeps a single wavefront per tile busy with one long dependent
// chain, i.e. one of the four SIMDs of a CU; the finite-difference sweep is independent per knot.  So wavefront
// 0 runs backward_quad, wavefronts 1..kProd are PRODUCERS that compute the derivative records of the tile's
// knots in descending t (4 knots x 16 trajectories per wavefront and round) into an LDS ring, perform the
// pending commit of the accepted candidate on the way (derivatives_of_knot, first pass only), and publish their
// progress in LDS.  The consumer follows a few hundred cycles behind the first round and never waits again (a
// producer round of 4 time steps costs about as much as ONE backward step): ds_read, no HBM round trip, no
// vmcnt wait in its loop.  The records never reach HBM: a lambda-retry pass (ilqr_core.cpp:136-150) makes the
// producers sweep again, for the trajectories that retry; whoever else wants records (getters, stage calls)
// has k_derivatives compute them.  Workgroup-scope release/acquire is all the ordering needed.
// role: what this wavefront does -- 0 the backward pass, 1..kProd producer role-1, anything else nothing (default: by wavefront index)
template <class M, int kProd, int RING_KB, class MFD, class SH, bool ONESET = false>
__device__ __forceinline__ void sweep_backward_tile(const BatchViewT<typename M::real>& v, const M& model, const MFD& fdm, const SolverParams& sp,
                                                    int mode, int force, const int* __restrict__ commit_idx, int tile, SH& sh, int role = -1) {
  using real = typename M::real;
  constexpr int kKnotsPerRound = 4 * kProd;                      // 4 knots per producer wavefront
  constexpr int kLeadKnots = ILQR_LEAD_ROUNDS * kKnotsPerRound;  // producers stay at most this far ahead of the consumer
  using RS = typename SH::RS;
  static_assert(RS::SLOTS >= kLeadKnots + 4, "the ring must hold the producers' lead plus the four knots in production");
  if (threadIdx.x < kProd) sh.rounds_done[threadIdx.x] = 0;
  if (threadIdx.x == kProd) {
    sh.consumer_at = 0;
    sh.passes_started = 0;
  }
  __syncthreads();
  const int wave = (role >= 0) ? role : (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int T = v.T;
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    RingGate<SH, kProd> gate(sh, T);
    if (sp.fixes)
      backward_quad<M, decltype(gate), RING_KB, true, ONESET>(v, model, sp, mode, tile, lane, sh.steps, gate, sh.ring);
    else
      backward_quad<M, decltype(gate), RING_KB, false, ONESET>(v, model, sp, mode, tile, lane, sh.steps, gate, sh.ring);
    gate.finish();
    __builtin_amdgcn_s_setprio(0);
  } else {
    // Producers pace themselves to the consumer (a bounded lead is what keeps a ring slot from being
    // overwritten before it is read; running flat out they also took issue slots from nobody but saturated
    // the CU's store path when the records still went to HBM).  A round is published as soon as its LDS
    // writes are done.
    const int w = wave - 1;
    const int l = lane & (TW - 1), sub = lane >> 4;
    const int nrounds = (T + 1 + kKnotsPerRound - 1) / kKnotsPerRound, N = nrounds * kKnotsPerRound;
    if (w < kProd)  // (a block may have more wavefronts than this phase uses: k_solve_tile's fourth one with two producers)
    for (int pass = 0;; pass++) {
      unsigned long long lanes = ~0ull;
      if (pass > 0) {  // a retry pass exists only if the backward wavefront starts one
        int started;
        while ((started = __hip_atomic_load(&sh.passes_started, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= 0 && started <= pass)
          __builtin_amdgcn_s_sleep(8);
        if (started < 0) break;
        lanes = __hip_atomic_load(&sh.pass_lanes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      const bool mine = (lanes >> (4 * l)) & 1ull;  // does trajectory l take part in this pass?
      for (int r = 0; r < nrounds; r++) {
        const int j0 = r * kKnotsPerRound + w * 4, G0 = pass * N + j0;
        while (G0 > __hip_atomic_load(&sh.consumer_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + kLeadKnots)
          __builtin_amdgcn_s_sleep(8);
        // Has the backward wavefront left this pass behind (abandoned it at a failed box-QP, or is through)?
        // Then its remaining records are of no use -- but the first pass still owes the commit of every knot.
        const int started = __hip_atomic_load(&sh.passes_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const bool moved_on = (started < 0) | (started > pass + 1);
        if (moved_on && (pass > 0 || commit_idx == nullptr)) break;
        const int t = T - (j0 + sub);
        if (t >= 0 && (pass == 0 || mine))
          derivatives_of_knot<M, true, MFD>(v, model, fdm, force, pass == 0 ? commit_idx : nullptr, tile, t, l,
                                            sh.ring + ((G0 + sub) % RS::SLOTS) * RS::ELEMS + l * 2, !moved_on);
        // LDS operations of a wavefront complete in order: once its writes are done the round is visible
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        if (lane == 0) __hip_atomic_store(&sh.rounds_done[w], pass * nrounds + r + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
}

//   grid = ntiles, block = 64 * (1 + kProd), LDS ~150 KB (one block per CU)
// Two instantiations are shipped: <3 producers, 150 KB ring> = one block per CU, for batches of up to
// 16 x #CU trajectories, and <1 producer, 60 KB ring> = two blocks (four wavefronts) per CU for up to
// twice that -- one producer cannot quite feed a backward wavefront (0.66 instead of 0.55 ms per tile
// at T = 499), but two tiles per CU side by side beat the two-kernel route (B = 8192: 1.26 against
// 1.42 ms per iteration).
template <class M, int kProd = kProducers, int RING_KB = ILQR_RING_KB, class MFD = M>
__global__ __launch_bounds__(64 * (1 + kProd)) void k_sweep_backward(BatchViewT<typename M::real> v, M model, MFD fdm, SolverParams sp, int mode, int force,
                                                        const int* __restrict__ commit_idx) {
  using real = typename M::real;
  __shared__ SweepShared<real, M::NX, M::NU, kProd, RING_KB> sh;
  load_step_table(sh.steps);  // (barrier)
  if (blockIdx.x == 0 && threadIdx.x == 64) *v.n_running = 0;  // k_accept of this iteration recounts
  sweep_backward_tile<M, kProd, RING_KB, MFD>(v, model, fdm, sp, mode, force, commit_idx, (int)blockIdx.x, sh);
}

// Whole iterations of the outer loop (ilqr_core.cpp:103-288) for ONE tile, start to finish, in one launch: the
// block alternates between the fused sweep + backward pass (STEP 1 + 2) and the 11-alpha rollouts with the
// accept logic (STEP 3 + 4), n_iters times or until all of its 16 trajectories have left their loops.  Tiles
// never wait for each other: launched per stage, every iteration lasted as long as its SLOWEST tile, twice --
// and late in a solve one tile in 256 is always repeating a backward pass at a raised lambda (ilqr_core.cpp:
// 136-150) or sitting in the slow paths of a box-QP, a different one every time; per tile those passes add up
// to little.  Everything a tile's phases hand each other (gains, status, lambda, candidates, commit indices)
// goes through global memory written and read by wavefronts of the same block, ordered by the block barrier.
//   grid = ntiles, block = 256; one block per CU (about 290 registers x 4 wavefronts, 150 KB of LDS)
// Between the phases of a persistent tile: what one wavefront of the block stored to global memory is read by
// another a moment later, at addresses this CU has read before (gains, the previous iteration's candidates, the
// nominal trajectory).  The LLVM memory model orders that at workgroup scope without waiting for the stores or
// touching the L1 (the wavefronts of a block share a CU); the phases hand over megabytes twice per iteration,
// so this barrier does not lean on it: stores are waited for (vmcnt(0)) and the CU's vector L1 is invalidated.
__device__ __forceinline__ void phase_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// Resources of one persistent tile, by how many tiles share a CU (OCC):
//   1: three producers, a 150 KB ring, rollout inputs prefetched 8 steps ahead, two register sets of records in the
//      backward wavefront: ~400 registers per wavefront, one block per CU -- the shortest iteration for a tile that has a
//      CU to itself (B <= 16 x #CU)
//   2: <= 256 registers and <= 78 KB of LDS, two blocks per CU: two producers (one for m = 2, whose slots are larger) on a
//      72 KB ring, prefetch depth 4, one register set.  Every tile is slower by itself, two side by side are faster:
//      the chip's issue slots, not a tile's latency, are what a big batch is short of.
template <class M, int OCC>
struct SolveCfg {
  using real = typename M::real;
  static constexpr int kRingKb = (OCC == 1) ? ILQR_RING_KB : 72;
  static constexpr int kSlotsAvail = RingSlot<M::NX, M::NU, real, kRingKb>::SLOTS;
  static constexpr int kProd = (OCC == 1) ? kProducers : (kSlotsAvail >= 12 ? 2 : 1);
  static constexpr int kPrefetch = (OCC == 1) ? kDeepPrefetch<M> : 4;
  static constexpr bool kOneSet = (OCC != 1);
};

// Which SIMD of its CU a wavefront runs on, and where its workgroup's LDS allocation starts (in the allocation granule):
// the second of two co-resident workgroups starts above zero.
__device__ __forceinline__ int hw_simd_id() { return (int)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)); }       // HW_ID[5:4]
__device__ __forceinline__ int hw_lds_base() { return (int)__builtin_amdgcn_s_getreg(6 | (0 << 6) | (11 << 11)); }     // LDS_ALLOC[11:0]

template <class M, class MFD, int OCC = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_solve_tile(BatchViewT<typename M::real> v, M model, MFD fdm, AlphaSet alphas, SolverParams sp, int n_iters,
                                                    int force, int* __restrict__ commit_idx, int commit_pending, long long* __restrict__ phase_ticks) {
  using real = typename M::real;
  using Cfg = SolveCfg<M, OCC>;
  __shared__ SweepShared<real, M::NX, M::NU, Cfg::kProd, Cfg::kRingKb> sh;
  __shared__ double lds_cost[NALPHA * TW];
  __shared__ int tile_running;
  __shared__ int simd_mask, chain_wave;
  if (threadIdx.x == 0) {
    simd_mask = 0;
    chain_wave = -1;
  }
  load_step_table(sh.steps);  // (barrier)
  // Roles.  One tile per CU: wavefront 0 runs the backward pass, 1..3 produce; 0..2 roll out.  Two tiles per CU: the
  // dispatcher puts the four wavefronts of a workgroup on the four SIMDs (measured: always, in varying order --
  // scripts/ubench/placement.hip), and a backward chain issues at 0.8 of what a SIMD can issue at all: two chains on one
  // SIMD would halve each other.  So roles go by SIMD: the workgroup whose LDS starts at 0 runs its chain on SIMD 0, the
  // other one on SIMD 2; SIMDs 1 and 3 host both tiles' producers and two of each tile's three rollout wavefronts, the
  // third one (alphas 8..10) runs where the tile's own chain -- idle in that phase -- sits.  A chain never shares its SIMD
  // with a wavefront that is busy at the same time, and the four SIMDs carry about the same number of instructions.
  int role = (int)(threadIdx.x >> 6), rwave = role;
  if constexpr (OCC != 1) {
    const int simd = hw_simd_id();
    if ((threadIdx.x & 63) == 0) atomicOr(&simd_mask, 1 << simd);
    const int chain_simd = (hw_lds_base() != 0) ? 2 : 0;
    if ((threadIdx.x & 63) == 0 && simd == chain_simd) chain_wave = (int)(threadIdx.x >> 6);
    __syncthreads();
    if (simd_mask == 0xF && chain_wave >= 0) {
      const int rel = (simd - chain_simd) & 3;  // 0: chain; 1, 3: the helper SIMDs; 2: the other tile's chain SIMD (this wavefront stays idle)
      role = (rel == 0) ? 0 : (rel == 1) ? 1 : (rel == 3) ? 2 : 3;
      rwave = (rel == 0) ? 2 : (rel == 1) ? 0 : (rel == 3) ? 1 : 3;
    }  // (else: not one wavefront per SIMD -- roles by wavefront index, as with one tile per CU)
  }
  const int tile = blockIdx.x;
  long long t_sweep = 0, t_roll = 0, t0 = 0;
  const bool timing = (phase_ticks != nullptr) & (threadIdx.x == 0);
  const long long c_begin = timing ? clock64() : 0, w_begin = timing ? wall_clock64() : 0;  // shader cycles (s_memtime) / constant-rate ticks
  int it = 0;
  for (; it < n_iters; it++) {
    if (timing) t0 = wall_clock64();
    sweep_backward_tile<M, Cfg::kProd, Cfg::kRingKb, MFD, decltype(sh), Cfg::kOneSet>(v, model, fdm, sp, 1, force, (it > 0 || commit_pending) ? commit_idx : nullptr, tile, sh, role);
    phase_barrier();  // the tile's gains, lambda, status are in memory for its rollout wavefronts
    if (timing) {
      const long long t1 = wall_clock64();
      t_sweep += t1 - t0;
      t0 = t1;
    }
    static_assert(4 * (4 * ((2 * M::NU + M::NU * M::NX + M::NX + 3) / 4) * TW) * sizeof(real) <= sizeof(sh.ring), "the ring holds the four wavefronts' rollout rows");
    // (the ring is idle in this phase: each wavefront's corner of it passes the nominal rows between its alpha groups --
    //  rollout.hpp, SHARE: phase 2 0.205 -> 0.191 ms with one tile per CU, 0.319 -> 0.289 with two)
    rollout_tile<M, true, true, Cfg::kPrefetch, true, true>(v, model, alphas, NALPHA, v.cost_c, 1, sp, commit_idx, tile, lds_cost, /*count_running=*/it == n_iters - 1, rwave,
                                                            sh.ring + (threadIdx.x >> 6) * (4 * ((2 * M::NU + M::NU * M::NX + M::NX + 3) / 4) * TW));
    if (threadIdx.x == 0) tile_running = 0;
    phase_barrier();  // candidates, costs, status, commit indices are in memory for the next sweep
    if (timing) t_roll += wall_clock64() - t0;
    if (!sp.fixed_work) {  // has every trajectory of the tile left its loop?
      const int b = tile * TW + (int)threadIdx.x;
      if (threadIdx.x < TW && b < v.B && v.status[b] == 0) tile_running = 1;
      __syncthreads();
      if (!tile_running) {
        it++;
        break;
      }
    }
  }
  if (timing) {
    phase_ticks[5 * tile + 0] += t_sweep;
    phase_ticks[5 * tile + 1] += t_roll;
    phase_ticks[5 * tile + 2] += it;
    phase_ticks[5 * tile + 3] += clock64() - c_begin;       // shader cycles over the tile's whole run ...
    phase_ticks[5 * tile + 4] += wall_clock64() - w_begin;  // ... and the wall ticks they took: the clock the CU ran at
  }
}

// ------------------------------------------------------------------------------------------
// commit of an accepted candidate
// ------------------------------------------------------------------------------------------
// copy candidate commit_idx[b] into the nominal trajectory.  block 256 = 16 traj x 16 steps.
template <class M>
__global__ __launch_bounds__(256) void k_commit(BatchViewT<typename M::real> v, M model, const int* __restrict__ commit_idx) {
  using real = typename M::real;
  constexpr int NX = M::NX, NU = M::NU;
  const int l = threadIdx.x & (TW - 1);
  const int t = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int tile = blockIdx.y;
  const int b = tile * TW + l;
  const int T = v.T;
  if (t > T || b >= v.B) return;
  const int a = commit_idx[b];
  if (a < 0) return;
  real x[NX], u[NU];
  candidate_knot(v, model, a, tile, t, l, x, u);
#pragma unroll
  for (int i = 0; i < NX; i++) v.xs[tidx(tile, t, i, l, T + 1, NX)] = x[i];
  if (t < T) {
#pragma unroll
    for (int j = 0; j < NU; j++) v.us[tidx(tile, t, j, l, T, NU)] = u[j];
  }
}

}  // namespace ilqr
