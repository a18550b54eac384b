// kernels.hpp -- HIP kernels of the batched iLQR hot path (gfx950 / MI355X).
//
//   k_rollout      forward_pass (src/ilqr_core.cpp:305-337), all line-search alphas concurrently
//   k_derivatives  finite-difference sweep (src/derivatives.cpp + include/finite_diff.h)
//   k_backward_t   backward_pass + box-QP + lambda retry (ilqr_core.cpp:350-401, 136-159),
//                  one THREAD per trajectory (everything in registers)
//   k_accept       first-accept selection, lambda schedule, termination (ilqr_core.cpp:185-282)
//   k_commit       rebuilds the accepted candidate from its checkpoints into the nominal trajectory
//   k_pack/unpack  canonical [B][S][E] <-> tiled [tile][S][E][16]
//
// Lane mapping everywhere: consecutive lanes = consecutive trajectories of a tile, so each
// vector load/store touches whole 128-byte lines of the tiled layout (common.hpp).
#pragma once
#include <type_traits>

#include "boxqp.hpp"
#include "common.hpp"
#include "models.hpp"

namespace ilqr {

// ------------------------------------------------------------------------------------------
// layout conversion
// ------------------------------------------------------------------------------------------
// canonical src[b][s][e]  ->  tiled dst[tile][s][e][l]      (one thread per tiled element)
template <class real>
__global__ void k_pack(const double* __restrict__ src, real* __restrict__ dst, int B, int ntiles, int S, int E) {
  const size_t n = (size_t)ntiles * S * E * TW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i % TW);
    size_t r = i / TW;
    const int e = (int)(r % E);
    r /= E;
    const int s = (int)(r % S);
    const int tile = (int)(r / S);
    const int b = tile * TW + l;
    dst[i] = (b < B) ? (real)src[((size_t)b * S + s) * E + e] : real(0);
  }
}
// tiled src -> canonical dst   (one thread per canonical element; reads are line-strided but
// this path only serves getters)
template <class real>
__global__ void k_unpack(const real* __restrict__ src, double* __restrict__ dst, int B, int S, int E) {
  const size_t n = (size_t)B * S * E;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E);
    size_t r = i / E;
    const int s = (int)(r % S);
    const int b = (int)(r / S);
    dst[i] = (double)src[tidx(b / TW, s, e, b % TW, S, E)];
  }
}
// tiled record sub-range [off, off+E) of a record of size REC  <->  canonical [B][S][E]
template <class real>
__global__ void k_pack_rec(const double* __restrict__ src, real* __restrict__ dst, int B, int ntiles, int S,
                           int REC, int off, int E) {
  const size_t n = (size_t)ntiles * S * E * TW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i % TW);
    size_t r = i / TW;
    const int e = (int)(r % E);
    r /= E;
    const int s = (int)(r % S);
    const int tile = (int)(r / S);
    const int b = tile * TW + l;
    dst[didx(tile, s, off + e, l, S, REC)] = (b < B) ? (real)src[((size_t)b * S + s) * E + e] : real(0);
  }
}
template <class real>
__global__ void k_unpack_rec(const real* __restrict__ src, double* __restrict__ dst, int B, int S, int REC,
                             int off, int E) {
  const size_t n = (size_t)B * S * E;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E);
    size_t r = i / E;
    const int s = (int)(r % S);
    const int b = (int)(r / S);
    dst[i] = (double)src[didx(b / TW, s, off + e, b % TW, S, REC)];
  }
}

// dst slot j <- src slot perm[j], for every slot of the padded batch (compaction of running trajectories between chunks
// of a full solve, capi.hip): tiled arrays [tile][S][E][16] and per-trajectory scalars
template <class real>
__global__ void k_permute_tiled(const real* __restrict__ src, real* __restrict__ dst, const int* __restrict__ perm, int ntiles, int S, int E) {
  const size_t n = (size_t)ntiles * S * E * TW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i % TW);
    size_t r = i / TW;
    const int e = (int)(r % E);
    r /= E;
    const int s = (int)(r % S);
    const int tile = (int)(r / S);
    const int p = perm[tile * TW + l];
    dst[i] = src[tidx(p / TW, s, e, p % TW, S, E)];
  }
}
template <class T>
__global__ void k_permute_scalar(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ perm, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) dst[j] = src[perm[j]];
}

// ------------------------------------------------------------------------------------------
// per-trajectory state reset (init_traj, ilqr_core.cpp:11-56; statics of ilqr.h:17-18)
// ------------------------------------------------------------------------------------------
template <class real>
__global__ void k_reset_state(BatchViewT<real> v, double lambda0, double dlambda0) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= v.Bp) return;
  v.lambda[b] = lambda0;
  v.dlambda[b] = dlambda0;
  v.dV[b] = 0;
  v.dV[v.Bp + b] = 0;
  v.gnorm[b] = 0;
  v.status[b] = (b < v.B) ? 0 : 4;  // padding lanes never run
  v.iters[b] = 0;
  v.flg_change[b] = 1;
  v.alpha_idx[b] = -1;
  v.diverge[b] = 0;
  v.backpass_done[b] = 0;
}

// ------------------------------------------------------------------------------------------
// forward rollout
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// line-search selection + lambda schedule + termination (ilqr_core.cpp:185-282)
// ------------------------------------------------------------------------------------------
// STEP 3/4 for trajectory b; cost_of(a) = cost of its candidate a
template <class View, class CostOf>
__device__ __forceinline__ void accept_one(const View& v, const SolverParams& sp, int b, CostOf cost_of,
                                           int* __restrict__ commit_idx, bool count_running = true) {
  if (b >= v.Bp) return;
  int commit = -1;
  if (b < v.B && v.status[b] == 0) {
    double lambda = v.lambda[b], dlambda = v.dlambda[b];
    const double cost_s = v.cost[b];
    bool fwd = false;
    double new_cost = 0, dcost = 0;
    int acc = -1;
    if (v.backpass_done[b]) {  // :184
      const double dV0 = v.dV[b], dV1 = v.dV[v.Bp + b];
      for (int a = 0; a < NALPHA; a++) {  // the serial order of :185-220, first z > zMin wins
        const double alpha = kAlpha[a];
        new_cost = cost_of(a);
        dcost = cost_s - new_cost;                          // :199
        const double expected = -alpha * (dV0 + alpha * dV1);  // :200
        double z;
        if (expected > 0)
          z = dcost / expected;
        else
          z = (double)((0.0 < dcost) - (dcost < 0.0));  // sgn, common.h:52
        if (z > sp.z_min) {
          fwd = true;
          acc = a;
          break;
        }
      }
    }
    int status = 0;
    if (fwd) {  // :242-263
      dlambda = fmin(dlambda / sp.lambda_factor, 1 / sp.lambda_factor);
      lambda = lambda * dlambda * (lambda > sp.lambda_min ? 1.0 : 0.0);
      v.cost[b] = new_cost;
      v.flg_change[b] = 1;
      commit = acc;
      if (!sp.fixed_work && dcost < sp.tol_fun) status = 2;
    } else {  // :264-282
      dlambda = fmax(dlambda * sp.lambda_factor, sp.lambda_factor);
      lambda = fmax(lambda * dlambda, sp.lambda_min);
      v.flg_change[b] = 0;
      if (!sp.fixed_work && lambda > sp.lambda_max) status = 3;
    }
    v.lambda[b] = lambda;
    v.dlambda[b] = dlambda;
    v.alpha_idx[b] = acc;
    const int it = v.iters[b] + 1;
    v.iters[b] = it;
    // :103.  Not in bench mode: ILQR_FLAG_FIXED_WORK promises that B*T*iters is exactly the work done,
    // whatever max_iter says.
    if (status == 0 && !sp.fixed_work && it >= sp.max_iter) status = 4;
    v.status[b] = status;
    if (status == 0 && count_running) atomicAdd(v.n_running, 1);
  }
  commit_idx[b] = commit;
}
template <class real>
__global__ void k_accept(BatchViewT<real> v, SolverParams sp, int* __restrict__ commit_idx) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  accept_one(v, sp, b, [&](int a) { return v.cost_c[(size_t)a * v.Bp + b]; }, commit_idx);
}

struct AlphaSet {
  double a[NALPHA];  // include/ilqr.h:24 as written; a kernel rounds it to its arithmetic once
};

// One thread per (trajectory, alpha).  A wavefront = one tile of 16 trajectories x 4 alphas
// (lane = 16*alpha_sub + l): the nominal controls, gains and states of the tile are fetched once
// per wavefront and shared by its four alphas (one 128-byte line per load instruction).  AW
// wavefronts of the same tile (alphas 4w..4w+3) form one block, i.e. sit on one CU and share its
// L1.  grid = ntiles, block = 64*AW.
//   GAINS=false : u_t = us[t]                                   (init_traj: K empty, :316)
//   GAINS=true  : u_t = us[t] + alpha k[t] + K[t] (x_t - xs[t]) (:188-190, :315-316)
//   CAND=false  : knots (x_t, u_t) go straight into the nominal tiled xs/us (init_traj)
//   CAND=true   : candidate `a` keeps every u_t and the state at every CT-th knot (common.hpp)
// The cost goes to cost_out[a][b].  mode: 0 = all trajectories, 1 = only running ones whose
// backward pass succeeded.
// ACCEPT: the block also performs STEP 3/4 for its 16 trajectories once its three wavefronts have
// their costs (k_accept's work without a launch of its own; sp, commit_idx are only used then).
// Prefetch depth of the rollout when a tile has a CU to itself: 8 steps for the acrobot (10 doubles per step: 160
// registers of ring), 4 for the double integrator (16 per step: at depth 8 the ring alone is 256 registers, the
// kernel spills -- and inside k_solve_tile the spilled build produced wrong rollouts from knot 59 on).
template <class M>
constexpr int kDeepPrefetch = (M::NU * M::NX + M::NX + 2 * M::NU <= 10) ? 8 : 4;

// (the body of k_rollout for one tile: the persistent kernel k_solve_tile runs it too, with a fourth, idle wavefront)
template <class M, bool GAINS, bool CAND, int PD, bool ACCEPT>
__device__ __forceinline__ void rollout_tile(const BatchViewT<typename M::real>& v, const M& model, const AlphaSet& alphas, int n_alpha,
                                             double* __restrict__ cost_out, int mode, const SolverParams& sp,
                                             int* __restrict__ commit_idx, int tile, double* lds_cost, bool count_running = true, int rwave = -1) {
  using real = typename M::real;
  constexpr int NX = M::NX, NU = M::NU;
  const int wave = (rwave >= 0) ? rwave : (int)(threadIdx.x >> 6);  // which four alphas this wavefront rolls out (>= 3: none)
  const int lane = threadIdx.x & 63;
  const int l = lane & (TW - 1);
  const int a_sub = lane >> 4;
  const int a = wave * 4 + a_sub;
  const int b = tile * TW + l;
  bool active = (b < v.B) && (a < n_alpha);
  if (active && mode == 1) active = (v.status[b] == 0 && v.backpass_done[b]);
  if (!ACCEPT && !active) return;
  if (active) {
  const int T = v.T;
  const real alpha = (real)alphas.a[a < NALPHA ? a : NALPHA - 1];
  const real dt = (real)v.dt;

  real x[NX];
#pragma unroll
  for (int i = 0; i < NX; i++) x[i] = v.x0[tidx(tile, 0, i, l, 1, NX)];
  double total = 0;  // (the sum over the horizon is a per-trajectory accumulator: double in both modes, common.hpp)

  // The nominal controls / gains / states of step t do not depend on the rollout's own state,
  // and one step of arithmetic (~600 cycles) is far shorter than an HBM round trip under load
  // (~2000+ cycles), so they are prefetched PD steps ahead into a ring of register sets; the
  // main loop is unrolled by PD so that every set is statically indexed.
  // Measured on the bench workload (10 loads per step, one block per CU): PD 2 -> 0.298 ms,
  // 4 -> 0.259, 8 -> 0.237, 12 -> 0.236 (more than vmcnt's 63 outstanding), 16 -> 0.76 (register
  // spills).  Depth 8 costs 266 registers, i.e. one block per CU; when the tiles outnumber the CUs
  // the launcher picks depth 4 (136 registers, several blocks per CU hide the latency instead).
  struct StepIn {
    real u[NU], k[GAINS ? NU : 1], K[GAINS ? NU * NX : 1], xnom[GAINS ? NX : 1];
  };
  auto load_step = [&](int t, StepIn& d) __attribute__((always_inline)) {
    t = (t < T) ? t : T - 1;  // tail: harmless re-load instead of a branch
#pragma unroll
    for (int j = 0; j < NU; j++) d.u[j] = v.us[tidx(tile, t, j, l, T, NU)];
    if (GAINS) {
#pragma unroll
      for (int j = 0; j < NU; j++) d.k[j] = v.kff[tidx(tile, t, j, l, T, NU)];
#pragma unroll
      for (int e = 0; e < NU * NX; e++) d.K[e] = v.Kfb[tidx(tile, t, e, l, T, NU * NX)];
#pragma unroll
      for (int i = 0; i < NX; i++) d.xnom[i] = v.xs[tidx(tile, t, i, l, T + 1, NX)];
    }
  };
  auto emit_knot = [&](int t, const real* xx, const real* uu) __attribute__((always_inline)) {  // knot t = (x_t, u_t)
    if (CAND) {
      const int ta = a * v.ntiles + tile;
      if (t < T) {
#pragma unroll
        for (int q = 0; q < NU; q++) v.cand_u[tidx(ta, t, q, l, T, NU)] = uu[q];
      }
      if ((t & (CT - 1)) == 0) {
#pragma unroll
        for (int i = 0; i < NX; i++) v.cand_x[tidx(ta, t / CT, i, l, v.nch, NX)] = xx[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NX; i++) v.xs[tidx(tile, t, i, l, T + 1, NX)] = xx[i];
      if (t < T) {
#pragma unroll
        for (int q = 0; q < NU; q++) v.us[tidx(tile, t, q, l, T, NU)] = uu[q];  // :323 (no clamping)
      }
    }
  };
  auto do_step = [&](int t, const StepIn& d) __attribute__((always_inline)) {
    real u[NU];
#pragma unroll
    for (int j = 0; j < NU; j++) u[j] = d.u[j];
    if (GAINS) {
#pragma unroll
      for (int j = 0; j < NU; j++) {
        u[j] += d.k[j] * alpha;  // :190
        real acc = 0;
#pragma unroll
        for (int i = 0; i < NX; i++) acc += d.K[j + NU * i] * (x[i] - d.xnom[i]);
        u[j] += acc;  // :316
      }
    }
    if (sp.fixes & 1) {  // opt-in fix: "the right way" of ilqr_core.cpp:327-329 -- the clamped control is stored and integrated
#pragma unroll
      for (int j = 0; j < NU; j++) u[j] = min_of(max_of(u[j], model.u_min[j]), model.u_max[j]);
    }
    emit_knot(t, x, u);
    total += (double)model.cost(x, u);  // :324
    real x1[NX];
    integrate_dynamics(model, x, u, dt, x1);  // :325
#pragma unroll
    for (int i = 0; i < NX; i++) x[i] = x1[i];
  };
  StepIn ring[PD];
#pragma unroll
  for (int d = 0; d < PD; d++) load_step(d, ring[d]);
  int t = 0;
  for (; t + PD <= T; t += PD) {
#pragma unroll
    for (int d = 0; d < PD; d++) {
      do_step(t + d, ring[d]);          // (the set is consumed in place and refilled right after: copying it out first so
      load_step(t + d + PD, ring[d]);   //  that the refill could be issued a step earlier cost ten register moves per step)
    }
  }
  for (; t < T; t++) {  // remainder (< PD steps)
    StepIn cur;
    load_step(t, cur);
    do_step(t, cur);
  }
  {  // knot T: the final state (no control)
    real uz[NU];
#pragma unroll
    for (int q = 0; q < NU; q++) uz[q] = 0;
    emit_knot(T, x, uz);
  }
  total += (double)model.final_cost(x);  // :335
  cost_out[(size_t)a * v.Bp + b] = total;
  if (ACCEPT) lds_cost[a * TW + l] = total;
  }  // if (active)
  if constexpr (ACCEPT) {
    __syncthreads();
    if (threadIdx.x < TW)
      accept_one(v, sp, tile * TW + (int)threadIdx.x, [&](int aa) { return lds_cost[aa * TW + threadIdx.x]; }, commit_idx, count_running);
  }
}

template <class M, bool GAINS, bool CAND, int PD = 4, bool ACCEPT = false>
__global__ __launch_bounds__(192) void k_rollout(BatchViewT<typename M::real> v, M model, AlphaSet alphas, int n_alpha,
                                                 double* __restrict__ cost_out, int mode, SolverParams sp,
                                                 int* __restrict__ commit_idx) {
  __shared__ double lds_cost[ACCEPT ? NALPHA * TW : 1];
  rollout_tile<M, GAINS, CAND, PD, ACCEPT>(v, model, alphas, n_alpha, cost_out, mode, sp, commit_idx, (int)blockIdx.x, lds_cost);
}

// Knot t of candidate `a` of trajectory (tile, l): the control as stored, the state re-integrated
// from the checkpoint at knot (t/CT)*CT with the rollout's own step (include/model.h:12-15).  The
// CT-1 controls of the chunk are fetched up front (one memory round trip), the steps run predicated.
template <class M>
__device__ __forceinline__ void candidate_knot(const BatchViewT<typename M::real>& v, const M& model, int a, int tile, int t, int l,
                                               typename M::real* x, typename M::real* u) {
  using real = typename M::real;
  constexpr int NX = M::NX, NU = M::NU;
  const int T = v.T, ta = a * v.ntiles + tile, c = t / CT, off = t - c * CT;
#pragma unroll
  for (int i = 0; i < NX; i++) x[i] = v.cand_x[tidx(ta, c, i, l, v.nch, NX)];
  real uq[CT][NU];
#pragma unroll
  for (int q = 0; q < CT; q++) {
    const int tq = (c * CT + q < T) ? c * CT + q : T - 1;
#pragma unroll
    for (int j = 0; j < NU; j++) uq[q][j] = v.cand_u[tidx(ta, tq, j, l, T, NU)];
  }
#pragma unroll
  for (int j = 0; j < NU; j++) u[j] = 0.0;  // knot T has no control
#pragma unroll
  for (int q = 0; q < CT; q++) {
    if (q < off) {
      real x1[NX];
      integrate_dynamics(model, x, uq[q], (real)v.dt, x1);
#pragma unroll
      for (int i = 0; i < NX; i++) x[i] = x1[i];
    }
    if (q == off && t < T) {
#pragma unroll
      for (int j = 0; j < NU; j++) u[j] = uq[q][j];
    }
  }
}

// candidate `a` -> canonical xs [B][T+1][nx], us [B][T][nu]   (getter only)
template <class M>
__global__ void k_unpack_cand(BatchViewT<typename M::real> v, M model, int a, double* __restrict__ xs, double* __restrict__ us) {
  using real = typename M::real;
  constexpr int NX = M::NX, NU = M::NU;
  const int T = v.T;
  const size_t n = (size_t)v.B * (T + 1);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % (T + 1));
    const int b = (int)(i / (T + 1));
    real x[NX], u[NU];
    candidate_knot(v, model, a, b / TW, t, b % TW, x, u);
    if (xs)
      for (int e = 0; e < NX; e++) xs[((size_t)b * (T + 1) + t) * NX + e] = (double)x[e];
    if (us && t < T)
      for (int e = 0; e < NU; e++) us[((size_t)b * T + t) * NU + e] = (double)u[e];
  }
}

// ------------------------------------------------------------------------------------------
// finite-difference derivatives
// ------------------------------------------------------------------------------------------
// include/finite_diff.h:67-86 applied to a scalar functor of an N-vector.
template <int N, class real, class F>
__device__ __forceinline__ void fd_hessian(const real* x, F f, real* out /* N x N col-major */) {
#pragma unroll
  for (int i = 0; i < N; i++)
#pragma unroll
    for (int j = i; j < N; j++) {
      real pp[N], pm[N], mp[N], mm[N];
#pragma unroll
      for (int l = 0; l < N; l++) pp[l] = pm[l] = mp[l] = mm[l] = x[l];
      pp[i] += real(kEps);
      pp[j] += real(kEps);
      pm[i] += real(kEps);
      pm[j] -= real(kEps);
      mp[i] -= real(kEps);
      mp[j] += real(kEps);
      mm[i] -= real(kEps);
      mm[j] -= real(kEps);
      const real v = (f(pp) - f(mp) - f(pm) + f(mm)) / real(4 * kEps * kEps);
      out[i + N * j] = v;
      out[j + N * i] = v;
    }
}
// include/finite_diff.h:22-33
template <int N, class real, class F>
__device__ __forceinline__ void fd_gradient(const real* x, F f, real* out) {
#pragma unroll
  for (int i = 0; i < N; i++) {
    real p[N], m[N];
#pragma unroll
    for (int l = 0; l < N; l++) p[l] = m[l] = x[l];
    p[i] += real(kEps);
    m[i] -= real(kEps);
    out[i] = (f(p) - f(m)) / real(2 * kEps);
  }
}

// One thread per knot point (b, t), t = 0..T.  block = 256 = 16 trajectories x 16 time steps,
// grid = (ceil((T+1)/16), ntiles).  force != 0: every trajectory (stage call / bench mode),
// otherwise only running trajectories whose flgChange is set (ilqr_core.cpp:115).
// commit_idx (may be null): a line search accepted candidate commit_idx[b] for trajectory b and
// its copy into the nominal trajectory is still pending -- this kernel reads the knot from the
// candidate and performs the copy on the way (the separate k_commit pass is only used to flush).
// RING (k_sweep_backward): the record and the knot's nominal control are ALSO written to the LDS
// slot `rs` (this lane's pair column of the slot: element e at rs[(e>>1)*2*TW + (e&1)], the
// control behind the record), where the backward wavefront of the same block reads them.
#ifndef ILQR_RING_KB
#define ILQR_RING_KB 150  // one block per CU; the two-blocks-per-CU variant of k_sweep_backward uses 60
#endif
// PAD: extra `real`s per pair row.  0 = the HBM tile layout (row = 16 trajectories x 2 elements = a whole number of LDS
// bank cycles: rows of the same trajectory share their banks, which is what the 4-lane backward wavefront wants -- its
// lanes read one row for 16 trajectories).  The 16-lane backward wavefronts (backward_hex.hpp) read up to 8 ROWS for
// one trajectory with one instruction: PAD = 2 (16 bytes) spreads the rows over the banks.
template <int NX, int NU, class real, int RING_KB = ILQR_RING_KB, int PAD = 0>
struct RingSlot {
  static constexpr int US = Rec<NX, NU>::SIZE;               // controls follow the record
  static constexpr int PAIRS = (Rec<NX, NU>::SIZE + NU + 1) / 2;
  static constexpr int ROW = 2 * TW + PAD;                   // `real`s from one element pair to the next
  static constexpr int ELEMS = PAIRS * ROW;                  // per slot, in units of `real`
  static constexpr int SLOTS = ((RING_KB * 1024 / (int)sizeof(real)) / ELEMS) / 4 * 4;  // ring size: what fits in RING_KB of LDS
};

// MFD / fdm: the model in the arithmetic the finite differences are TAKEN in.  For an fp64 handle that is
// the model itself.  For an fp32 handle it is its double-precision twin: eps = 1e-3 second differences of
// a cost of O(1e3) are rounding noise in float (1e3 x 6e-8 / 4e-6 = 15 against Hessian entries of 800), so
// the knot (float) is widened, the sweep below runs in double exactly as for an fp64 handle, and the record
// is rounded to float when stored.  Rollouts, the commit and the analytic route stay in the handle's own
// arithmetic (`model`).
template <class M, bool RING = false, class MFD = M, int RING_PAD = 0>
__device__ __forceinline__ void derivatives_of_knot(const BatchViewT<typename M::real>& v, const M& model, const MFD& fdm, int force,
                                                    const int* __restrict__ commit_idx, int tile, int t, int l,
                                                    typename M::real* rs = nullptr, bool records = true) {
  using real = typename M::real;
  using fdr = typename MFD::real;
  using RSl = RingSlot<M::NX, M::NU, real>;
  constexpr int NX = M::NX, NU = M::NU;
  using R = Rec<NX, NU>;
  typedef real real2_t __attribute__((ext_vector_type(2)));
  const int b = tile * TW + l;
  const int T = v.T;
  if (t > T || b >= v.B) return;
  const int ci = commit_idx ? commit_idx[b] : -1;
  // Stand-alone sweep (k_derivatives): the records live in HBM, and a trajectory whose last line search failed
  // keeps them (flgChange = 0, ilqr_core.cpp:115).  Fused sweep (RING): the records exist ONLY in the LDS ring,
  // for as long as the backward wavefront needs them -- every running trajectory's are recomputed each time
  // (they are a function of the nominal trajectory: same values), nothing is written to HBM but the commit.
  // (records == false: the caller only wants the pending commit performed -- a fused sweep whose backward pass has moved on)
  const bool want = records && (force || (v.status[b] == 0 && (RING || v.flg_change[b])));
  if (ci < 0 && !want) return;
  const real dt = (real)v.dt;

  real xk[NX], uk[NU];  // the knot as stored
  {
    if (ci >= 0) {  // knot t of the accepted candidate
      candidate_knot(v, model, ci, tile, t, l, xk, uk);
    } else {
#pragma unroll
      for (int i = 0; i < NX; i++) xk[i] = v.xs[tidx(tile, t, i, l, T + 1, NX)];
#pragma unroll
      for (int j = 0; j < NU; j++) uk[j] = (t < T) ? v.us[tidx(tile, t, j, l, T, NU)] : real(0);  // derivatives.cpp:35-38
    }
    if (ci >= 0) {  // the pending commit of ilqr_core.cpp:210-213 ("accept": xs, us keep the new rollout)
#pragma unroll
      for (int i = 0; i < NX; i++) v.xs[tidx(tile, t, i, l, T + 1, NX)] = xk[i];
      if (t < T) {
#pragma unroll
        for (int j = 0; j < NU; j++) v.us[tidx(tile, t, j, l, T, NU)] = uk[j];
      }
    }
  }
  if (!want) return;  // (finished trajectory whose last candidate was committed above)

  real* D = RING ? nullptr : v.D + didx(tile, t, 0, l, T + 1, R::SIZE);
  auto put = [&](int e, fdr val_) {
    const real val = (real)val_;
    if (RING)
      rs[(e >> 1) * (2 * TW + RING_PAD) + (e & 1)] = val;
    else
      D[(size_t)(e >> 1) * (2 * TW) + (e & 1)] = val;
  };
  auto put2 = [&](int e, fdr v0, fdr v1) {  // e even: one store of a pair
    real2_t w;
    w.x = (real)v0;
    w.y = (real)v1;
    if (RING)
      *reinterpret_cast<real2_t*>(rs + (e >> 1) * (2 * TW + RING_PAD)) = w;
    else
      *reinterpret_cast<real2_t*>(D + (size_t)(e >> 1) * (2 * TW)) = w;
  };
  if (RING) {
#pragma unroll
    for (int j = 0; j < NU; j++) rs[((RSl::US + j) >> 1) * (2 * TW + RING_PAD) + ((RSl::US + j) & 1)] = uk[j];
    // m = 1: the slot's last pair has a free half next to u -- the weight 1 / (|u| + 1) of this knot's gradient-norm
    // term (:405-412) goes there, computed here (the same recip() of the same value) instead of on the backward
    // wavefront's chain
    if constexpr (NU == 1) rs[((RSl::US + 1) >> 1) * (2 * TW + RING_PAD) + ((RSl::US + 1) & 1)] = recip(abs_of(uk[0]) + real(1));
  }

  if constexpr (has_analytic_record<M>::value) {
    if (v.analytic) {  // opt-in: the model's exact derivatives (wave-uniform branch)
      real rec[R::SIZE];
      model.analytic_record(xk, uk, dt, t == T, rec);
#pragma unroll
      for (int e = 0; e < R::SIZE; e += 2) put2(e, (fdr)rec[e], (fdr)rec[e + 1]);
      return;
    }
  }
  // the knot in the finite differences' arithmetic (a no-op unless the handle is fp32)
  fdr x[NX], u[NU];
#pragma unroll
  for (int i = 0; i < NX; i++) x[i] = (fdr)xk[i];
#pragma unroll
  for (int j = 0; j < NU; j++) u[j] = (fdr)uk[j];
  const fdr dtf = (fdr)dt;
  if (t < T) {
    // fx, fu: central differences of the Euler map (derivatives.cpp:19-25, finite_diff.h:35-47)
#pragma unroll
    for (int i = 0; i < NX; i++) {
      fdr p[NX], m[NX], fp[NX], fm[NX];
#pragma unroll
      for (int q = 0; q < NX; q++) p[q] = m[q] = x[q];
      p[i] += fdr(kEps);
      m[i] -= fdr(kEps);
      integrate_dynamics(fdm, p, u, dtf, fp);
      integrate_dynamics(fdm, m, u, dtf, fm);
#pragma unroll
      for (int r = 0; r < NX; r += 2)
        put2(R::FX + r + NX * i, (fp[r] - fm[r]) / fdr(2 * kEps), (fp[r + 1] - fm[r + 1]) / fdr(2 * kEps));
    }
#pragma unroll
    for (int i = 0; i < NU; i++) {
      fdr p[NU], m[NU], fp[NX], fm[NX];
#pragma unroll
      for (int q = 0; q < NU; q++) p[q] = m[q] = u[q];
      p[i] += fdr(kEps);
      m[i] -= fdr(kEps);
      integrate_dynamics(fdm, x, p, dtf, fp);
      integrate_dynamics(fdm, x, m, dtf, fm);
#pragma unroll
      for (int r = 0; r < NX; r += 2)
        put2(R::FU + r + NX * i, (fp[r] - fm[r]) / fdr(2 * kEps), (fp[r + 1] - fm[r + 1]) / fdr(2 * kEps));
    }
    // cx, cu (derivatives.cpp:44-47)
    fdr g[NX > NU ? NX : NU];
    fd_gradient<NX>(x, [&](const fdr* xx) { return fdm.cost(xx, u); }, g);
#pragma unroll
    for (int i = 0; i < NX; i += 2) put2(R::CX + i, g[i], g[i + 1]);
    fd_gradient<NU>(u, [&](const fdr* uu) { return fdm.cost(x, uu); }, g);
#pragma unroll
    for (int i = 0; i < NU; i++) put(R::CU + i, g[i]);
    // cxx (derivatives.cpp:76-96)
    fdr H[NX * NX];
    fd_hessian<NX>(x, [&](const fdr* xx) { return fdm.cost(xx, u); }, H);
#pragma unroll
    for (int e = 0; e < NX * NX; e += 2) put2(R::CXX + e, H[e], H[e + 1]);
  } else {
#pragma unroll
    for (int e = 0; e < NX * NX + NX * NU; e += 2) put2(R::FX + e, fdr(0), fdr(0));  // fx[T], fu[T] stay zero
    fdr g[NX];
    fd_gradient<NX>(x, [&](const fdr* xx) { return fdm.final_cost(xx); }, g);  // :49
#pragma unroll
    for (int i = 0; i < NX; i += 2) put2(R::CX + i, g[i], g[i + 1]);
#pragma unroll
    for (int i = 0; i < NU; i++) put(R::CU + i, fdr(0));  // :50-51
    fdr H[NX * NX];
    fd_hessian<NX>(x, [&](const fdr* xx) { return fdm.final_cost(xx); }, H);  // :92
#pragma unroll
    for (int e = 0; e < NX * NX; e += 2) put2(R::CXX + e, H[e], H[e + 1]);
  }
  // cuu at every t, with u = 0 at t = T (derivatives.cpp:98-112)
  {
    fdr H[NU * NU];
    fd_hessian<NU>(u, [&](const fdr* uu) { return fdm.cost(x, uu); }, H);
#pragma unroll
    for (int e = 0; e < NU * NU; e++) put(R::CUU + e, H[e]);
  }
  // cxu (derivatives.cpp:114-144)
#pragma unroll
  for (int i = 0; i < NX; i++)
#pragma unroll
    for (int j = 0; j < NU; j++) {
      fdr px[NX], mx[NX], pu[NU], mu[NU];
#pragma unroll
      for (int q = 0; q < NX; q++) px[q] = mx[q] = x[q];
#pragma unroll
      for (int q = 0; q < NU; q++) pu[q] = mu[q] = u[q];
      px[i] += fdr(kEps);
      mx[i] -= fdr(kEps);
      pu[j] += fdr(kEps);
      mu[j] -= fdr(kEps);
      fdr val;
      if (t < T)
        val = (fdm.cost(px, pu) - fdm.cost(mx, pu) - fdm.cost(px, mu) + fdm.cost(mx, mu)) / fdr(4 * (kEps * kEps));
      else  // :140 (the reference's own "TODO this is wrong"; value is never consumed)
        val = (fdm.final_cost(px) - fdm.final_cost(mx) - fdm.final_cost(px) + fdm.final_cost(mx)) /
              (4 * (fdr(kEps) * fdr(kEps)));
      put(R::CXU + i + NX * j, val);
    }
}

// grid = (ceil((T+1)/16), ntiles), block = 256 = 16 time steps x 16 trajectories
template <class M, class MFD = M>
__global__ __launch_bounds__(256) void k_derivatives(BatchViewT<typename M::real> v, M model, MFD fdm, int force, const int* __restrict__ commit_idx) {
  const int l = threadIdx.x & (TW - 1);
  const int t = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *v.n_running = 0;  // k_accept of this iteration recounts
  derivatives_of_knot<M, false, MFD>(v, model, fdm, force, commit_idx, (int)blockIdx.y, t, l);
}

// ------------------------------------------------------------------------------------------
// backward pass, one thread per trajectory
// ------------------------------------------------------------------------------------------
// mode 0: exactly one backward_pass() at the current lambda for every trajectory (stage call)
// mode 1: STEP 2 of the outer loop for running trajectories: retry with increased lambda while
//         the pass diverges (ilqr_core.cpp:136-150), then the gradient-norm test (:153-159).
template <class M>
__global__ __launch_bounds__(64) void k_backward_t(BatchViewT<typename M::real> v, M model, SolverParams sp, int mode) {
  using real = typename M::real;
  constexpr int NX = M::NX, NU = M::NU;
  using R = Rec<NX, NU>;
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= v.B) return;
  if (mode == 1 && v.status[b] != 0) return;
  const int tile = b / TW, l = b % TW;
  const int T = v.T;
  double lambda = v.lambda[b], dlambda = v.dlambda[b];
  const real* Dt = v.D + didx(tile, 0, 0, l, T + 1, R::SIZE);
  auto rec = [&](int t, int e) { return Dt[((size_t)t * (R::SIZE / 2) + (e >> 1)) * (2 * TW) + (e & 1)]; };

  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0;  // (per-trajectory accumulators: double in both modes)
  while (true) {
    real Vx[NX], Vxx[NX * NX], kprev[NU];
#pragma unroll
    for (int i = 0; i < NX; i++) Vx[i] = rec(T, R::CX + i);  // :353
#pragma unroll
    for (int e = 0; e < NX * NX; e++) Vxx[e] = rec(T, R::CXX + e);  // :354
#pragma unroll
    for (int j = 0; j < NU; j++) kprev[j] = v.kff[tidx(tile, T - 1, j, l, T, NU)];  // k[min(i+1,T-1)] at i=T-1
    dV0 = dV1 = 0;  // :356
    diverge = 0;

    for (int i = T - 1; i >= 0; i--) {
      real fx[NX * NX], fu[NX * NU], cx[NX], cu[NU], cxx[NX * NX], cxu[NX * NU], cuu[NU * NU], us[NU];
#pragma unroll
      for (int e = 0; e < NX * NX; e++) fx[e] = rec(i, R::FX + e);
#pragma unroll
      for (int e = 0; e < NX * NU; e++) fu[e] = rec(i, R::FU + e);
#pragma unroll
      for (int e = 0; e < NX; e++) cx[e] = rec(i, R::CX + e);
#pragma unroll
      for (int e = 0; e < NU; e++) cu[e] = rec(i, R::CU + e);
#pragma unroll
      for (int e = 0; e < NX * NX; e++) cxx[e] = rec(i, R::CXX + e);
#pragma unroll
      for (int e = 0; e < NX * NU; e++) cxu[e] = rec(i, R::CXU + e);
#pragma unroll
      for (int e = 0; e < NU * NU; e++) cuu[e] = rec(i, R::CUU + e);
#pragma unroll
      for (int j = 0; j < NU; j++) us[j] = v.us[tidx(tile, i, j, l, T, NU)];

      real Qx[NX], Qu[NU], Qxx[NX * NX], Qux[NU * NX], Quu[NU * NU], QuuF[NU * NU];
      real A1[NX * NX], A2[NU * NX];
      // :359-360
#pragma unroll
      for (int a = 0; a < NX; a++) {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < NX; q++) acc += fx[q + NX * a] * Vx[q];
        Qx[a] = cx[a] + acc;
      }
#pragma unroll
      for (int a = 0; a < NU; a++) {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < NX; q++) acc += fu[q + NX * a] * Vx[q];
        Qu[a] = cu[a] + acc;
      }
      // :361  Qxx = cxx + (fx'Vxx) fx
#pragma unroll
      for (int a = 0; a < NX; a++)
#pragma unroll
        for (int c = 0; c < NX; c++) {
          real acc = 0;
#pragma unroll
          for (int q = 0; q < NX; q++) acc += fx[q + NX * a] * Vxx[q + NX * c];
          A1[a + NX * c] = acc;
        }
#pragma unroll
      for (int a = 0; a < NX; a++)
#pragma unroll
        for (int c = 0; c < NX; c++) {
          real acc = 0;
#pragma unroll
          for (int q = 0; q < NX; q++) acc += A1[a + NX * q] * fx[q + NX * c];
          Qxx[a + NX * c] = cxx[a + NX * c] + acc;
        }
      // :362/:366  Qux = cxu' + (fu'Vxx) fx
#pragma unroll
      for (int a = 0; a < NU; a++)
#pragma unroll
        for (int c = 0; c < NX; c++) {
          real acc = 0;
#pragma unroll
          for (int q = 0; q < NX; q++) acc += fu[q + NX * a] * Vxx[q + NX * c];
          A2[a + NU * c] = acc;
        }
#pragma unroll
      for (int a = 0; a < NU; a++)
#pragma unroll
        for (int c = 0; c < NX; c++) {
          real acc = 0;
#pragma unroll
          for (int q = 0; q < NX; q++) acc += A2[a + NU * q] * fx[q + NX * c];
          Qux[a + NU * c] = cxu[c + NX * a] + acc;
        }
      // :363/:367  Quu = cuu + (fu'Vxx) fu ; QuuF = cuu + lambda I + (fu'Vxx) fu
#pragma unroll
      for (int a = 0; a < NU; a++)
#pragma unroll
        for (int c = 0; c < NU; c++) {
          real acc = 0;
#pragma unroll
          for (int q = 0; q < NX; q++) acc += A2[a + NU * q] * fu[q + NX * c];
          Quu[a + NU * c] = cuu[a + NU * c] + acc;
          QuuF[a + NU * c] = (cuu[a + NU * c] + ((a == c) ? (real)lambda : real(0))) + acc;
        }
      // opt-in (sp.fixes & 4): lambda regularises Vxx' ([Tassa 2012] eq. 10) instead of Quu:
      // Quu_reg = Quu + lambda fu'fu, Qux_reg = Qux + lambda fu'fx; the value update keeps Quu, Qux
      real Quxr[NU * NX];
#pragma unroll
      for (int e = 0; e < NU * NX; e++) Quxr[e] = Qux[e];
      if (sp.fixes & 4) {
        const real lam = (real)lambda;
#pragma unroll
        for (int a = 0; a < NU; a++) {
#pragma unroll
          for (int c = 0; c < NU; c++) {
            real acc = 0;
#pragma unroll
            for (int q = 0; q < NX; q++) acc += fu[q + NX * a] * fu[q + NX * c];
            QuuF[a + NU * c] = Quu[a + NU * c] + lam * acc;
          }
#pragma unroll
          for (int c = 0; c < NX; c++) {
            real acc = 0;
#pragma unroll
            for (int q = 0; q < NX; q++) acc += fu[q + NX * a] * fx[q + NX * c];
            Quxr[a + NU * c] = Qux[a + NU * c] + lam * acc;
          }
        }
      }

      // :369
      real lo[NU], hi[NU];
#pragma unroll
      for (int j = 0; j < NU; j++) {
        lo[j] = model.u_min[j] - us[j];
        hi[j] = model.u_max[j] - us[j];
      }
      BoxQPResult<NU, real> qp;
      box_qp<NU>(QuuF, Qu, kprev, lo, hi, qp, (sp.fixes & 2) != 0);
      if (qp.result < 1) {  // :371
        diverge = i;
        break;
      }

      // :373-385
      real K[NU * NX];
#pragma unroll
      for (int e = 0; e < NU * NX; e++) K[e] = 0;
      {
        int rank[NU], nf = 0;
#pragma unroll
        for (int j = 0; j < NU; j++) {
          rank[j] = nf;
          nf += qp.v_free[j] ? 1 : 0;
        }
        if (nf > 0) {
          real Minv[NU * NU];
          rinv_rinvT<NU>(qp.nfR, qp.R, Minv);
          const int nuse = (nf < qp.nfR) ? nf : qp.nfR;
#pragma unroll
          for (int c = 0; c < NX; c++) {
            real qf[NU];  // rows_w_ind(Qux_reg, v_free)(:, c)
#pragma unroll
            for (int a = 0; a < NU; a++) {
              real val = 0;
#pragma unroll
              for (int j = 0; j < NU; j++)
                if (qp.v_free[j] && rank[j] == a) val = Quxr[j + NU * c];
              qf[a] = val;
            }
#pragma unroll
            for (int j = 0; j < NU; j++) {
              if (qp.v_free[j] && rank[j] < nuse) {
                real acc = 0;
#pragma unroll
                for (int a = 0; a < NU; a++)
                  if (a < nuse) {
                    real mrow = 0;  // Minv[rank[j]][a]
#pragma unroll
                    for (int r = 0; r < NU; r++)
                      if (r == rank[j]) mrow = Minv[r + NU * a];
                    acc += -mrow * qf[a];
                  }
                K[j + NU * c] = acc;
              }
            }
          }
        }
      }

      // :388-389
      {
        real d0 = 0;
#pragma unroll
        for (int j = 0; j < NU; j++) d0 += qp.x[j] * Qu[j];
        dV0 += (double)d0;
        real d1 = 0;
#pragma unroll
        for (int c = 0; c < NU; c++) {
          real r = 0;
#pragma unroll
          for (int a = 0; a < NU; a++) r += (real(0.5) * qp.x[a]) * Quu[a + NU * c];
          d1 += r * qp.x[c];
        }
        dV1 += (double)d1;
      }
      // :391-393
      {
        real T1[NX * NU];  // K' Quu  (NX x NU)
#pragma unroll
        for (int a = 0; a < NX; a++)
#pragma unroll
          for (int c = 0; c < NU; c++) {
            real acc = 0;
#pragma unroll
            for (int q = 0; q < NU; q++) acc += K[q + NU * a] * Quu[q + NU * c];
            T1[a + NX * c] = acc;
          }
        real Vxn[NX], Vn[NX * NX];
#pragma unroll
        for (int a = 0; a < NX; a++) {
          real t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
          for (int c = 0; c < NU; c++) {
            t1 += T1[a + NX * c] * qp.x[c];
            t2 += K[c + NU * a] * Qu[c];
            t3 += Qux[c + NU * a] * qp.x[c];
          }
          Vxn[a] = ((Qx[a] + t1) + t2) + t3;
        }
#pragma unroll
        for (int a = 0; a < NX; a++)
#pragma unroll
          for (int c = 0; c < NX; c++) {
            real t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
            for (int q = 0; q < NU; q++) {
              t1 += T1[a + NX * q] * K[q + NU * c];
              t2 += K[q + NU * a] * Qux[q + NU * c];
              t3 += Qux[q + NU * a] * K[q + NU * c];
            }
            Vn[a + NX * c] = ((Qxx[a + NX * c] + t1) + t2) + t3;
          }
#pragma unroll
        for (int a = 0; a < NX; a++) {
          Vx[a] = Vxn[a];
#pragma unroll
          for (int c = 0; c < NX; c++) Vxx[a + NX * c] = real(0.5) * (Vn[a + NX * c] + Vn[c + NX * a]);
        }
      }
      // :396-397
#pragma unroll
      for (int j = 0; j < NU; j++) {
        v.kff[tidx(tile, i, j, l, T, NU)] = qp.x[j];
        kprev[j] = qp.x[j];
      }
#pragma unroll
      for (int e = 0; e < NU * NX; e++) v.Kfb[tidx(tile, i, e, l, T, NU * NX)] = K[e];
    }  // for i

    if (mode == 0) {
      done = (diverge == 0);
      break;
    }
    if (diverge != 0) {  // :142-148
      dlambda = fmax(dlambda * sp.lambda_factor, sp.lambda_factor);
      lambda = fmax(lambda * dlambda, sp.lambda_min);
      if (lambda > sp.lambda_max) break;
      continue;
    }
    done = true;
    break;
  }

  v.dV[b] = dV0;
  v.dV[v.Bp + b] = dV1;
  v.diverge[b] = diverge;
  v.backpass_done[b] = done ? 1 : 0;
  if (mode == 1) {
    v.lambda[b] = lambda;
    v.dlambda[b] = dlambda;
  }
  // :153 / :405-412  gnorm = mean_t max_j |k_j| / (|u_j| + 1), ascending t like std::accumulate
  double acc = 0;
  for (int t = 0; t < T; t++) {
    real mx = 0;
#pragma unroll
    for (int j = 0; j < NU; j++) {
      const real val = abs_of(v.kff[tidx(tile, t, j, l, T, NU)]) / (abs_of(v.us[tidx(tile, t, j, l, T, NU)]) + 1);
      mx = (j == 0 || val > mx) ? val : mx;
    }
    acc += (double)mx;
  }
  const double gnorm = acc / T;
  v.gnorm[b] = gnorm;
  if (mode == 1 && !sp.fixed_work && gnorm < sp.tol_grad && lambda < 1e-5) {  // :154-159
    v.status[b] = 1;
    v.iters[b] += 1;  // this iteration was started
  }
}

// ------------------------------------------------------------------------------------------
// backward pass, one QUAD of lanes per trajectory (NX == 4): one wavefront = one tile of 16
// trajectories.  Lane (l, s) = 4 l + s owns column s of every nx-by-nx quantity of trajectory l;
// the small dense products are split by column, the box-QP (m x m, scalar-sized) is evaluated
// redundantly by the four lanes, and columns are exchanged with DPP quad_perm broadcasts
// (v_mov_b32 dpp, no LDS).  The derivative records of step i-1 are prefetched into a second
// register set while step i computes: the recursion never waits on HBM.
// ------------------------------------------------------------------------------------------
// value of the neighbouring lane l ^ 1 (quad_perm:[1,0,3,2])
__device__ __forceinline__ double dpp_swap1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float dpp_swap1(float x) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xf, 0xf, true));
}
template <int SRC>
__device__ __forceinline__ double quad_bcast(double x) {
  constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);  // quad_perm:[SRC,SRC,SRC,SRC]
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int SRC>
__device__ __forceinline__ float quad_bcast(float x) {  // fp32: one v_mov_b32 dpp per broadcast instead of two
  constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), ctrl, 0xf, 0xf, true));
}
template <class real>
__device__ __forceinline__ void quad_gather(real x, real out[4]) {
  out[0] = quad_bcast<0>(x);
  out[1] = quad_bcast<1>(x);
  out[2] = quad_bcast<2>(x);
  out[3] = quad_bcast<3>(x);
}

// backtracking step sizes as the reference's loop produces them, in each arithmetic (boxqp.hpp)
__device__ __constant__ const StepTableT<double> kStepTable{};
__device__ __constant__ const StepTableT<float> kStepTableF{};
__device__ __forceinline__ const double* step_table(double) { return kStepTable.s; }
__device__ __forceinline__ const float* step_table(float) { return kStepTableF.s; }

// Quad-parallel Armijo line search for the scalar QP (all four lanes of a quad hold the same
// QP1State).  The reference's loop (boxqp.cpp:156-173) tries step_k = 0.6^k for k = 0, 1, 2, ...
// until the Armijo test passes; a Newton step that a bound truncates to a tiny fraction needs
// 10+ trips, and a wavefront pays for its slowest quad.  The set of passing k is upward
// closed (while the trial point sits on the bound the value is constant and the threshold shrinks
// with the step; once it is inside the bound a Newton step always passes).  So the four lanes
// evaluate four candidates in ONE instruction stream -- lane 0 the unit step, lanes 1..3 a window
// k1, k1+1, k1+2 around an fp32 estimate of the answer -- with the exact test and the exact step
// table, and the first passing candidate whose predecessor is known to fail is taken.  Anything
// else (estimate off, Q <= 0, k near the minStep cut-off) returns false and the caller runs the
// sequential loop: the result is the reference's either way.
template <class real>
__device__ __forceinline__ bool qp1_search_quad(QP1StateT<real>& q, int s, int lane, const real* __restrict__ lds_steps) {
  const real bound = (q.search > 0) ? q.hi : q.lo;
  const real v_b = qp1_value(q, bound);
  // fp32 estimates: f = fraction of the step inside the box, r = Armijo threshold on the bound
  const float f = (float)(bound - q.x) * __builtin_amdgcn_rcpf((float)q.search);  // 1-ulp v_rcp_f32: only an estimate
  const float r = (float)(v_b - q.old_v) * __builtin_amdgcn_rcpf((float)(real(kArmijo) * q.slope));
  const float thr = fmaxf(f, r);
  int kg = (int)ceilf(__log2f(thr) * -1.35691545f);  // log(thr)/log(0.6)
  // (Q < 0 too -- Eigen's unchecked factor makes that a legal QP, and in float Quu = cuu + fu'Vxx fu cancels to <= 0
  //  for a few trajectories late in a solve: along a descent direction the value change of a trial on the bound is a
  //  negative constant N, the test passes iff step <= N / (0.1 slope), and an interior trial has ratio
  //  1 + Q step search^2 / (2 slope) > 1: the passing set is upward closed for either sign of Q)
  const bool sane = (q.Q != real(0)) & (thr > 0.f) & (thr < 1.f) & (kg >= 1) & (kg <= 96);
  const int k1 = (sane & (kg > 2)) ? kg - 1 : 1;
  const int my_k = (s == 0) ? 0 : k1 + s - 1;
  const real my_step = lds_steps[my_k];
  const real my_x1 = qp1_trial(q, my_step);
  const real my_v1 = qp1_value(q, my_x1);
  const bool my_pass = !qp1_armijo_fails(q, my_v1, my_step);
  const unsigned long long bal = __ballot(my_pass);
  const unsigned int m4 = (unsigned int)(bal >> (lane & ~3)) & 0xFu;
  // which candidate wins: lane 0 if the unit step passes, else the first passing window lane,
  // provided its predecessor failed (in the window, or k = 0 when the window starts at k = 1)
  const bool unit = (m4 & 1u) != 0u;
  const unsigned int w = m4 >> 1;
  const int wwin = __ffs(w);  // 1..3, 0 if none
  const bool wok = (w != 0u) & ((wwin > 1) | (k1 == 1)) & (sane | (k1 == 1));
  const bool ok = unit | wok;
  const int win = unit ? 0 : wwin;
  const int src = (lane & ~3) + (ok ? win : 0);
  q.x1 = __shfl(my_x1, src, 64);
  q.v1 = __shfl(my_v1, src, 64);
  q.step = __shfl(my_step, src, 64);
  // A unit-step trial that lands on x itself (x sits on the bound the search points across, or the
  // step is below half an ulp of x) stays there for every shorter step: value == old value, the
  // Armijo ratio is 0 at every k, and the reference's loop runs its ~100 trips down to minStep and
  // reports failure (boxqp.cpp:167-171).  Same outcome, without the trips -- late in a solve this
  // was a quarter of the steps of the slowest tiles.
  const bool stuck = (qp1_trial(q, real(1)) == q.x) & !q.early;
  // The same once the search direction is rounding noise (late in a solve Quu reaches 1e12+ and x
  // sits on the optimum to an ulp: search ~ 1e-19): steps 1 and 0.6 still move x by an ulp, from
  // 0.36 on the trial IS x.  With the window at k = 1, 2, 3 every k <= 3 has been tested exactly;
  // if none passes and the k = 3 trial equals x, no later k can pass either.
  const unsigned int s4 = (unsigned int)(__ballot(my_x1 == q.x) >> (lane & ~3)) & 0xFu;
  const bool dead = (k1 == 1) & (m4 == 0u) & ((s4 & 8u) != 0u) & !q.early;
  q.ls_failed = q.ls_failed | stuck | dead;
  return ok | q.early | stuck | dead;
}

template <int NU, class real>
struct QuadStep {  // what lane (l, s) needs of one derivative record, AS LOADED: element pairs stay pairs until the
  // step that consumes them unpacks them.  (Unpacked into scalars at load time, the two halves of one 8-byte
  // load flowed into separate loop-carried registers; for float hipcc then put a copy -- and the s_waitcnt vmcnt
  // it needs -- right behind the freshly issued prefetch: one exposed HBM round trip per step.)
  typedef real pair_t __attribute__((ext_vector_type(2)));
  pair_t fx[8];                        // full fx (replicated over s)
  pair_t fxc[2];                       // fx[:, s] again, loaded by address so no register array is indexed by s
  pair_t fu[2 * NU];                   // full fu
  pair_t tail[(NU + NU * NU) / 2];     // cu, cuu
  pair_t cxx[2];                       // cxx[:, s]
  real us[NU];
  real usw;                            // m = 1, from the ring: 1 / (|us| + 1), written there by the producers
  real cx;                             // cx[s]
  real cxu[NU];                        // cxu[s, :]
};

// The body of the quad backward pass for one tile, run by ONE wavefront (lane = 4*l + s).
// gate.wait(t) returns once the derivative record and the nominal control of knot t may be read: a
// no-op when the records were written by an earlier kernel (k_backward_q), a wait on the
// co-resident producer wavefronts in k_sweep_backward.
// ring != nullptr: the FIRST pass reads each knot from LDS slot (T - t) % SLOTS, where the producers
// put it; lambda-retry passes (and ring == nullptr) read the records from HBM.
// A Gate says where the records come from.  NoGate: they are in HBM (k_backward_q, written by k_derivatives).
// RingGate (k_sweep_backward): the producer wavefronts of the block compute them into the LDS ring, pass after
// pass -- a lambda-retry pass is a second sweep, nothing is ever read back from HBM.
struct NoGate {
  static constexpr bool kRing = false;
  __device__ __forceinline__ void begin_pass() {}
  __device__ __forceinline__ void wait(int) {}
  __device__ __forceinline__ int slot(int) const { return 0; }
  __device__ __forceinline__ void finish() {}
};

// FIXES: the opt-in deviations (sp.fixes) are compiled in; callers branch ONCE on sp.fixes != 0 and
// run the copy without them otherwise -- inside the step their tests were 10 instructions of the common path.
// ONESET: one register set for the records instead of two (ring only): the load of a step is issued at the top of that step
// and waited for -- ~150 exposed cycles per step, 72 registers less: what lets two tiles share a CU (k_solve_tile<.., 2>),
// where the other tile's wavefronts fill the gap.
template <class M, class Gate, int RING_KB = ILQR_RING_KB, bool FIXES = true, bool ONESET = false>
__device__ __forceinline__ void backward_quad(const BatchViewT<typename M::real>& v, const M& model, const SolverParams& sp, int mode,
                                              int tile, int lane, const typename M::real* __restrict__ lds_steps, Gate& gate,
                                              const typename M::real* ring = nullptr) {
  using real = typename M::real;
  static_assert(M::NX == 4, "quad kernel: one lane per state dimension");
  constexpr int NX = 4, NU = M::NU;
  using R = Rec<NX, NU>;
  const int l = lane >> 2, s = lane & 3;
  const int b = tile * TW + l;
  if (b >= v.B) return;                        // quad-uniform
  if (mode == 1 && v.status[b] != 0) return;   // quad-uniform
  const int T = v.T;
  double lambda = v.lambda[b], dlambda = v.dlambda[b];
  typedef real real2_t __attribute__((ext_vector_type(2)));
  const real* __restrict__ Dt = Gate::kRing ? nullptr : v.D + didx(tile, 0, 0, l, T + 1, R::SIZE);
  const real* __restrict__ ust = v.us + tidx(tile, 0, 0, l, T, NU);
  real* __restrict__ kt = v.kff + tidx(tile, 0, 0, l, T, NU);
  real* __restrict__ Kt = v.Kfb + tidx(tile, 0, 0, l, T, NU * NX);

  using RS = RingSlot<NX, NU, real, RING_KB>;
  constexpr bool RP = Gate::kRing;  // records (and the knot's control) come from the LDS ring
  // what lane (l, s) needs of knot t, given accessors for element pairs (e even) / single elements
  auto fill = [&](auto pair, auto one, QuadStep<NU, real>& d) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 16; e += 2) d.fx[e >> 1] = pair(R::FX + e);
#pragma unroll
    for (int q = 0; q < 4; q += 2) d.fxc[q >> 1] = pair(R::FX + q + 4 * s);
#pragma unroll
    for (int e = 0; e < 4 * NU; e += 2) d.fu[e >> 1] = pair(R::FU + e);
#pragma unroll
    for (int e = 0; e < NU + NU * NU; e += 2) d.tail[e >> 1] = pair(R::CU + e);  // cu and cuu are adjacent: nu(nu+1) elements, an even count at an even offset
    d.cx = one(R::CX + s);
#pragma unroll
    for (int i = 0; i < 4; i += 2) d.cxx[i >> 1] = pair(R::CXX + i + 4 * s);
#pragma unroll
    for (int a = 0; a < NU; a++) d.cxu[a] = one(R::CXU + s + 4 * a);
  };
  // (explicit address spaces: with generic pointers hipcc merges the two sources' loads into flat
  // instructions, which wait on vmcnt and lgkmcnt alike)
  typedef const __attribute__((address_space(3))) real lds_cd;
  typedef const __attribute__((address_space(3))) real2_t lds_cd2;
  auto load = [&](int t, QuadStep<NU, real>& d) __attribute__((always_inline)) {
    gate.wait(t);
    if constexpr (RP) {  // ds_read_b128 / b64 from the producers' slot
      lds_cd* r = (lds_cd*)(ring + gate.slot(t) * RS::ELEMS + l * 2);
      auto pair = [&](int e) { return *(lds_cd2*)(r + (e >> 1) * (2 * TW)); };
      auto one = [&](int e) { return r[(e >> 1) * (2 * TW) + (e & 1)]; };
      fill(pair, one, d);
#pragma unroll
      for (int a = 0; a < NU; a++) d.us[a] = one(RS::US + a);
      if constexpr (NU == 1) d.usw = one(RS::US + 1);
    } else {  // 16-byte / 8-byte global loads of the record in HBM
      const real* r = Dt + (unsigned)(t * ((R::SIZE / 2) * 2 * TW));  // in-tile offsets fit 32 bits
      auto pair = [&](int e) { return *reinterpret_cast<const real2_t*>(r + (unsigned)((e >> 1) * (2 * TW))); };
      auto one = [&](int e) { return r[(unsigned)((e >> 1) * (2 * TW) + (e & 1))]; };
      fill(pair, one, d);
#pragma unroll
      for (int a = 0; a < NU; a++) d.us[a] = ust[(unsigned)((t * NU + a) * TW)];
    }
  };

  constexpr int kWaitAll = (7 << 4) | (15 << 8);  // s_waitcnt vmcnt(0) only (expcnt/lgkmcnt untouched)
  constexpr int kWaitLds = 0xC07F;                // s_waitcnt lgkmcnt(0) only (vmcnt = 63, expcnt = 7)
  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0, gacc = 0;  // per-trajectory accumulators: double in both modes
  // one backward_pass() at the current lambda
  auto one_pass = [&]() __attribute__((always_inline)) {
    gate.begin_pass();
    // carried state: full Vxx / Vx in every lane
    real Vx[4], Vxx[16], kprev[NU];
    const real lam_r = (real)lambda;  // the regularisation of this pass in the handle's arithmetic (:367)
    {
      gate.wait(T);
      if constexpr (RP) {
        lds_cd* r = (lds_cd*)(ring + gate.slot(T) * RS::ELEMS + l * 2);
#pragma unroll
        for (int i = 0; i < 4; i++) Vx[i] = r[((R::CX + i) >> 1) * (2 * TW) + ((R::CX + i) & 1)];  // :353
#pragma unroll
        for (int e = 0; e < 16; e++) Vxx[e] = r[((R::CXX + e) >> 1) * (2 * TW) + ((R::CXX + e) & 1)];  // :354
      } else {
        const real* r = Dt + (size_t)T * (R::SIZE / 2) * (2 * TW);
#pragma unroll
        for (int i = 0; i < 4; i++) Vx[i] = r[(size_t)((R::CX + i) >> 1) * (2 * TW) + ((R::CX + i) & 1)];  // :353
#pragma unroll
        for (int e = 0; e < 16; e++) Vxx[e] = r[(size_t)((R::CXX + e) >> 1) * (2 * TW) + ((R::CXX + e) & 1)];  // :354
      }
    }
#pragma unroll
    for (int a = 0; a < NU; a++) kprev[a] = kt[((size_t)(T - 1) * NU + a) * TW];
    dV0 = dV1 = 0;
    diverge = 0;
    gacc = 0;

    // one Riccati step; returns false if the box-QP reports failure (ilqr_core.cpp:371)
    auto step = [&](int i, const QuadStep<NU, real>& raw) -> bool {
      struct {  // the record, unpacked (register renames: the loads have landed, see QuadStep)
        real fx[16], fxc[4], fu[4 * NU], cu[NU], cuu[NU * NU], us[NU], cx, cxx[4], cxu[NU];
      } d;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        d.fx[2 * e] = raw.fx[e].x;
        d.fx[2 * e + 1] = raw.fx[e].y;
      }
#pragma unroll
      for (int e = 0; e < 2; e++) {
        d.fxc[2 * e] = raw.fxc[e].x;
        d.fxc[2 * e + 1] = raw.fxc[e].y;
        d.cxx[2 * e] = raw.cxx[e].x;
        d.cxx[2 * e + 1] = raw.cxx[e].y;
      }
#pragma unroll
      for (int e = 0; e < 2 * NU; e++) {
        d.fu[2 * e] = raw.fu[e].x;
        d.fu[2 * e + 1] = raw.fu[e].y;
      }
      {
        real tail[NU + NU * NU];
#pragma unroll
        for (int e = 0; e < (NU + NU * NU) / 2; e++) {
          tail[2 * e] = raw.tail[e].x;
          tail[2 * e + 1] = raw.tail[e].y;
        }
#pragma unroll
        for (int e = 0; e < NU; e++) d.cu[e] = tail[e];
#pragma unroll
        for (int e = 0; e < NU * NU; e++) d.cuu[e] = tail[NU + e];
      }
      d.cx = raw.cx;
#pragma unroll
      for (int a = 0; a < NU; a++) {
        d.cxu[a] = raw.cxu[a];
        d.us[a] = raw.us[a];
      }
      // W = Vxx' * fx[:, s]   (column s)
      real W[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += Vxx[r + 4 * q] * d.fxc[q];
        W[r] = acc;
      }
      // Qxx[:, s] = cxx[:, s] + fx' W      :361
      real Qxxc[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += d.fx[q + 4 * r] * W[q];
        Qxxc[r] = d.cxx[r] + acc;
      }
      // Qx[s] = cx[s] + fx[:, s]' Vx'      :359
      real Qxs;
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += d.fxc[q] * Vx[q];
        Qxs = d.cx + acc;
      }
      // Qux[:, s] = cxu[s, :]' + fu' W     :362/:366
      real Quxc[NU];
#pragma unroll
      for (int a = 0; a < NU; a++) {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += d.fu[q + 4 * a] * W[q];
        Quxc[a] = d.cxu[a] + acc;
      }
      // replicated: Qu, wv = Vxx' fu, Quu, QuuF     :360, :363, :367
      real Qu[NU], Quu[NU * NU], QuuF[NU * NU];
#pragma unroll
      for (int a = 0; a < NU; a++) {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += d.fu[q + 4 * a] * Vx[q];
        Qu[a] = d.cu[a] + acc;
      }
#pragma unroll
      for (int c = 0; c < NU; c++) {
        real wv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          real acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += Vxx[r + 4 * q] * d.fu[q + 4 * c];
          wv[r] = acc;
        }
#pragma unroll
        for (int a = 0; a < NU; a++) {
          real acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += d.fu[q + 4 * a] * wv[q];
          Quu[a + NU * c] = d.cuu[a + NU * c] + acc;
          QuuF[a + NU * c] = (d.cuu[a + NU * c] + ((a == c) ? lam_r : real(0))) + acc;
        }
      }
      // opt-in (sp.fixes & 4, see k_backward_t): Quu_reg = Quu + lambda fu'fu, Qux_reg[:, s] = Qux[:, s] + lambda fu'fx[:, s]
      real Quxr[NU];
#pragma unroll
      for (int a = 0; a < NU; a++) Quxr[a] = Quxc[a];
      const bool reg_vxx = FIXES && (sp.fixes & 4) != 0;
      if (reg_vxx) {
#pragma unroll
        for (int a = 0; a < NU; a++) {
#pragma unroll
          for (int c = 0; c < NU; c++) {
            real acc = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) acc += d.fu[q + 4 * a] * d.fu[q + 4 * c];
            QuuF[a + NU * c] = Quu[a + NU * c] + lam_r * acc;
          }
          real acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += d.fu[q + 4 * a] * d.fxc[q];
          Quxr[a] = Quxc[a] + lam_r * acc;
        }
      }
      // :369  box-QP (replicated in the quad)
      real lo[NU], hi[NU];
#pragma unroll
      for (int a = 0; a < NU; a++) {
        lo[a] = model.u_min[a] - d.us[a];
        hi[a] = model.u_max[a] - d.us[a];
      }
      // :371  a failed QP ends the pass.  No early return: the rest of the step is computed
      // anyway (its results are discarded) so that the vmcnt wait below sits on every path.
      struct { real x[NU]; } qp;
      real Kc[NU];
      bool ok;
      int k_free = 0;      // (NU == 1: what K is scaled from, see the exchange below)
      real k_minv = 0;
      if constexpr (NU == 1) {
        int free0;
        real minv;
        QP1StateT<real> q1;
        qp1_begin<false>(QuuF[0], Qu[0], kprev[0], lo[0], hi[0], q1, FIXES && (sp.fixes & 2) != 0);
        if (__builtin_expect(!qp1_search_quad(q1, s, lane, lds_steps), 0)) {  // fallback: rare, out of line
          q1.step = 1;
          q1.x1 = qp1_trial(q1, real(1));
          q1.v1 = qp1_value(q1, q1.x1);
          qp1_backtrack_seq(q1);
        }
        int result = qp1_finish(q1, qp.x[0], free0, minv);
        if (result == kQpGoesOn)  // the QP goes on (rare early in a solve, a quarter of the steps of some tiles later)
          result = qp1_continue(
              q1,
              [&](QP1StateT<real>& qs) __attribute__((always_inline)) {
                if (__builtin_expect(!qp1_search_quad(qs, s, lane, lds_steps), 0)) {
                  qp1_line_search_seq(qs);
                }
              },
              qp.x[0], free0);
        ok = result >= 1;
        Kc[0] = free0 ? -minv * Quxr[0] : real(0);  // :373-385
        k_free = free0;
        k_minv = minv;
      } else if constexpr (NU == 2) {
        // m = 2: the scalarised solver (boxqp.hpp: box_qp2); K[:, s] = -(R^-1 R^-T) Qux[free, s] scattered to the free rows (:373-385)
        BoxQP2Result<real> r;
        box_qp2(QuuF, Qu, kprev, lo, hi, r, FIXES && (sp.fixes & 2) != 0);
        ok = r.result >= 1;
        qp.x[0] = r.x[0];
        qp.x[1] = r.x[1];
        const bool both = r.free0 & r.free1;
        const real q0 = r.free0 ? Quxr[0] : Quxr[1];  // rows_w_ind(Qux_reg, v_free)(:, s), by rank
        const real kA = (r.nfR == 2) ? (-r.m00 * q0 + -r.m01 * Quxr[1]) : -r.m00 * q0;  // rank 0 (the second term only if both are free)
        const real kB = -r.m01 * Quxr[0] + -r.m11 * Quxr[1];                              // rank 1
        Kc[0] = r.free0 ? kA : real(0);
        Kc[1] = r.free1 ? (both ? kB : kA) : real(0);
      } else {
        BoxQPResult<NU, real> r;
        box_qp<NU>(QuuF, Qu, kprev, lo, hi, r, FIXES && (sp.fixes & 2) != 0);
        ok = r.result >= 1;
#pragma unroll
        for (int a = 0; a < NU; a++) {
          qp.x[a] = r.x[a];
          Kc[a] = 0;
        }
        // :373-385  K[:, s] = -(R^-1 R^-T) Qux[free, s] scattered to the free rows
        int rank[NU], nf = 0;
#pragma unroll
        for (int a = 0; a < NU; a++) {
          rank[a] = nf;
          nf += r.v_free[a] ? 1 : 0;
        }
        if (nf > 0) {
          real Minv[NU * NU], qf[NU];
          rinv_rinvT<NU>(r.nfR, r.R, Minv);
          const int nuse = (nf < r.nfR) ? nf : r.nfR;
#pragma unroll
          for (int a = 0; a < NU; a++) {
            real val = 0;
#pragma unroll
            for (int j = 0; j < NU; j++)
              if (r.v_free[j] && rank[j] == a) val = Quxr[j];
            qf[a] = val;
          }
#pragma unroll
          for (int j = 0; j < NU; j++)
            if (r.v_free[j] && rank[j] < nuse) {
              real acc = 0;
#pragma unroll
              for (int a = 0; a < NU; a++)
                if (a < nuse) {
                  real mrow = 0;
#pragma unroll
                  for (int rr = 0; rr < NU; rr++)
                    if (rr == rank[j]) mrow = Minv[rr + NU * a];
                  acc += -mrow * qf[a];
                }
              Kc[j] = acc;
            }
        }
      }
      if (!ok) diverge = i;
      // :388-389
      {
        real d0 = 0;
#pragma unroll
        for (int a = 0; a < NU; a++) d0 += qp.x[a] * Qu[a];
        if (ok) dV0 += (double)d0;
        real d1 = 0;
#pragma unroll
        for (int c = 0; c < NU; c++) {
          real r = 0;
#pragma unroll
          for (int a = 0; a < NU; a++) r += (real(0.5) * qp.x[a]) * Quu[a + NU * c];
          d1 += r * qp.x[c];
        }
        if (ok) dV1 += (double)d1;
      }
      // T1s[c] = (K' Quu)[s, c]
      real T1s[NU];
#pragma unroll
      for (int c = 0; c < NU; c++) {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < NU; q++) acc += Kc[q] * Quu[q + NU * c];
        T1s[c] = acc;
      }
      // :391  Vx[s]
      real Vxs;
      {
        real t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
        for (int c = 0; c < NU; c++) {
          t1 += T1s[c] * qp.x[c];
          t2 += Kc[c] * Qu[c];
          t3 += Quxc[c] * qp.x[c];
        }
        Vxs = ((Qxs + t1) + t2) + t3;
      }
      // exchange K, Qux, K'Quu columns inside the quad
      real Kall[NU][4], Qall[NU][4], T1all[NU][4];
#pragma unroll
      for (int a = 0; a < NU; a++) quad_gather(Quxc[a], Qall[a]);  // (does not wait for the box-QP)
      if (NU == 1 && !reg_vxx) {
        // K[0, r] = -minv Qux[0, r] in lane r; the same product of the same operands here: no second exchange
#pragma unroll
        for (int r = 0; r < 4; r++) Kall[0][r] = k_free ? -k_minv * Qall[0][r] : real(0);
      } else {
#pragma unroll
        for (int a = 0; a < NU; a++) quad_gather(Kc[a], Kall[a]);
      }
#pragma unroll
      for (int c = 0; c < NU; c++)  // (K'Quu)[r, c] for every r, from the gathered K (no third exchange)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          real acc = 0;
#pragma unroll
          for (int q = 0; q < NU; q++) acc += Kall[q][r] * Quu[q + NU * c];
          T1all[c][r] = acc;
        }
      // :392  Vn[r, s] = Qxx[r,s] + (K'Quu)[r,:] K[:,s] + K[:,r]' Qux[:,s] + Qux[:,r]' K[:,s]
      real Vn[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        real t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
        for (int q = 0; q < NU; q++) {
          t1 += T1all[q][r] * Kc[q];
          t2 += Kall[q][r] * Quxc[q];
          t3 += Qall[q][r] * Kc[q];
        }
        Vn[r] = ((Qxxc[r] + t1) + t2) + t3;
      }
      // all-gather, then :393 symmetrise (every lane keeps the full matrix)
      real Vf[16];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        real col[4];
        quad_gather(Vn[r], col);  // col[c] = Vn[r, c]
#pragma unroll
        for (int c = 0; c < 4; c++) Vf[r + 4 * c] = col[c];
      }
      // 0.5 (V + V'): the diagonal is 0.5 (a + a) = a exactly, each off-diagonal pair is one sum
      // (fp addition commutes), so 6 add/mul pairs instead of 16
#pragma unroll
      for (int r = 0; r < 4; r++) {
        Vxx[r + 4 * r] = Vf[r + 4 * r];
#pragma unroll
        for (int c = r + 1; c < 4; c++) {
          const real sym = real(0.5) * (Vf[r + 4 * c] + Vf[c + 4 * r]);
          Vxx[r + 4 * c] = sym;
          Vxx[c + 4 * r] = sym;
        }
      }
      quad_gather(Vxs, Vx);
      // :405-412 term of the gradient norm for this step (summed here in descending t)
      {
        real mx = 0;
#pragma unroll
        for (int a = 0; a < NU; a++) {
          const real val = abs_of(qp.x[a]) * ((RP && NU == 1) ? raw.usw : recip(abs_of(d.us[a]) + 1));
          mx = (a == 0 || val > mx) ? val : mx;
        }
        if (ok) gacc += (double)mx;
      }
      // the prefetch issued at the top of this step has had the whole step to land.  From HBM:
      // vmcnt(0) (this also drains the previous step's stores).  From the ring: only the LDS
      // reads are waited for -- they must have landed before the next gate() frees the slot --
      // and the stores are never waited for.
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (RP)
        __builtin_amdgcn_s_waitcnt(kWaitLds);
      else
        __builtin_amdgcn_s_waitcnt(kWaitAll);
      __builtin_amdgcn_sched_barrier(0);
      // :396-397
      if (ok) {
#pragma unroll
        for (int a = 0; a < NU; a++) {
          kprev[a] = qp.x[a];
          Kt[(unsigned)((i * NU * NX + a + NU * s) * TW)] = Kc[a];
        }
        if (s == 0) {
#pragma unroll
          for (int a = 0; a < NU; a++) kt[(unsigned)((i * NU + a) * TW)] = qp.x[a];
        }
      }
      return ok;
    };

    {
      // Two register sets, ping-pong.  Order inside one half-iteration:
      //   issue the loads of the NEXT step into the idle set
      //   -> compute this step from the set that has already landed
      //   -> s_waitcnt vmcnt(0): everything outstanding here was issued a whole step ago (the
      //      prefetch above, the previous step's stores), so this wait is normally free
      //   -> issue this step's stores (never waited for).
      // The explicit wait + sched_barriers keep hipcc from parking its own vmcnt(0) right behind
      // the freshly issued prefetch, which would expose one HBM round trip per step.
      QuadStep<NU, real> A, Bd;
      int i = T - 1;
      if constexpr (RP && (NU > 1 || ONESET)) {
        // m > 1 from the ring: ONE register set, loaded at the top of its own step.  The second set (86 registers for
        // m = 2) pushed the step's live values into AGPR copies; an LDS read is ~150 cycles of a 6000-cycle step.
        while (true) {
          __builtin_amdgcn_sched_barrier(0);
          load(i, A);
          __builtin_amdgcn_s_waitcnt(kWaitLds);
          __builtin_amdgcn_sched_barrier(0);
          if (!step(i, A)) break;
          if (--i < 0) break;
        }
      } else {
      load(i, A);
      __builtin_amdgcn_s_waitcnt(kWaitAll & kWaitLds);
      while (true) {
        __builtin_amdgcn_sched_barrier(0);
        if (i >= 1) load(i - 1, Bd);
        __builtin_amdgcn_sched_barrier(0);
        if (!step(i, A)) break;
        if (--i < 0) break;
        __builtin_amdgcn_sched_barrier(0);
        if (i >= 1) load(i - 1, A);
        __builtin_amdgcn_sched_barrier(0);
        if (!step(i, Bd)) break;
        if (--i < 0) break;
      }
      }
    }

  };

  while (true) {
    one_pass();
    if (mode == 0) {
      done = (diverge == 0);
      break;
    }
    if (diverge != 0) {  // :142-148
      dlambda = fmax(dlambda * sp.lambda_factor, sp.lambda_factor);
      lambda = fmax(lambda * dlambda, sp.lambda_min);
      if (lambda > sp.lambda_max) break;
      continue;  // (a fused sweep produces the records again: Gate::begin_pass)
    }
    done = true;
    break;
  }

  // :153 / :405-412 gradient norm.  A completed pass has summed its terms on the fly (descending
  // t; the reference sums ascending -- same value to rounding).  Only when the pass was abandoned
  // (lambda > lambdaMax) do k[0..T) hold a mix of old and new gains; then re-read them.
  double acc = gacc;
  if (!done) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    acc = 0;
    for (int t0 = 0; t0 < T; t0 += 8) {
      real kv[8][NU], uv[8][NU];
#pragma unroll
      for (int j = 0; j < 8; j++)
#pragma unroll
        for (int a = 0; a < NU; a++) {
          const int t = (t0 + j < T) ? t0 + j : T - 1;
          kv[j][a] = kt[((size_t)t * NU + a) * TW];
          uv[j][a] = ust[((size_t)t * NU + a) * TW];
        }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        real mx = 0;
#pragma unroll
        for (int a = 0; a < NU; a++) {
          const real val = abs_of(kv[j][a]) / (abs_of(uv[j][a]) + 1);
          mx = (a == 0 || val > mx) ? val : mx;
        }
        if (t0 + j < T) acc += (double)mx;
      }
    }
  }
  const double gnorm = acc / T;
  if (s == 0) {
    v.dV[b] = dV0;
    v.dV[v.Bp + b] = dV1;
    v.diverge[b] = diverge;
    v.backpass_done[b] = done ? 1 : 0;
    v.gnorm[b] = gnorm;
    if (mode == 1) {
      v.lambda[b] = lambda;
      v.dlambda[b] = dlambda;
      if (!sp.fixed_work && gnorm < sp.tol_grad && lambda < 1e-5) {  // :154-159
        v.status[b] = 1;
        v.iters[b] += 1;
      }
    }
  }
}

template <class real>
__device__ __forceinline__ void load_step_table(real* lds_steps) {
  const real* tab = step_table(real(0));
  for (int k = threadIdx.x; k < 104; k += blockDim.x) lds_steps[k] = tab[k];
  __syncthreads();
}

// stage call / records already in HBM: grid = ntiles, block = 64
template <class M>
__global__ __launch_bounds__(64) void k_backward_q(BatchViewT<typename M::real> v, M model, SolverParams sp, int mode) {
  using real = typename M::real;
  __shared__ real lds_steps[104];  // backtracking step sizes (per-lane indexed -> LDS, not constant cache)
  load_step_table(lds_steps);
  NoGate gate;
  if (sp.fixes)
    backward_quad<M, NoGate, ILQR_RING_KB, true>(v, model, sp, mode, (int)blockIdx.x, (int)threadIdx.x, lds_steps, gate);
  else
    backward_quad<M, NoGate, ILQR_RING_KB, false>(v, model, sp, mode, (int)blockIdx.x, (int)threadIdx.x, lds_steps, gate);
}

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
  real steps[104];                    // backtracking step sizes (per-lane indexed -> LDS, not constant cache)
  real ring[RS::SLOTS * RS::ELEMS];   // knot with running index G lives in slot G % SLOTS
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
    rollout_tile<M, true, true, Cfg::kPrefetch, true>(v, model, alphas, NALPHA, v.cost_c, 1, sp, commit_idx, tile, lds_cost, /*count_running=*/it == n_iters - 1, rwave);
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
