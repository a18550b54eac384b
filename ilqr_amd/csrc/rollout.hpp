// rollout.hpp -- forward_pass (src/ilqr_core.cpp:305-337) for all line-search alphas concurrently, the accept logic
// (STEP 3/4, :185-282), candidate checkpoints and their re-integration.
#pragma once
#include "layout.hpp"

namespace ilqr {

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
#ifndef ILQR_ROLLOUT_UNCLAMPED_FETCH
#define ILQR_ROLLOUT_UNCLAMPED_FETCH 1
#endif
constexpr int kRolloutFetchSlack = 8;  // >= every prefetch depth PD of the shared-row rollouts
template <class M>
constexpr int kDeepPrefetch = (M::NU * M::NX + M::NX + 2 * M::NU <= 10) ? 8 : 4;

// (the body of k_rollout for one tile: the persistent kernel k_solve_tile runs it too, with a fourth, idle wavefront)
//
// SHARE (with `share` = 4 * ceil(rows / 4) * TW reals of LDS owned by this wavefront): the four alpha groups of a wavefront
// need the SAME nominal rows (u, k, K, xs of 16 trajectories at step t) and each fetched all of them -- ten vector memory
// instructions per step, every one ~10.6 cycles of the CU's address path whatever its lane mask or footprint
// (scripts/ubench/vmem.hip), which with eight wavefronts rolling out is two thirds of a step's issue time: the wide
// kernels' rollouts ran 1.5 x their instruction count.  Shared, group s fetches rows s, s + 4, s + 8 (three instructions
// for ten rows), the wavefront passes them through LDS one step ahead (in-order LDS: no barrier) and every lane reads its
// ten values back with broadcast reads.  Same values: bit-identical.  It pays even with one rollout wavefront per SIMD
// (the 16-trajectory kernel at B <= 16 x #CU: 0.205 -> 0.191 ms per rollout phase); the stage kernel k_rollout, which has
// no LDS to spare a priori, keeps the per-lane loads.
// NOFIX: the caller's route is never taken with the opt-in fixes (sp.fixes == 0: the persistent matrix-core and wide kernels) --
// the clamp of ilqr_core.cpp:327-329's "right way" and its selects leave the step (4 of its ~170 instructions).
// CANDT (k_solve_hex, nu = 1): the candidates in GROUPS of CG consecutive controls (and whole checkpoint states) per trajectory,
//     cand_u [alpha plane][tile][t / CG][TW][CG]        cand_x [alpha plane][tile][chunk][TW][nx]
// instead of one control (one state component) per trajectory.  In the plane layout the commit of the accepted candidates gathers 8
// bytes per trajectory and step from up to eleven planes (the sixteen trajectories of a tile accept different alphas): a whole memory
// sector fetched for every 8 bytes, 437 MB read per iteration at B = 4096 against 204 algorithmic.  Grouped, a task of the commit reads
// its checkpoint as ONE 32-byte piece and its eight controls as two -- whole sectors of what it needs -- and a rollout lane stores four
// controls every fourth step (16 lanes x 32 bytes: whole lines, a quarter of the store instructions).  (A time-innermost row per
// trajectory reads as well but WRITES 16-byte pieces of 64 different lines per instruction: + 90 MB of write traffic per iteration and
// + 1 % time, profiles/r06b_candidate_layouts.txt.)
constexpr int CG = 4;
static_assert(CT % CG == 0, "a chunk of the commit is a whole number of control groups");
__host__ __device__ inline int cand_groups(int T) { return (T + CT + CG - 1) / CG; }   // groups per plane row: whole chunks past T - 1
__host__ __device__ inline size_t cand_g_u(int ta, int grp, int l, int T) { return (((size_t)ta * cand_groups(T) + grp) * TW + l) * CG; }
__host__ __device__ inline size_t cand_g_x(int ta, int c, int l, int nch, int NX) { return (((size_t)ta * nch + c) * TW + l) * NX; }
template <class M, bool GAINS, bool CAND, int PD, bool ACCEPT, bool SHARE = false, bool NOFIX = false, int NG = 1, bool CANDT = false>
__device__ __forceinline__ void rollout_tile(const BatchViewT<typename M::real>& v, const M& model, const AlphaSet& alphas, int n_alpha,
                                             double* __restrict__ cost_out, int mode, const SolverParams& sp,
                                             int* __restrict__ commit_idx, int tile, double* lds_cost, bool count_running = true, int rwave = -1,
                                             typename M::real* share = nullptr) {
  using real = typename M::real;
  constexpr int NX = M::NX, NU = M::NU;
  static_assert(NG == 1 || (SHARE && GAINS && !ACCEPT), "several alpha groups per wavefront: the shared-row rollout of the wide tiles");
  static_assert(!CANDT || (CAND && SHARE && GAINS && NU == 1 && NX == 4 && PD % CG == 0), "grouped candidates: the matrix-core kernel's rollouts (nx = 4, nu = 1)");
  const int wave = (rwave >= 0) ? rwave : (int)(threadIdx.x >> 6);  // which four alphas this wavefront rolls out (>= 3: none)
  const int lane = threadIdx.x & 63;
  const int l = lane & (TW - 1);
  const int a_sub = lane >> 4;
  const int b = tile * TW + l;
  // NG alpha groups per wavefront (the wide tiles' NG = 3: every lane carries the rollouts of alphas a_sub, 4 + a_sub, 8 + a_sub of its
  // trajectory, step by step side by side): the nominal rows of a step are fetched and passed through LDS once for the three, and three
  // independent chains per lane fill each other's dependent-issue gaps.  Each rollout is the expression sequence of NG = 1: same bits.
  int a[NG];
  bool active[NG];
  bool any_active = false;
  const bool eligible = (b < v.B) && (mode != 1 || (v.status[b < v.B ? b : 0] == 0 && v.backpass_done[b < v.B ? b : 0]));
#pragma unroll
  for (int g = 0; g < NG; g++) {
    a[g] = (wave + g) * 4 + a_sub;
    active[g] = eligible && (a[g] < n_alpha);
    any_active = any_active || active[g];
  }
  // SHARE: every lane of a wavefront with work fetches its rows and runs the steps (an alpha group without an alpha, the
  // lanes of a finished trajectory: their arithmetic is discarded); `active` only decides who stores
  const bool run = (SHARE && GAINS) ? (__ballot(any_active) != 0ull) : any_active;
  if (!ACCEPT && !run) return;
  if (run) {
  const int T = v.T;
  real alpha[NG];
#pragma unroll
  for (int g = 0; g < NG; g++) alpha[g] = (real)alphas.a[a[g] < NALPHA ? a[g] : NALPHA - 1];
  const real dt = (real)v.dt;

  real x[NG][NX];
#pragma unroll
  for (int i = 0; i < NX; i++) {
    const real x0i = v.x0[tidx(tile, 0, i, l, 1, NX)];
#pragma unroll
    for (int g = 0; g < NG; g++) x[g][i] = x0i;
  }
  double total[NG];  // (the sum over the horizon is a per-trajectory accumulator: double in both modes, common.hpp)
#pragma unroll
  for (int g = 0; g < NG; g++) total[g] = 0;

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
  // SHARE && CAND: a lane without a rollout of its own (an alpha group beyond the last alpha, a finished trajectory) stores too,
  // into the spare plane behind the last alpha's (the buffers hold NALPHA + 1): no predicate around the stores of every step
  int ta_store[NG];
#pragma unroll
  for (int g = 0; g < NG; g++) ta_store[g] = ((SHARE && CAND && !active[g]) ? NALPHA : a[g]) * v.ntiles + tile;
  // knot t = (x_t, u_t); with_u: a step's call (t < T by construction: no test in the step), false for the final state
  // ph: t mod CG, a compile-time value in the unrolled loops (CANDT gathers the controls of CG steps into one store)
  typedef real real4v __attribute__((ext_vector_type(4)));
  real pend[NG][CG - 1];  // CANDT: the controls of the group's first steps, waiting for its last one
#pragma unroll
  for (int g = 0; g < NG; g++)
#pragma unroll
    for (int q = 0; q < CG - 1; q++) pend[g][q] = 0;
  auto store_group = [&](int g, int t, real last) __attribute__((always_inline)) {  // the group of step t, its last slot = `last`
    real4v w;
    w.x = pend[g][0];
    w.y = pend[g][1];
    w.z = pend[g][2];
    w.w = last;
    *reinterpret_cast<real4v*>(v.cand_u + cand_g_u(ta_store[g], t / CG, l, T)) = w;
  };
  auto emit_knot = [&](int g, int t, const real* xx, const real* uu, auto with_u, auto ph) __attribute__((always_inline)) {
    if (SHARE && !CAND && !active[g]) return;
    if constexpr (CANDT) {
      constexpr int p = decltype(ph)::value;
      if (with_u) {
        if constexpr (p == CG - 1)
          store_group(g, t, uu[0]);
        else
          pend[g][p] = uu[0];
      }
      if ((t & (CT - 1)) == 0) {
        real4v w;
        w.x = xx[0];
        w.y = xx[1];
        w.z = xx[2];
        w.w = xx[3];
        *reinterpret_cast<real4v*>(v.cand_x + cand_g_x(ta_store[g], t / CT, l, v.nch, NX)) = w;
      }
    } else if (CAND) {
      const int ta = ta_store[g];
      if (with_u) {
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
      if (with_u) {
#pragma unroll
        for (int q = 0; q < NU; q++) v.us[tidx(tile, t, q, l, T, NU)] = uu[q];  // :323 (no clamping)
      }
    }
  };
  const WithTrigConsts<M> rmodel(model);  // (models.hpp: the trig constants of the dynamics in registers once, not once per step)
  auto do_step = [&](int t, const StepIn& d, auto ph) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < NG; g++) {
      real u[NU];
#pragma unroll
      for (int j = 0; j < NU; j++) u[j] = d.u[j];
      if (GAINS) {
#pragma unroll
        for (int j = 0; j < NU; j++) {
          u[j] += d.k[j] * alpha[g];  // :190
          real acc = 0;
#pragma unroll
          for (int i = 0; i < NX; i++) acc += d.K[j + NU * i] * (x[g][i] - d.xnom[i]);
          u[j] += acc;  // :316
        }
      }
      if (!NOFIX && (sp.fixes & 1)) {  // opt-in fix: "the right way" of ilqr_core.cpp:327-329 -- the clamped control is stored and integrated
#pragma unroll
        for (int j = 0; j < NU; j++) u[j] = min_of(max_of(u[j], model.u_min[j]), model.u_max[j]);
      }
      emit_knot(g, t, x[g], u, std::true_type(), ph);
      total[g] += (double)model.cost(x[g], u);  // :324
      real x1[NX];
      integrate_dynamics(rmodel, x[g], u, dt, x1);  // :325
#pragma unroll
      for (int i = 0; i < NX; i++) x[g][i] = x1[i];
    }
  };
  if constexpr (SHARE && GAINS) {
    static_assert(PD % 2 == 0 || PD == 1, "static ring indices");
    static_assert(PD <= kRolloutFetchSlack, "the arrays end in kRolloutFetchSlack spare rows");
    constexpr int NROWS = 2 * NU + NU * NX + NX;  // u, k, K, xs
    constexpr int NLD = (NROWS + 3) / 4;          // rows per alpha group
    struct Raw {
      real r[NLD];
    };
    // this lane's rows: a_sub, a_sub + 4, ... (past the end: the last row again, parked in a dummy LDS row)
    const real* rbase[NLD];
    unsigned rstride[NLD];  // elements from step t to step t + 1
#pragma unroll
    for (int j = 0; j < NLD; j++) {
      const int row = (a_sub + 4 * j < NROWS) ? a_sub + 4 * j : NROWS - 1;
      if (row < NU) {
        rbase[j] = v.us + tidx(tile, 0, row, l, T, NU);
        rstride[j] = NU * TW;
      } else if (row < 2 * NU) {
        rbase[j] = v.kff + tidx(tile, 0, row - NU, l, T, NU);
        rstride[j] = NU * TW;
      } else if (row < 2 * NU + NU * NX) {
        rbase[j] = v.Kfb + tidx(tile, 0, row - 2 * NU, l, T, NU * NX);
        rstride[j] = NU * NX * TW;
      } else {
        rbase[j] = v.xs + tidx(tile, 0, row - 2 * NU - NU * NX, l, T + 1, NX);
        rstride[j] = NX * TW;
      }
    }
    auto fetch = [&](int tt, Raw& d) __attribute__((always_inline)) {
      // tail (tt >= T): rows that no step consumes.  Not clamped to T - 1: the arrays end in kRolloutFetchSlack spare rows (ilqr_create), and without
      // the clamp the row addresses are plain induction variables -- three pointer increments per step instead of min / multiply / shift / add
      // (seven instructions of a step's ~145)
#if !ILQR_ROLLOUT_UNCLAMPED_FETCH
      tt = (tt < T) ? tt : T - 1;
#endif
#pragma unroll
      for (int j = 0; j < NLD; j++) d.r[j] = rbase[j][(size_t)tt * rstride[j]];
    };
    // LDS layout of a step's rows: [row pair][trajectory][2] -- a lane reads rows 2 q and 2 q + 1 of its trajectory with ONE 16-byte
    // (fp32: 8-byte) LDS instruction, five per step for the acrobot's ten rows instead of ten
    typedef real real2_t __attribute__((ext_vector_type(2)));
    constexpr int NPAIR = (NROWS + 1) / 2;
    real* const mine = share + (((a_sub >> 1) * TW + l) * 2 + (a_sub & 1));  // row a_sub + 4 j: 4 j TW further on
    auto put = [&](const Raw& d) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NLD; j++) mine[4 * j * TW] = d.r[j];
    };
    const real2_t* const row0 = reinterpret_cast<const real2_t*>(share + 2 * l);
    auto get = [&](StepIn& d) __attribute__((always_inline)) {
      real rows[2 * NPAIR];
#pragma unroll
      for (int q = 0; q < NPAIR; q++) {
        const real2_t pr = row0[q * TW];
        rows[2 * q] = pr.x;
        rows[2 * q + 1] = pr.y;
      }
#pragma unroll
      for (int j = 0; j < NU; j++) d.u[j] = rows[j];
#pragma unroll
      for (int j = 0; j < NU; j++) d.k[j] = rows[NU + j];
#pragma unroll
      for (int e = 0; e < NU * NX; e++) d.K[e] = rows[2 * NU + e];
#pragma unroll
      for (int i = 0; i < NX; i++) d.xnom[i] = rows[2 * NU + NU * NX + i];
    };
    Raw ring[PD];
#pragma unroll
    for (int d = 0; d < PD; d++) fetch(d, ring[d]);
    StepIn cur, nxt;
    put(ring[0]);
    get(cur);
    // step s: pass step s + 1 through LDS (its rows are in ring[(s + 1) % PD]), refill ring[s % PD] -- free since step
    // s - 1 -- with step s + PD, compute step s from `cur`.  LDS executes a wavefront's operations in order: the reads
    // of step s + 1 follow its writes, and the next writes follow those reads.
    auto one = [&](int s_, Raw& free_set, const Raw& next_set, auto ph) __attribute__((always_inline)) {
      put(next_set);
      get(nxt);
      fetch(s_ + PD, free_set);
      do_step(s_, cur, ph);
      cur = nxt;
    };
    // (t stays a multiple of PD, and CANDT has PD a multiple of CG: (t + d) mod CG is d mod CG -- a compile-time value)
    int t = 0;
    for (; t + PD <= T; t += PD) {
      static_for<PD>([&](auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        one(t + d, ring[d], ring[(d + 1) % PD], std::integral_constant<int, d % CG>());
      });
    }
    static_for<PD>([&](auto dc) __attribute__((always_inline)) {  // remainder (< PD steps; t is a multiple of PD)
      constexpr int d = decltype(dc)::value;
      if (t + d < T) one(t + d, ring[d], ring[(d + 1) % PD], std::integral_constant<int, d % CG>());
    });
  } else {
  StepIn ring[PD];
#pragma unroll
  for (int d = 0; d < PD; d++) load_step(d, ring[d]);
  int t = 0;
  for (; t + PD <= T; t += PD) {
#pragma unroll
    for (int d = 0; d < PD; d++) {
      do_step(t + d, ring[d], std::integral_constant<int, 0>());   // (the set is consumed in place and refilled right after: copying it out first so
      load_step(t + d + PD, ring[d]);               //  that the refill could be issued a step earlier cost ten register moves per step)
    }
  }
  for (; t < T; t++) {  // remainder (< PD steps)
    StepIn cur;
    load_step(t, cur);
    do_step(t, cur, std::integral_constant<int, 0>());
  }
  }
#pragma unroll
  for (int g = 0; g < NG; g++) {
    {  // knot T: the final state (no control)
      real uz[NU];
#pragma unroll
      for (int q = 0; q < NU; q++) uz[q] = 0;
      emit_knot(g, T, x[g], uz, std::false_type(), std::integral_constant<int, 0>());
      if constexpr (CANDT) {
        if (T % CG) store_group(g, T - 1, pend[g][CG - 2]);  // a last, partial group (its slots past T - 1 are never used for anything that is stored)
      }
    }
    total[g] += (double)model.final_cost(x[g]);  // :335
    if (active[g]) {
      cost_out[(size_t)a[g] * v.Bp + b] = total[g];
      if (ACCEPT) lds_cost[a[g] * TW + l] = total[g];
    }
  }
  }  // if (run)
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
// CANDT: the candidates lie in the grouped layout k_solve_hex's rollouts leave (cand_g_u / cand_g_x above; nu = 1)
template <class M, bool CANDT = false>
__device__ __forceinline__ void candidate_knot(const BatchViewT<typename M::real>& v, const M& model, int a, int tile, int t, int l,
                                               typename M::real* x, typename M::real* u) {
  using real = typename M::real;
  constexpr int NX = M::NX, NU = M::NU;
  const int T = v.T, ta = a * v.ntiles + tile, c = t / CT, off = t - c * CT;
#pragma unroll
  for (int i = 0; i < NX; i++) x[i] = CANDT ? v.cand_x[cand_g_x(ta, c, l, v.nch, NX) + i] : v.cand_x[tidx(ta, c, i, l, v.nch, NX)];
  real uq[CT][NU];
#pragma unroll
  for (int q = 0; q < CT; q++) {
    const int tq = (c * CT + q < T) ? c * CT + q : T - 1;
#pragma unroll
    for (int j = 0; j < NU; j++) uq[q][j] = CANDT ? v.cand_u[cand_g_u(ta, tq / CG, l, T) + tq % CG] : v.cand_u[tidx(ta, tq, j, l, T, NU)];
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
template <class M, bool CANDT = false>
__global__ void k_unpack_cand(BatchViewT<typename M::real> v, M model, int a, double* __restrict__ xs, double* __restrict__ us) {
  using real = typename M::real;
  constexpr int NX = M::NX, NU = M::NU;
  const int T = v.T;
  const size_t n = (size_t)v.B * (T + 1);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % (T + 1));
    const int b = (int)(i / (T + 1));
    real x[NX], u[NU];
    candidate_knot<M, CANDT>(v, model, a, b / TW, t, b % TW, x, u);
    if (xs)
      for (int e = 0; e < NX; e++) xs[((size_t)b * (T + 1) + t) * NX + e] = (double)x[e];
    if (us && t < T)
      for (int e = 0; e < NU; e++) us[((size_t)b * T + t) * NU + e] = (double)u[e];
  }
}

}  // namespace ilqr
