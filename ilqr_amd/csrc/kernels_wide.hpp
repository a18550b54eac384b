// kernels_wide.hpp -- persistent WIDE tiles for batches that saturate the device (nx = 4, nu = 1).
//
// k_solve_tile (kernels.hpp) gives every 16-trajectory tile a backward wavefront of its own: four lanes per trajectory,
// the scalar box-QP evaluated redundantly by the four -- the shortest chain for a tile that has a CU to itself.  Once the
// batch holds several tiles per CU the chip is short of ISSUE SLOTS, not of latency, and that layout spends them badly: the
// box-QP and the loop around it are ~300 of the ~530 instructions of a step, issued once per 16 trajectories.  Here a
// WIDE tile is 64 trajectories (four tiles of the HBM layout) and its backward wavefront runs ONE THREAD PER TRAJECTORY:
// the same ~300 instructions now serve 64 box-QPs, the 4 x 4 algebra (every lane the whole matrices, no DPP exchange)
// costs ~260 more -- about 9 instead of 33 wave-instructions per trajectory-timestep for the backward pass.
//
//   block = 8 wavefronts, one block per CU (one 64-trajectory ring slot is 24 KB in fp64: six slots fill the LDS)
//   phase 1: wavefront 0 = the chain (backward_wide), wavefronts 1..3 = producers, one knot x 64 trajectories each per round,
//            wavefront 4 = the pending commit of the accepted candidates, one chunk of CT knots at a time (commit_chunks_wide)
//   phase 2: 12 rollout units (4 tiles x 3 alpha groups of rollout_tile, nominal rows shared through the idle ring) over
//            the 8 wavefronts, then accept_one x 64
//
// Every element is computed by the expression, in the order, that backward_quad uses for it (that kernel is a column split
// of the same arithmetic), and the single-lane Armijo search evaluates the quad search's four candidates with the same
// functions: the two routes leave the same bits (tests/test_gpu_fused_sweep.py), which is what lets a batch-size
// threshold choose between them -- and eight 4096-trajectory shards equal one 32768-trajectory batch.
#pragma once
#include "kernels.hpp"

namespace ilqr {

constexpr int WT = 64;  // trajectories per wide tile = 4 tiles of TW

template <int NX, int NU, class real, int RING_KB = 148>
struct WideRing {
  static constexpr int US = Rec<NX, NU>::SIZE;                 // the control (and, m = 1, the gradient-norm weight) follow the record
  static constexpr int PAIRS = (Rec<NX, NU>::SIZE + NU + 1) / 2;
  static constexpr int ROW = 2 * WT;                           // `real`s from one element pair to the next
  static constexpr int PAD = ROW - 2 * TW;                     // what derivatives_of_knot adds to its 16-trajectory row
  static constexpr int ELEMS = PAIRS * ROW;
  static constexpr int SLOTS = (RING_KB * 1024 / (int)sizeof(real)) / ELEMS;  // 148 KB: fp64 6, fp32 12; 74 KB: 3, 6
};

template <class real, int NX, int NU, int kProd, int RING_KB = 148>
struct WideShared {
  using RS = WideRing<NX, NU, real, RING_KB>;
  double steps[104];  // (the chain runs in double for every handle, backward_quad.hpp)
  alignas(16) real ring[RS::SLOTS * RS::ELEMS];
  int rounds_done[kProd];
  int consumer_at;
  int passes_started;
  int committed_to;               // knots >= this one hold the accepted candidate (commit_chunks_wide)
  unsigned long long pass_lanes;  // bit l = trajectory l of the wide tile takes part in the current pass
};

// The consumer side of the wide ring (cf. RingGate): one knot per producer and round.  The chain would pay an LDS round
// trip per step if it asked "is MY knot there" (it is a different producer's every step); it asks all producers at once
// and remembers the frontier -- the first knot not yet produced -- so that with the producers their usual few knots ahead
// the question comes up once per few steps.  consumer_at is published at every knot (a store, nothing to wait for): the
// release keeps every read of the knots below it ahead of the moment their slots may be overwritten.
template <class SH, int kProd>
struct WideGate {
  static constexpr bool kRing = true;
  SH& sh;
  const int T, nrounds, N;
  int pass = -1, have = 0;
  int g_next = 0, slot_next = 0, slot_cur = 0;
  __device__ __forceinline__ WideGate(SH& s, int T_) : sh(s), T(T_), nrounds((T_ + 1 + kProd - 1) / kProd), N(nrounds * kProd) {}
  __device__ __forceinline__ void begin_pass() {
    pass++;
    have = pass * N;
    g_next = pass * N;
    slot_next = g_next % SH::RS::SLOTS;
    __hip_atomic_store(&sh.pass_lanes, __ballot(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&sh.passes_started, pass + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ int slot(int) const { return slot_cur; }
  __device__ __forceinline__ void wait(int) {
    const int G = g_next++;
    slot_cur = slot_next;
    slot_next = (slot_next + 1 == SH::RS::SLOTS) ? 0 : slot_next + 1;
    __hip_atomic_store(&sh.consumer_at, G, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (G < have) return;
    const int j = G - pass * N;
    while (true) {
      int J = 0x3fffffff;
#pragma unroll
      for (int w = 0; w < kProd; w++) {
        int cw = __hip_atomic_load(&sh.rounds_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - pass * nrounds;
        cw = cw < 0 ? 0 : cw;
        const int first_missing = cw * kProd + w;  // producer w's knots are j = r kProd + w
        J = first_missing < J ? first_missing : J;
      }
      if (J > j) {
        have = pass * N + J;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __device__ __forceinline__ void finish() {
    __hip_atomic_store(&sh.consumer_at, 0x3fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&sh.passes_started, -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
};

// qp1_search_quad for ONE lane: the same four candidates (the unit step and the window k1, k1+1, k1+2 around the fp32
// estimate), each with the exact expressions, then the same decision -- evaluated one after the other instead of by the four
// lanes of a quad.  Leaves q.x1 / q.v1 / q.step / q.ls_failed and returns exactly what the quad search does.
template <class real>
__device__ __forceinline__ bool qp1_search_lane(QP1StateT<real>& q, const real* __restrict__ lds_steps) {
  const real bound = (q.search > 0) ? q.hi : q.lo;
  const real v_b = qp1_value(q, bound);
  const float f = (float)(bound - q.x) * __builtin_amdgcn_rcpf((float)q.search);
  const float r = (float)(v_b - q.old_v) * __builtin_amdgcn_rcpf((float)(real(kArmijo) * q.slope));
  const float thr = fmaxf(f, r);
  const int kg = (int)ceilf(__log2f(thr) * -1.35691545f);
  const bool sane = p_and(p_and(q.Q != real(0), p_and(thr > 0.f, thr < 1.f)), p_and(kg >= 1, kg <= 96));
  const int k1 = p_and(sane, kg > 2) ? kg - 1 : 1;
  real cx[4], cv[4], cs[4];
  unsigned int m4 = 0u, s4 = 0u;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const int my_k = (c == 0) ? 0 : k1 + c - 1;
    cs[c] = lds_steps[my_k];
    cx[c] = qp1_trial(q, cs[c]);
    cv[c] = qp1_value(q, cx[c]);
    m4 |= (!qp1_armijo_fails(q, cv[c], cs[c]) ? 1u : 0u) << c;
    s4 |= ((cx[c] == q.x) ? 1u : 0u) << c;
  }
  const bool unit = (m4 & 1u) != 0u;
  const unsigned int w = m4 >> 1;
  const int wwin = __ffs(w);
  const bool wok = p_and(p_and(w != 0u, p_or(wwin > 1, k1 == 1)), p_or(sane, k1 == 1));
  const bool ok = p_or(unit, wok);
  const int src = ok ? (unit ? 0 : wwin) : 0;
  q.x1 = (src == 1) ? cx[1] : (src == 2) ? cx[2] : (src == 3) ? cx[3] : cx[0];
  q.v1 = (src == 1) ? cv[1] : (src == 2) ? cv[2] : (src == 3) ? cv[3] : cv[0];
  q.step = (src == 1) ? cs[1] : (src == 2) ? cs[2] : (src == 3) ? cs[3] : cs[0];
  const bool stuck = p_and((s4 & 1u) != 0u, !q.early);  // (the unit-step trial is candidate 0)
  const bool dead = p_and(p_and(k1 == 1, m4 == 0u), p_and((s4 & 8u) != 0u, !q.early));
  q.ls_failed = p_or(q.ls_failed, p_or(stuck, dead));
  return p_or(p_or(ok, q.early), p_or(stuck, dead));
}

// The backward pass of one wide tile, one thread per trajectory (lane = trajectory of the wide tile), records from the ring.
// Structure and arithmetic of backward_quad (kernels.hpp) with the column index s as a loop instead of a lane.
template <class M, class Gate, class RS>
__device__ __forceinline__ void backward_wide(const BatchViewT<typename M::real>& v, const M& model, const SolverParams& sp, int mode, int wtile,
                                              int lane, const double* __restrict__ lds_steps, Gate& gate,
                                              const typename M::real* __restrict__ ring) {
  using real = typename M::real;   // what is stored per knot
  using creal = double;            // what the recursion computes in (backward_quad.hpp: the mixed mode of fp32 handles)
  static_assert(M::NX == 4 && M::NU == 1, "wide tiles: nx = 4, nu = 1");
  using R = Rec<4, 1>;
  typedef real real2_t __attribute__((ext_vector_type(2)));
  const int tile = wtile * (WT / TW) + (lane >> 4), l = lane & (TW - 1);
  const int b = tile * TW + l;
  if (b >= v.B) return;
  if (mode == 1 && v.status[b] != 0) return;
  const int T = v.T;
  double lambda = v.lambda[b], dlambda = v.dlambda[b];
  const real* __restrict__ ust = v.us + tidx(tile, 0, 0, l, T, 1);
  real* __restrict__ kt = v.kff + tidx(tile, 0, 0, l, T, 1);
  real* __restrict__ Kt = v.Kfb + tidx(tile, 0, 0, l, T, 4);
  typedef const __attribute__((address_space(3))) real lds_cd;
  typedef const __attribute__((address_space(3))) real2_t lds_cd2;

  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0, gacc = 0;
  auto one_pass = [&]() __attribute__((always_inline)) {
    gate.begin_pass();
    creal Vx[4], Vxx[16], kprev;
    const creal lam_r = (creal)lambda;
    {
      gate.wait(T);
      lds_cd* r = (lds_cd*)(ring + gate.slot(T) * RS::ELEMS + lane * 2);
#pragma unroll
      for (int i = 0; i < 4; i++) Vx[i] = r[((R::CX + i) >> 1) * RS::ROW + ((R::CX + i) & 1)];  // :353
#pragma unroll
      for (int e = 0; e < 16; e++) Vxx[e] = r[((R::CXX + e) >> 1) * RS::ROW + ((R::CXX + e) & 1)];  // :354
    }
    kprev = kt[(size_t)(T - 1) * TW];
    // (gfx950 counts stores in vmcnt too: with this load still "in flight" at the loop header the compiler made every step wait for
    //  vmcnt(0) before its first use of kprev -- i.e. for the previous step's gain stores to be acknowledged by the L2, a few hundred
    //  cycles of a 1500-cycle step.  Waited for here, once per pass, no step waits for memory at all.)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    dV0 = dV1 = 0;
    diverge = 0;
    gacc = 0;
    // The record of a knot minus its cxx block (read column by column where it is used).  (Fetching the NEXT knot's in the
    // middle of a step, under the box-QP, was tried: the 96 extra live registers spill, in the rollout loop too -- 1.7 -> 2.6 ms.)
    struct KnotRec {
      real fx[16], fu[4], cx[4], cxu[4], cu, cuu, us, usw;
      int slot;
    };
    auto fetch = [&](int t, KnotRec& d) __attribute__((always_inline)) {
      gate.wait(t);
      d.slot = gate.slot(t);
      lds_cd* r = (lds_cd*)(ring + d.slot * RS::ELEMS + lane * 2);
      auto pair = [&](int e) { return *(lds_cd2*)(r + (e >> 1) * RS::ROW); };
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const real2_t p = pair(R::FX + e);
        d.fx[e] = p.x;
        d.fx[e + 1] = p.y;
      }
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const real2_t p = pair(R::FU + e), c = pair(R::CX + e), xu = pair(R::CXU + e);
        d.fu[e] = p.x;
        d.fu[e + 1] = p.y;
        d.cx[e] = c.x;
        d.cx[e + 1] = c.y;
        d.cxu[e] = xu.x;
        d.cxu[e + 1] = xu.y;
      }
      const real2_t tail = pair(R::CU);  // cu, cuu
      const real2_t uw = pair(RS::US);   // the knot's nominal control, 1 / (|u| + 1)
      d.cu = tail.x;
      d.cuu = tail.y;
      d.us = uw.x;
      d.usw = uw.y;
    };
    for (int i = T - 1; i >= 0; i--) {
      KnotRec cur;
      fetch(i, cur);
      lds_cd* r = (lds_cd*)(ring + cur.slot * RS::ELEMS + lane * 2);
      auto pair = [&](int e) { return *(lds_cd2*)(r + (e >> 1) * RS::ROW); };
      creal fx[16], fu[4], cx[4], cxu[4];
#pragma unroll
      for (int e = 0; e < 16; e++) fx[e] = cur.fx[e];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        fu[e] = cur.fu[e];
        cx[e] = cur.cx[e];
        cxu[e] = cur.cxu[e];
      }
      const creal cu = cur.cu, cuu = cur.cuu, us = cur.us, usw = cur.usw;
      // replicated in the quad kernel: Qu, wv = Vxx' fu, Quu, QuuF     :360, :363, :367
      creal Qu, Quu, QuuF;
      {
        creal acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fu[q] * Vx[q];
        Qu = cu + acc;
        creal wv[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          creal a2 = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) a2 += Vxx[rr + 4 * q] * fu[q];
          wv[rr] = a2;
        }
        creal a3 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) a3 += fu[q] * wv[q];
        Quu = cuu + a3;
        QuuF = (cuu + lam_r) + a3;
      }
      // the quad kernel's lane s, for s = 0..3: W = Vxx' fx[:, s];  Qxx[:, s], Qx[s], Qux[s]     :359, :361, :362
      creal Qxx[16], Qx[4], Qux[4];
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const real2_t c0 = pair(R::CXX + 4 * s), c1 = pair(R::CXX + 4 * s + 2);
        const creal cxxc[4] = {c0.x, c0.y, c1.x, c1.y};
        creal W[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += Vxx[rr + 4 * q] * fx[q + 4 * s];
          W[rr] = acc;
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += fx[q + 4 * rr] * W[q];
          Qxx[rr + 4 * s] = cxxc[rr] + acc;
        }
        {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += fx[q + 4 * s] * Vx[q];
          Qx[s] = cx[s] + acc;
        }
        {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += fu[q] * W[q];
          Qux[s] = cxu[s] + acc;
        }
      }
      // :369  box-QP
      const creal lo = model.u_min[0] - us, hi = model.u_max[0] - us;
      creal x;
      int free0;
      creal minv;
      QP1StateT<creal> q1;
      qp1_begin<false>(QuuF, Qu, kprev, lo, hi, q1, false);
      if (__builtin_expect(!qp1_search_lane(q1, lds_steps), 0)) {
        q1.step = 1;
        q1.x1 = qp1_trial(q1, creal(1));
        q1.v1 = qp1_value(q1, q1.x1);
        qp1_backtrack_seq(q1);
      }
      bool goes_on;
      bool ok = qp1_finish_ok(q1, x, free0, minv, goes_on);
      if (goes_on)
        ok = qp1_continue(
                 q1,
                 [&](QP1StateT<creal>& qs) __attribute__((always_inline)) {
                   if (__builtin_expect(!qp1_search_lane(qs, lds_steps), 0)) qp1_line_search_seq(qs);
                 },
                 x, free0) >= 1;
      if (!ok) diverge = i;
      creal K[4];  // :373-385
      const creal k_scale = free0 ? -minv : creal(0);
#pragma unroll
      for (int rr = 0; rr < 4; rr++) K[rr] = k_scale * Qux[rr];
      // :388-389
      {
        creal d0 = 0;
        d0 += x * Qu;
        if (ok) dV0 += (double)d0;
        creal rq = 0;
        rq += (creal(0.5) * x) * Quu;
        creal d1 = 0;
        d1 += rq * x;
        if (ok) dV1 += (double)d1;
      }
      // :391-393
      creal T1[4], Vxn[4], Vn[16];
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        creal acc = 0;
        acc += K[rr] * Quu;
        T1[rr] = acc;
      }
#pragma unroll
      for (int s = 0; s < 4; s++) {
        creal t1 = 0, t2 = 0, t3 = 0;
        t1 += T1[s] * x;
        t2 += K[s] * Qu;
        t3 += Qux[s] * x;
        Vxn[s] = ((Qx[s] + t1) + t2) + t3;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          creal u1 = 0, u2 = 0, u3 = 0;
          u1 += T1[rr] * K[s];
          u2 += K[rr] * Qux[s];
          u3 += Qux[rr] * K[s];
          Vn[rr + 4 * s] = ((Qxx[rr + 4 * s] + u1) + u2) + u3;
        }
      }
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        Vxx[rr + 4 * rr] = Vn[rr + 4 * rr];
#pragma unroll
        for (int c = rr + 1; c < 4; c++) {
          const creal sym = creal(0.5) * (Vn[rr + 4 * c] + Vn[c + 4 * rr]);
          Vxx[rr + 4 * c] = sym;
          Vxx[c + 4 * rr] = sym;
        }
        Vx[rr] = Vxn[rr];
      }
      // :405-412 term of the gradient norm
      {
        const creal mx = abs_of(x) * usw;
        if (ok) gacc += (double)mx;
      }
      // :396-397
      if (ok) {
        kprev = (creal)(real)x;  // the stored gain, as the reference reads k[i + 1] back (:369)
#pragma unroll
        for (int s = 0; s < 4; s++) Kt[(unsigned)((i * 4 + s) * TW)] = (real)K[s];
        kt[(unsigned)(i * TW)] = (real)x;
      }
      if (!ok) break;
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
      continue;
    }
    done = true;
    break;
  }

  double acc = gacc;
  if (!done) {  // an abandoned pass leaves a mix of old and new gains: re-read them (as backward_quad does)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    acc = 0;
    for (int t0 = 0; t0 < T; t0 += 8) {
      real kv[8], uv[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int t = (t0 + j < T) ? t0 + j : T - 1;
        kv[j] = kt[(size_t)t * TW];
        uv[j] = ust[(size_t)t * TW];
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const real mx = abs_of(kv[j]) / (abs_of(uv[j]) + 1);
        if (t0 + j < T) acc += (double)mx;
      }
    }
  }
  const double gnorm = acc / T;
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

// The pending commit of the accepted candidates (ilqr_core.cpp:210-213) for the 64 trajectories of a wide tile, by ONE
// wavefront that runs ahead of the producers.  A candidate is stored as its controls and every CT-th state
// (candidate_knot): knot t costs t mod CT integration steps, 3.5 on average -- as much as the knot's finite differences --
// when every knot is re-derived by itself, which is what a producer that meets a pending commit does in the 16-trajectory
// kernel (its producers have the time).  Here the producers share their SIMDs and were what the chain waited for (two
// tiles per CU: 6200 cycles per knot against the chain's 3500).  Integrating a chunk of CT knots forward ONCE gives the
// same states by the same sequence of steps (bit-identical) for 7 steps per 8 knots; they go to xs / us, committed_to
// tells the producers how far down the nominal trajectory is valid, and the producers read it as if nothing were pending.
template <class M>
__device__ __forceinline__ void commit_chunks_wide(const BatchViewT<typename M::real>& v, const M& model, const int* __restrict__ commit_idx,
                                                   int tile, int l, int* committed_to) {
  using real = typename M::real;
  constexpr int NX = M::NX, NU = M::NU;
  const int T = v.T, b = tile * TW + l;
  const int ci = (b < v.B) ? commit_idx[b] : -1;
  const int ta = (ci >= 0 ? ci : 0) * v.ntiles + tile;
  const real dt = (real)v.dt;
  struct Chunk {
    real x[NX], u[CT][NU];
  };
  auto fetch = [&](int c, Chunk& d) __attribute__((always_inline)) {
    if (ci < 0 || c < 0) return;
#pragma unroll
    for (int i = 0; i < NX; i++) d.x[i] = v.cand_x[tidx(ta, c, i, l, v.nch, NX)];
#pragma unroll
    for (int q = 0; q < CT; q++) {
      const int tq = (c * CT + q < T) ? c * CT + q : T - 1;
#pragma unroll
      for (int j = 0; j < NU; j++) d.u[q][j] = v.cand_u[tidx(ta, tq, j, l, T, NU)];
    }
  };
  auto run = [&](int c, Chunk& d) __attribute__((always_inline)) {
    if (ci >= 0) {
      real x[NX];
#pragma unroll
      for (int i = 0; i < NX; i++) x[i] = d.x[i];
#pragma unroll
      for (int q = 0; q < CT; q++) {
        const int t = c * CT + q;
        if (t <= T) {  // (wave-uniform)
#pragma unroll
          for (int i = 0; i < NX; i++) v.xs[tidx(tile, t, i, l, T + 1, NX)] = x[i];
          if (t < T) {
#pragma unroll
            for (int j = 0; j < NU; j++) v.us[tidx(tile, t, j, l, T, NU)] = d.u[q][j];
            if (q + 1 < CT) {
              real x1[NX];
              integrate_dynamics(model, x, d.u[q], dt, x1);
#pragma unroll
              for (int i = 0; i < NX; i++) x[i] = x1[i];
            }
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);  // the chunk's stores are out
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(committed_to, c * CT, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  Chunk A, Bc;  // two register sets: the next chunk's loads fly while this one integrates
  int c = T / CT;
  fetch(c, A);
  while (true) {
    fetch(c - 1, Bc);
    run(c, A);
    if (--c < 0) break;
    fetch(c - 1, A);
    run(c, Bc);
    if (--c < 0) break;
  }
}

template <class M, class Gate, class RS>  // (kernels_wide2.hpp: the chain for two controls)
__device__ __forceinline__ void backward_wide2(const BatchViewT<typename M::real>& v, const M& model, const SolverParams& sp, int mode, int wtile,
                                               int lane, Gate& gate, const typename M::real* __restrict__ ring);

// STEP 1 + STEP 2 of one iteration for one wide tile (see sweep_backward_tile): wavefront 0 the chain, 1..kProd producers
// (one knot x 64 trajectories per producer and round), wavefront kProd + 1 the pending commit, the rest of the block idle
// in this phase.
template <class M, int kProd, class MFD, class SH>
__device__ __forceinline__ void sweep_backward_wide(const BatchViewT<typename M::real>& v, const M& model, const MFD& fdm, const SolverParams& sp,
                                                    int mode, int force, const int* __restrict__ commit_idx, int wtile, SH& sh, int role = -1) {
  using RS = typename SH::RS;
  constexpr int kLeadKnots = RS::SLOTS - 1;  // a producer may write knot G once knot G - SLOTS has been consumed
  static_assert(RS::SLOTS >= kProd + 1, "the ring must hold the knots in production plus one");
  if (threadIdx.x < kProd) sh.rounds_done[threadIdx.x] = 0;
  if (threadIdx.x == kProd) {
    sh.consumer_at = 0;
    sh.passes_started = 0;
    sh.committed_to = (commit_idx != nullptr) ? v.T + 1 : 0;
  }
  __syncthreads();
  const int wave = (role >= 0) ? role : (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;  // 0 the chain, 1..kProd producers, else idle
  const int T = v.T;
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    WideGate<SH, kProd> gate(sh, T);
    if constexpr (M::NU == 1)
      backward_wide<M, decltype(gate), RS>(v, model, sp, mode, wtile, lane, sh.steps, gate, sh.ring);
    else
      backward_wide2<M, decltype(gate), RS>(v, model, sp, mode, wtile, lane, gate, sh.ring);
    gate.finish();
    __builtin_amdgcn_s_setprio(0);
  } else if (wave <= kProd) {
    __builtin_amdgcn_s_setprio(2);  // the chain waits for these: ahead of another tile's rollout wavefronts on the same SIMD
    const int w = wave - 1;
    const int tile = wtile * (WT / TW) + (lane >> 4), l = lane & (TW - 1);
    const int nrounds = (T + 1 + kProd - 1) / kProd, N = nrounds * kProd;
    for (int pass = 0;; pass++) {
      unsigned long long lanes = ~0ull;
      if (pass > 0) {
        int started;
        while ((started = __hip_atomic_load(&sh.passes_started, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= 0 && started <= pass)
          __builtin_amdgcn_s_sleep(8);
        if (started < 0) break;
        lanes = __hip_atomic_load(&sh.pass_lanes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      const bool mine = (lanes >> lane) & 1ull;
      for (int r = 0; r < nrounds; r++) {
        const int j = r * kProd + w, G = pass * N + j;
        while (G > __hip_atomic_load(&sh.consumer_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + kLeadKnots) __builtin_amdgcn_s_sleep(4);
        const int started = __hip_atomic_load(&sh.passes_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((started < 0) | (started > pass + 1)) break;  // the chain has left this pass behind
        const int t = T - j;
        if (t >= 0 && pass == 0 && commit_idx != nullptr)  // (the nominal knot t is the accepted candidate's from here on)
          while (__hip_atomic_load(&sh.committed_to, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) > t) __builtin_amdgcn_s_sleep(2);
        if (t >= 0 && (pass == 0 || mine))
          derivatives_of_knot<M, true, MFD, RS::PAD>(v, model, fdm, force, nullptr, tile, t, l,
                                                     sh.ring + (G % RS::SLOTS) * RS::ELEMS + lane * 2, true);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the round's LDS writes are done
        if (lane == 0) __hip_atomic_store(&sh.rounds_done[w], pass * nrounds + r + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  } else if (wave == kProd + 1 && commit_idx != nullptr) {
    __builtin_amdgcn_s_setprio(1);
    commit_chunks_wide<M>(v, model, commit_idx, wtile * (WT / TW) + (lane >> 4), lane & (TW - 1), &sh.committed_to);
    __builtin_amdgcn_s_setprio(0);
  }
}

// OCC = 1: one wide tile per CU -- 8 wavefronts, three producers, a 148 KB ring.
// OCC = 2: TWO wide tiles per CU, so that one tile's rollouts (12 units, the bulk of the instructions) run beside the other's
//          chain: 4 wavefronts each, two producers, a 74 KB ring (three slots in fp64), roles by SIMD as in k_solve_tile<.., 2>.
// OCC = 3: the two-control tile (kernels_wide2.hpp): one per CU, FOUR wavefronts (one per SIMD: 512 registers for the chain), two
//          producers, a 148 KB ring (four 30 KB slots in fp64).
template <int OCC>
struct WideCfg {
  static constexpr int kWaves = (OCC == 1) ? 8 : 4;
  static constexpr int kProd = (OCC == 1) ? 3 : 2;
  static constexpr int kRingKb = (OCC == 2) ? 74 : 148;
  // rollout prefetch depth (steps of nominal rows in flight per wavefront).  Measured, alternating builds on one box: with two
  // tiles per CU depth 8 is 7.35e9/s against 6.96e9/s at B = 32768, with one tile per CU depth 4 is 6.21e9/s against 5.97e9/s at
  // B = 16384 -- in both cases through phase 1 (the register allocation of the whole kernel moves), not through the rollouts.
  static constexpr int kPrefetch = (OCC == 2) ? 8 : 4;
  // alpha groups a rollout wavefront carries per lane (rollout.hpp, NG): with four wavefronts per tile each takes one 16-trajectory
  // tile whole; eight wavefronts keep the twelve (tile, alpha group) units
#ifndef ILQR_WIDE_ROLL_GROUPS
#define ILQR_WIDE_ROLL_GROUPS 3
#endif
#ifndef ILQR_WIDE1_ROLL_GROUPS
#define ILQR_WIDE1_ROLL_GROUPS 2
#endif
  static constexpr int kRollGroups = (OCC == 1) ? ILQR_WIDE1_ROLL_GROUPS : ILQR_WIDE_ROLL_GROUPS;
};

// Whole iterations for ONE wide tile (see k_solve_tile).  grid = ntiles / 4, block = 64 x kWaves.
template <class M, class MFD, int OCC = 1>
__global__ __launch_bounds__(64 * WideCfg<OCC>::kWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_solve_wide(
    BatchViewT<typename M::real> v, M model, MFD fdm, AlphaSet alphas, SolverParams sp, int n_iters, int force, int* __restrict__ commit_idx,
    int commit_pending, long long* __restrict__ phase_ticks) {
  using real = typename M::real;
  using Cfg = WideCfg<OCC>;
  constexpr int kWaves = Cfg::kWaves;
  __shared__ WideShared<real, M::NX, M::NU, Cfg::kProd, Cfg::kRingKb> sh;
  __shared__ int tile_running;
  __shared__ int simd_mask, chain_wave;
  if (threadIdx.x == 0) {
    simd_mask = 0;
    chain_wave = -1;
  }
  load_step_table(sh.steps);  // (barrier)
  const int wtile = blockIdx.x, wave = threadIdx.x >> 6;
  int role = wave, rwave = wave, roll_waves = kWaves;  // phase-1 role; index of this wavefront among the roll_waves that roll out
  if constexpr (OCC != 1) {  // roles by SIMD (k_solve_tile<.., 2>): chains of the two co-resident tiles on SIMD 0 and SIMD 2
    const int simd = hw_simd_id();
    if ((threadIdx.x & 63) == 0) atomicOr(&simd_mask, 1 << simd);
    const int chain_simd = (hw_lds_base() != 0) ? 2 : 0;
    if ((threadIdx.x & 63) == 0 && simd == chain_simd) chain_wave = wave;
    __syncthreads();
    if (simd_mask == 0xF && chain_wave >= 0) {
      const int rel = (simd - chain_simd) & 3;
      role = (rel == 0) ? 0 : (rel == 1) ? 1 : (rel == 3) ? 2 : 3;
      rwave = (rel == 1) ? 0 : (rel == 3) ? 1 : (rel == 0) ? 2 : 3;  // (keeping the wavefront on the OTHER tile's chain SIMD out of the rollouts: 6 % slower)
    }
  }
  // (the ring is idle while the tile rolls out: each wavefront's share of it passes the nominal rows between its alpha groups)
  constexpr int kShareReals = 4 * ((2 * M::NU + M::NU * M::NX + M::NX + 3) / 4) * TW;
  static_assert(kWaves * kShareReals <= (int)(sizeof(sh.ring) / sizeof(real)), "the ring holds every wavefront's rollout rows");
  real* const roll_share = sh.ring + wave * kShareReals;
  long long t_sweep = 0, t_roll = 0, t0 = 0;
  const bool timing = (phase_ticks != nullptr) & (threadIdx.x == 0);
  const long long c_begin = timing ? clock64() : 0, w_begin = timing ? wall_clock64() : 0;
  int it = 0;
  for (; it < n_iters; it++) {
    if (timing) t0 = wall_clock64();
    sweep_backward_wide<M, Cfg::kProd, MFD>(v, model, fdm, sp, 1, force, (it > 0 || commit_pending) ? commit_idx : nullptr, wtile, sh, role);
    phase_barrier();
    if (timing) {
      const long long t1 = wall_clock64();
      t_sweep += t1 - t0;
      t0 = t1;
    }
    // 12 rollout units (tile of the wide tile, alpha group) over the wavefronts that roll out.  (Handing units out
    // dynamically, as wavefronts become free, measured 3 % slower than this fixed assignment.)
    if constexpr (Cfg::kRollGroups == 3) {
      // four wavefronts, four 16-trajectory tiles: a wavefront rolls out ALL eleven alphas of its tile, three alpha groups per lane
      // (rollout.hpp, NG): the tile's nominal rows are fetched once instead of by three units at three different times
      for (int u = rwave; u < WT / TW; u += roll_waves)
        rollout_tile<M, true, true, Cfg::kPrefetch, false, true, true, 3>(v, model, alphas, NALPHA, v.cost_c, 1, sp, nullptr, wtile * (WT / TW) + u, nullptr, false, 0, roll_share);
    } else if constexpr (Cfg::kRollGroups == 2) {
      // eight wavefronts: wavefront u < 4 carries alpha groups 0 and 1 of tile u per lane, wavefront 4 + u alpha group 2 of tile u
      // (a SIMD holds one of each: the same three chains per SIMD as twelve single units, two of them in one instruction stream)
      for (int u = rwave; u < 2 * (WT / TW); u += roll_waves) {
        if (u < WT / TW)
          rollout_tile<M, true, true, Cfg::kPrefetch, false, true, true, 2>(v, model, alphas, NALPHA, v.cost_c, 1, sp, nullptr, wtile * (WT / TW) + u, nullptr, false, 0, roll_share);
        else
          rollout_tile<M, true, true, Cfg::kPrefetch, false, true, true, 1>(v, model, alphas, NALPHA, v.cost_c, 1, sp, nullptr, wtile * (WT / TW) + u - WT / TW, nullptr, false, 2, roll_share);
      }
    } else {
      for (int u = rwave; u < (WT / TW) * 3; u += roll_waves)
        rollout_tile<M, true, true, Cfg::kPrefetch, false, true, true>(v, model, alphas, NALPHA, v.cost_c, 1, sp, nullptr, wtile * (WT / TW) + u / 3, nullptr, false, u % 3, roll_share);
    }
    phase_barrier();  // the candidates' costs are in memory
    if (threadIdx.x < WT)
      accept_one(v, sp, wtile * WT + (int)threadIdx.x, [&](int a) { return v.cost_c[(size_t)a * v.Bp + wtile * WT + threadIdx.x]; }, commit_idx,
                 /*count_running=*/it == n_iters - 1);
    if (threadIdx.x == 0) tile_running = 0;
    phase_barrier();  // status, lambda, commit indices are in memory for the next sweep
    if (timing) t_roll += wall_clock64() - t0;
    if (!sp.fixed_work) {
      const int b = wtile * WT + (int)threadIdx.x;
      if (threadIdx.x < WT && b < v.B && v.status[b] == 0) tile_running = 1;
      __syncthreads();
      if (!tile_running) {
        it++;
        break;
      }
    }
  }
  if (timing) {
    const long long cyc = clock64() - c_begin, wall = wall_clock64() - w_begin;
    for (int q = 0; q < WT / TW; q++) {  // (the host averages over 16-trajectory tiles)
      long long* p = phase_ticks + 5 * (wtile * (WT / TW) + q);
      p[0] += t_sweep;
      p[1] += t_roll;
      p[2] += it;
      p[3] += cyc;
      p[4] += wall;
    }
  }
}

}  // namespace ilqr
