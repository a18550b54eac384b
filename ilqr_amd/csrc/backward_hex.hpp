// backward_hex.hpp -- the backward pass with SIXTEEN lanes per trajectory (nx = 4, nu = 1), for batches that leave every CU one
// 16-trajectory tile (B <= 16 x #CU: BASELINE's metric batch).
//
// There the chip is short of nothing but the LATENCY of one Riccati chain per tile: backward_quad runs it on one wavefront
// (4 lanes per trajectory) at ~430 instructions per step, one every ~5 cycles, while three SIMDs of the CU mostly wait.  Here
// the tile is split into four SUB-TILES of four trajectories; each gets a chain wavefront of its own, on its own SIMD, with
// one lane per ELEMENT of the 4 x 4 quantities (lane = 16 lp + 4 a + b: trajectory lp of the sub-tile, element (a, b)):
//   * a matrix product costs four FMAs per lane instead of sixteen; rows / columns of the operands that live in other lanes
//     are exchanged through a few hundred bytes of LDS (one ds_write + two ds_read_b128 per operand -- a wavefront's LDS
//     operations execute in order, so no barrier and no DPP chains: an fp64 quad gather is 8 v_mov_dpp);
//   * the scalar box-QP is still evaluated redundantly by the lanes of a trajectory, but its Armijo search tests SIXTEEN
//     step sizes 0.6^k at once (k = 0..15 in one pass: exactly the reference's sequential loop, boxqp.cpp:156-173, first
//     passing k wins) instead of a four-lane window around an fp32 estimate of the answer;
//   * each chain wavefront has its OWN producer wavefront (same SIMD, lower priority) and its own LDS ring of derivative
//     records: four independent (chain, producer) pairs per tile, each the protocol of solve_tile.hpp with one producer of
//     16 knots x 4 trajectories per round.  Nothing is shared between pairs, so no new cross-wavefront protocol exists.
// Every element is computed by the expression, in the order, that backward_quad uses for it, and the search returns what the
// sequential loop returns: the route leaves the bits of every other route (tests/test_gpu_fused_sweep.py, scripts/soak.py).
#pragma once
#include "solve_tile.hpp"

namespace ilqr {

constexpr int HT = 4;  // trajectories per sub-tile (= per chain wavefront)
#ifdef ILQR_HEX_DEBUG
__device__ long long g_hex_dbg[16];
#define HEX_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_amdgcn_s_memtime(); dbg[i] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HEX_STAMP(i) do {} while (0)
#endif

// One pair's ring: a slot holds one knot of the sub-tile's 4 trajectories, pair-interleaved like the HBM records:
// [pair][trajectory lp][2].
template <int NX, int NU, class real, int SLOTS_>
struct HexRing {
  static constexpr int US = Rec<NX, NU>::SIZE;
  static constexpr int PAIRS = (Rec<NX, NU>::SIZE + NU + 1) / 2;
  static constexpr int ROW = 2 * HT;             // `real`s from one element pair to the next
  static constexpr int PAD = ROW - 2 * TW;       // what derivatives_of_knot adds to its 16-trajectory row (negative: a narrower row)
  static constexpr int ELEMS = PAIRS * ROW;
  static constexpr int SLOTS = SLOTS_;
};

constexpr int kHexKnotsPerRound = 64 / HT;       // a producer round: 16 knots x 4 trajectories
constexpr int kHexSlots = 24;
constexpr int kHexLead = 7;                      // a round may be written once its first knot is at most this far ahead of the consumer
static_assert(kHexSlots > kHexLead + kHexKnotsPerRound - 1, "a round must not reach a slot the chain has not left");

template <class real, int NX, int NU>
struct HexPair {  // LDS of one (chain, producer) pair
  using RS = HexRing<NX, NU, real, kHexSlots>;
  real ring[RS::SLOTS * RS::ELEMS];
  real xch[(16 + 4 + 16 + 4) * HT];   // the chain wavefront's exchange buffers: W', V fu, Vxx, Vx of its 4 trajectories
  int rounds_done;                    // rounds (counted across passes) whose records are in the ring
  int consumer_at;                    // running index of the knot the chain waits for (everything below is consumed)
  int passes_started;                 // backward passes begun; -1 once the chain is through
  unsigned long long pass_lanes;      // exec mask of the chain wavefront in the current pass (bit 16 lp = trajectory lp)
};

// Consumer side (cf. RingGate / WideGate).  consumer_at is published at EVERY knot (one LDS store, nothing to wait for): with
// 16 knots per producer round the producer must be released in the middle of a round, not at its end.  No release fence: the
// LDS executes a wavefront's operations in order, so the reads of every knot below G were executed before this store is.
template <class P>
struct HexGate {
  static constexpr bool kRing = true;
  P& sh;
  const int T, nrounds, N;
  int pass = -1, have = 0;
  int g_next = 0, slot_next = 0, slot_cur = 0;
  __device__ __forceinline__ HexGate(P& s, int T_) : sh(s), T(T_), nrounds((T_ + 1 + kHexKnotsPerRound - 1) / kHexKnotsPerRound), N(nrounds * kHexKnotsPerRound) {}
  __device__ __forceinline__ void begin_pass() {
    pass++;
    have = pass * N;
    g_next = pass * N;
    slot_next = g_next % P::RS::SLOTS;
    __hip_atomic_store(&sh.pass_lanes, __ballot(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&sh.passes_started, pass + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ int slot(int) const { return slot_cur; }
  __device__ __forceinline__ void wait(int) {
    const int G = g_next++;
    slot_cur = slot_next;
    slot_next = (slot_next + 1 == P::RS::SLOTS) ? 0 : slot_next + 1;
    __hip_atomic_store(&sh.consumer_at, G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (G < have) return;
    const int round = pass * nrounds + (G - pass * N) / kHexKnotsPerRound;
    while (__hip_atomic_load(&sh.rounds_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= round) __builtin_amdgcn_s_sleep(1);
    have = pass * N + ((G - pass * N) / kHexKnotsPerRound + 1) * kHexKnotsPerRound;
  }
  __device__ __forceinline__ void finish() {
    __hip_atomic_store(&sh.consumer_at, 0x3fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(&sh.passes_started, -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
};

// The Armijo backtracking loop of the scalar box-QP (boxqp.cpp:156-173; qp1_backtrack_seq is its literal form) by the sixteen
// lanes of a trajectory: lane j evaluates step 0.6^(base + j) with the loop's own expressions; the loop would have stopped at
// the FIRST k whose trial passes the test -- or, before that, at the first k >= 1 whose step is below minStep or whose trial
// lands on x itself (failure: the shortcut of qp1_backtrack_seq, the outcome of the reference's ~100 trips).  A trial on x has
// value == old value and fails the test, so "passes" and "stuck" never meet on one k and the order of the two is the loop's.
// The winner's step comes back from the table and every lane recomputes the winner's trial point and value (same expression,
// same operands: the winner's bits).  step_j0 = table[j] (loop-invariant, in a register); later rounds read the table.
template <class real>
__device__ __forceinline__ void qp1_search_hex(QP1StateT<real>& q, int j, int lane, real step_j0, const real* __restrict__ lds_steps) {
  if (q.early) return;  // (row-uniform; the caller's selects ignore x1 / v1 of an early exit)
  real my_step = step_j0;
  for (int base = 0;; base += 16) {
    const int k = base + j;
    const real my_x1 = qp1_trial(q, my_step);
    const real my_v1 = qp1_value(q, my_x1);
    const bool pass = !qp1_armijo_fails(q, my_v1, my_step);
    const bool stuck = p_and(k >= 1, p_or(my_step < real(kMinStep), my_x1 == q.x));
    const unsigned int P = (unsigned int)(__ballot(pass) >> (lane & 48)) & 0xFFFFu;
    const unsigned int S = (unsigned int)(__ballot(stuck) >> (lane & 48)) & 0xFFFFu;
    const int fp = __ffs(P), fs = __ffs(S);  // 1-based, 0 = none
    if (fp != 0 && (fs == 0 || fp < fs)) {
      q.step = lds_steps[base + fp - 1];
      q.x1 = qp1_trial(q, q.step);
      q.v1 = qp1_value(q, q.x1);
      return;
    }
    if (fs != 0) {
      q.ls_failed = true;
      return;
    }
    const int kn = k + 16;
    my_step = lds_steps[kn < 104 ? kn : 103];  // (k = 100 is below minStep: the loop never gets further)
  }
}

template <class real>
struct HexStep {  // what lane (lp, a, b) needs of one derivative record, as loaded
  typedef real pair_t __attribute__((ext_vector_type(2)));
  pair_t Fa[2], Fb[2], fu[2], tail, uw;   // fx[:, a], fx[:, b], fu, (cu, cuu), (us, 1 / (|us| + 1))
  real cxb, cxx_ab, cxx_ba, cxu_a, cxu_b;
};

// One sub-tile's backward pass: run by ONE wavefront, lane = 16 lp + 4 a + b.  Records from the pair's ring.
template <class M, class Gate, class RS>
__device__ __forceinline__ void backward_hex(const BatchViewT<typename M::real>& v, const M& model, const SolverParams& sp, int mode, int tile, int sub,
                                             int lane, const typename M::real* __restrict__ lds_steps, Gate& gate, const typename M::real* __restrict__ ring,
                                             typename M::real* __restrict__ xch) {
  using real = typename M::real;
  static_assert(M::NX == 4 && M::NU == 1, "hex chain: nx = 4, nu = 1");
  using R = Rec<4, 1>;
  typedef real real2_t __attribute__((ext_vector_type(2)));
  typedef real real4_t __attribute__((ext_vector_type(4)));
  const int lp = lane >> 4, a = (lane >> 2) & 3, bq = lane & 3, j16 = lane & 15;
  const int l = sub * HT + lp;
  const int b = tile * TW + l;
  if (b >= v.B) return;                        // row-uniform
  if (mode == 1 && v.status[b] != 0) return;   // row-uniform
  const int T = v.T;
  double lambda = v.lambda[b], dlambda = v.dlambda[b];
  const real* __restrict__ ust = v.us + tidx(tile, 0, 0, l, T, 1);
  real* __restrict__ kt = v.kff + tidx(tile, 0, 0, l, T, 1);
  real* __restrict__ Kt = v.Kfb + tidx(tile, 0, 0, l, T, 4);
  typedef const __attribute__((address_space(3))) real lds_cd;
  typedef const __attribute__((address_space(3))) real2_t lds_cd2;
  typedef const __attribute__((address_space(3))) real4_t lds_cd4;
  typedef __attribute__((address_space(3))) real lds_d;
  // exchange buffers of this wavefront (element (r, c) of trajectory lp at [16 lp + 4 r + c])
  lds_d* xW = (lds_d*)(xch + 16 * lp);                  // W' : xW[4 c + r] = W[r, c]  (column c contiguous)
  lds_d* xwv = (lds_d*)(xch + 16 * HT + 4 * lp);        // V fu
  lds_d* xV = (lds_d*)(xch + 20 * HT + 16 * lp);        // Vxx, row-major
  lds_d* xVx = (lds_d*)(xch + 36 * HT + 4 * lp);        // Vx
  const real step_j0 = lds_steps[j16];

  auto load = [&](int t, HexStep<real>& d) __attribute__((always_inline)) {
    gate.wait(t);
    lds_cd* r = (lds_cd*)(ring + gate.slot(t) * RS::ELEMS + lp * 2);
    auto pair = [&](int e) { return *(lds_cd2*)(r + (e >> 1) * RS::ROW); };
    auto one = [&](int e) { return r[(e >> 1) * RS::ROW + (e & 1)]; };
    d.Fa[0] = pair(R::FX + 4 * a);
    d.Fa[1] = pair(R::FX + 4 * a + 2);
    d.Fb[0] = pair(R::FX + 4 * bq);
    d.Fb[1] = pair(R::FX + 4 * bq + 2);
    d.fu[0] = pair(R::FU);
    d.fu[1] = pair(R::FU + 2);
    d.tail = pair(R::CU);
    d.uw = pair(RS::US);
    d.cxb = one(R::CX + bq);
    d.cxx_ab = one(R::CXX + a + 4 * bq);
    d.cxx_ba = one(R::CXX + bq + 4 * a);
    d.cxu_a = one(R::CXU + a);
    d.cxu_b = one(R::CXU + bq);
  };

  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0, gacc = 0;
#ifdef ILQR_HEX_DEBUG
  long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif
  auto one_pass = [&]() __attribute__((always_inline)) {
    gate.begin_pass();
    real Vr[4], Vx[4], kprev;   // row a of Vxx' (= column a: the matrix is symmetric to the bit), Vx'
    const real lam_r = (real)lambda;
    {
      gate.wait(T);
      lds_cd* r = (lds_cd*)(ring + gate.slot(T) * RS::ELEMS + lp * 2);
#pragma unroll
      for (int i = 0; i < 4; i++) Vx[i] = r[((R::CX + i) >> 1) * RS::ROW + ((R::CX + i) & 1)];  // :353
#pragma unroll
      for (int q = 0; q < 4; q++) Vr[q] = r[((R::CXX + a + 4 * q) >> 1) * RS::ROW + ((R::CXX + a + 4 * q) & 1)];  // :354
    }
    kprev = kt[(unsigned)((T - 1) * TW)];
    dV0 = dV1 = 0;
    diverge = 0;
    gacc = 0;
    real* __restrict__ Kt_i = Kt + (unsigned)(((T - 1) * 4 + bq) * TW);  // K[b] of step i
    real* __restrict__ kt_i = kt + (unsigned)((T - 1) * TW);

    // one Riccati step from the record `raw` of knot i; `nxt` is loaded with knot i - 1 on the way
    auto step = [&](int i, const HexStep<real>& raw, HexStep<real>& nxt) -> bool {
      HEX_STAMP(0);
      const real Fa[4] = {raw.Fa[0].x, raw.Fa[0].y, raw.Fa[1].x, raw.Fa[1].y};
      const real Fb[4] = {raw.Fb[0].x, raw.Fb[0].y, raw.Fb[1].x, raw.Fb[1].y};
      const real fu[4] = {raw.fu[0].x, raw.fu[0].y, raw.fu[1].x, raw.fu[1].y};
      const real cu = raw.tail.x, cuu = raw.tail.y, us = raw.uw.x, usw = raw.uw.y;
      // W[a, b] = (Vxx' fx)[a, b]                                   (backward_quad: W[r] of lane s, r = a, s = b)
      real Wab;
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += Vr[q] * Fb[q];
        Wab = acc;
      }
      // (Vxx' fu)[a]
      real wva;
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += Vr[q] * fu[q];
        wva = acc;
      }
      xW[4 * bq + a] = Wab;
      xwv[a] = wva;
      // while the exchange is in flight: Qu, Qx[b], the box      :360, :359
      real Qu, Qxb;
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fu[q] * Vx[q];
        Qu = cu + acc;
      }
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += Fb[q] * Vx[q];
        Qxb = raw.cxb + acc;
      }
      const real lo = model.u_min[0] - us, hi = model.u_max[0] - us;
      __builtin_amdgcn_sched_barrier(0);
      const real4_t wcb4 = *(lds_cd4*)(xW + 4 * bq), wca4 = *(lds_cd4*)(xW + 4 * a), wv4 = *(lds_cd4*)(xwv);
      const real Wcb[4] = {wcb4.x, wcb4.y, wcb4.z, wcb4.w};  // W[:, b]
      const real Wca[4] = {wca4.x, wca4.y, wca4.z, wca4.w};  // W[:, a]
      const real wv[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
      HEX_STAMP(1);
      // Qxx[a, b], Qxx[b, a]      :361
      real Qxx_ab, Qxx_ba, Qux_a, Qux_b;
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += Fa[q] * Wcb[q];
        Qxx_ab = raw.cxx_ab + acc;
      }
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += Fb[q] * Wca[q];
        Qxx_ba = raw.cxx_ba + acc;
      }
      // Qux[a], Qux[b]            :362/:366
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fu[q] * Wca[q];
        Qux_a = raw.cxu_a + acc;
      }
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fu[q] * Wcb[q];
        Qux_b = raw.cxu_b + acc;
      }
      // Quu, QuuF                 :363, :367
      real Quu, QuuF;
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fu[q] * wv[q];
        Quu = cuu + acc;
        QuuF = (cuu + lam_r) + acc;
      }
      // the next step's record: issued here, it lands under the box-QP
      __builtin_amdgcn_sched_barrier(0);
      HEX_STAMP(2);
      if (i >= 1) load(i - 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      HEX_STAMP(3);
      // :369  box-QP, replicated over the 16 lanes; 16 step sizes per pass of the Armijo search
      real x;
      int free0;
      real minv;
      QP1StateT<real> q1;
      qp1_begin<false>(QuuF, Qu, kprev, lo, hi, q1, false);
      qp1_search_hex(q1, j16, lane, step_j0, lds_steps);
      bool goes_on;
      bool ok = qp1_finish_ok(q1, x, free0, minv, goes_on);
      if (goes_on)
        ok = qp1_continue(
                 q1,
                 [&](QP1StateT<real>& qs) __attribute__((always_inline)) {
                   qp1_search_hex(qs, j16, lane, step_j0, lds_steps);
                 },
                 x, free0) >= 1;
      if (!ok) diverge = i;
      HEX_STAMP(4);
      // :373-385  K = -(R^-1 R^-T) Qux on a free control, 0 on a clamped one
      const real k_scale = free0 ? -minv : real(0);
      const real K_a = k_scale * Qux_a, K_b = k_scale * Qux_b;
      // :388-389
      {
        real d0 = 0;
        d0 += x * Qu;
        if (ok) dV0 += (double)d0;
        real rq = 0;
        rq += (real(0.5) * x) * Quu;
        real d1 = 0;
        d1 += rq * x;
        if (ok) dV1 += (double)d1;
      }
      // :391-393
      real T1_a, T1_b;
      {
        real acc = 0;
        acc += K_a * Quu;
        T1_a = acc;
      }
      {
        real acc = 0;
        acc += K_b * Quu;
        T1_b = acc;
      }
      real Vxn_b, Vn_ab, Vn_ba;
      {
        real t1 = 0, t2 = 0, t3 = 0;
        t1 += T1_b * x;
        t2 += K_b * Qu;
        t3 += Qux_b * x;
        Vxn_b = ((Qxb + t1) + t2) + t3;
      }
      {
        real u1 = 0, u2 = 0, u3 = 0;
        u1 += T1_a * K_b;
        u2 += K_a * Qux_b;
        u3 += Qux_a * K_b;
        Vn_ab = ((Qxx_ab + u1) + u2) + u3;
      }
      {
        real u1 = 0, u2 = 0, u3 = 0;
        u1 += T1_b * K_a;
        u2 += K_b * Qux_a;
        u3 += Qux_b * K_a;
        Vn_ba = ((Qxx_ba + u1) + u2) + u3;
      }
      // 0.5 (V + V'): on the diagonal the two are the same value and 0.5 (x + x) = x exactly
      const real Vsym = real(0.5) * (Vn_ab + Vn_ba);
      // :405-412 term of the gradient norm; :396-397 the gains
      {
        const real mx = abs_of(x) * usw;
        if (ok) gacc += (double)mx;
      }
      if (ok) {
        kprev = x;
        if (a == 0) Kt_i[0] = K_b;
        if (j16 == 0) kt_i[0] = x;
      }
      Kt_i -= 4 * TW;
      kt_i -= TW;
      // row a of the new Vxx and the new Vx sit in this lane's own quad: DPP broadcasts (an LDS round trip costs several hundred cycles here)
      quad_gather(Vsym, Vr);
      quad_gather(Vxn_b, Vx);
      HEX_STAMP(5);
      return ok;
    };

    HexStep<real> A, Bd;
    int i = T - 1;
    load(i, A);
#ifdef ILQR_HEX_DEBUG
    tlast = __builtin_amdgcn_s_memtime();
#endif
    while (true) {
      if (!step(i, A, Bd)) break;
      if (--i < 0) break;
      if (!step(i, Bd, A)) break;
      if (--i < 0) break;
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
      for (int jj = 0; jj < 8; jj++) {
        const int t = (t0 + jj < T) ? t0 + jj : T - 1;
        kv[jj] = kt[(size_t)t * TW];
        uv[jj] = ust[(size_t)t * TW];
      }
#pragma unroll
      for (int jj = 0; jj < 8; jj++) {
        const real mx = abs_of(kv[jj]) / (abs_of(uv[jj]) + 1);
        if (t0 + jj < T) acc += (double)mx;
      }
    }
  }
  const double gnorm = acc / T;
#ifdef ILQR_HEX_DEBUG
  if (tile == 0 && sub == 0 && lane == 0)
    for (int q = 0; q < 6; q++) g_hex_dbg[q] += dbg[q];
#endif
  if (j16 == 0) {
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

// STEP 1 + STEP 2 of one iteration for one tile as four (chain, producer) pairs.  role: 0..3 chain of pair `role`,
// 4..7 producer of pair `role - 4`.  The nominal trajectory is complete in HBM (the accepted candidates were committed
// right after the line search: commit_tile_chunks), so the producers read knots and nothing else.
template <class M, class MFD, class P>
__device__ __forceinline__ void sweep_backward_hex(const BatchViewT<typename M::real>& v, const M& model, const MFD& fdm, const SolverParams& sp, int mode,
                                                   int force, int tile, P* pairs, const typename M::real* __restrict__ lds_steps, int role) {
  using RS = typename P::RS;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 4) {
    pairs[threadIdx.x].rounds_done = 0;
    pairs[threadIdx.x].consumer_at = 0;
    pairs[threadIdx.x].passes_started = 0;
  }
  __syncthreads();
  const int T = v.T;
  P& sh = pairs[role & 3];
  if (role < 4) {
    __builtin_amdgcn_s_setprio(3);
    HexGate<P> gate(sh, T);
    backward_hex<M, decltype(gate), RS>(v, model, sp, mode, tile, role, lane, lds_steps, gate, sh.ring, sh.xch);
    gate.finish();
    __builtin_amdgcn_s_setprio(0);
  } else {
    const int sub = role - 4;
    const int lp = lane & (HT - 1), ks = lane >> 2;  // trajectory of the sub-tile, knot of the round
    const int l = sub * HT + lp;
    const int nrounds = (T + 1 + kHexKnotsPerRound - 1) / kHexKnotsPerRound, N = nrounds * kHexKnotsPerRound;
    for (int pass = 0;; pass++) {
      unsigned long long lanes = ~0ull;
      if (pass > 0) {  // a retry pass exists only if the chain starts one
        int started;
        while ((started = __hip_atomic_load(&sh.passes_started, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= 0 && started <= pass)
          __builtin_amdgcn_s_sleep(8);
        if (started < 0) break;
        lanes = __hip_atomic_load(&sh.pass_lanes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      const bool mine = (lanes >> (16 * lp)) & 1ull;  // does trajectory lp take part in this pass?
      for (int r = 0; r < nrounds; r++) {
        const int j0 = r * kHexKnotsPerRound, G0 = pass * N + j0;
        while (G0 > __hip_atomic_load(&sh.consumer_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + kHexLead) __builtin_amdgcn_s_sleep(4);
        const int started = __hip_atomic_load(&sh.passes_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((started < 0) | (started > pass + 1)) break;  // the chain has left this pass behind
        const int t = T - (j0 + ks);
#ifdef ILQR_HEX_NOPROD
        if (t >= 0 && (pass == 0 || mine) && v.status[0] == 12345)  // experiment: the producers publish rounds without computing them
#else
        if (t >= 0 && (pass == 0 || mine))
#endif
          derivatives_of_knot<M, true, MFD, RS::PAD>(v, model, fdm, force, nullptr, tile, t, l, sh.ring + ((G0 + ks) % RS::SLOTS) * RS::ELEMS + lp * 2, true);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the round's LDS writes are done
        if (lane == 0) __hip_atomic_store(&sh.rounds_done, pass * nrounds + r + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
}

// The commit of the accepted candidates (ilqr_core.cpp:210-213) of one tile, by the whole block right after the line search:
// a candidate is stored as its controls and every CT-th state (common.hpp); every chunk of CT knots is integrated forward
// once from its checkpoint -- the steps candidate_knot takes, in its order: the same bits -- and written to xs / us.
// 16 trajectories x (T / CT + 1) chunks over the block's threads: a few microseconds per iteration.
template <class M>
__device__ __forceinline__ void commit_tile_chunks(const BatchViewT<typename M::real>& v, const M& model, const int* __restrict__ commit_idx, int tile) {
  using real = typename M::real;
  constexpr int NX = M::NX, NU = M::NU;
  const int T = v.T;
  const real dt = (real)v.dt;
  const int ntask = v.nch * TW;
  for (int task = threadIdx.x; task < ntask; task += blockDim.x) {
    const int l = task & (TW - 1), c = task >> 4;
    const int b = tile * TW + l;
    const int ci = (b < v.B) ? commit_idx[b] : -1;
    if (ci < 0) continue;
    const int ta = ci * v.ntiles + tile;
    real x[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) x[i] = v.cand_x[tidx(ta, c, i, l, v.nch, NX)];
#pragma unroll
    for (int q = 0; q < CT; q++) {
      const int t = c * CT + q;
      if (t > T) break;
#pragma unroll
      for (int i = 0; i < NX; i++) v.xs[tidx(tile, t, i, l, T + 1, NX)] = x[i];
      if (t < T) {
        real u[NU];
#pragma unroll
        for (int jj = 0; jj < NU; jj++) {
          u[jj] = v.cand_u[tidx(ta, t, jj, l, T, NU)];
          v.us[tidx(tile, t, jj, l, T, NU)] = u[jj];
        }
        if (q + 1 < CT) {
          real x1[NX];
          integrate_dynamics(model, x, u, dt, x1);
#pragma unroll
          for (int i = 0; i < NX; i++) x[i] = x1[i];
        }
      }
    }
  }
}

// Whole iterations for ONE tile (see k_solve_tile), the backward pass as four hex chains.
//   grid = ntiles, block = 512 = 8 wavefronts, two per SIMD: (chain, producer) of pair = SIMD id; one block per CU
template <class M, class MFD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_solve_hex(BatchViewT<typename M::real> v, M model, MFD fdm, AlphaSet alphas,
                                                                                                 SolverParams sp, int n_iters, int force,
                                                                                                 int* __restrict__ commit_idx, int commit_pending,
                                                                                                 long long* __restrict__ phase_ticks) {
  using real = typename M::real;
  using P = HexPair<real, M::NX, M::NU>;
  __shared__ P pairs[4];
  __shared__ real steps[104];
  __shared__ double lds_cost[NALPHA * TW];
  __shared__ int tile_running;
  __shared__ int simd_count[4];
  if (threadIdx.x < 4) simd_count[threadIdx.x] = 0;
  load_step_table(steps);  // (barrier)
  // Roles by SIMD: the first wavefront of the block that reports from SIMD s runs the chain of pair s, the second its
  // producer -- a chain shares its SIMD with the one wavefront that feeds it.  (Should the dispatcher ever place the eight
  // wavefronts otherwise, roles go by wavefront index: correct, only slower.)
  const int wave = (int)(threadIdx.x >> 6);
  int role = wave;
  {
    const int simd = hw_simd_id();
    int slot = 0;
    if ((threadIdx.x & 63) == 0) slot = atomicAdd(&simd_count[simd], 1);
    slot = __shfl(slot, 0, 64);
    __syncthreads();
    if (simd_count[0] == 2 && simd_count[1] == 2 && simd_count[2] == 2 && simd_count[3] == 2) role = simd + 4 * slot;
  }
  const int rwave = (role < 3) ? role : 3;  // the chains of pairs 0..2 roll out (one per SIMD); rwave 3 has no alphas
  const int tile = blockIdx.x;
  constexpr int kShareReals = 4 * ((2 * M::NU + M::NU * M::NX + M::NX + 3) / 4) * TW;
  static_assert(kShareReals <= (int)(sizeof(pairs[0].ring) / sizeof(real)), "a pair's ring holds a wavefront's rollout rows");
  long long t_sweep = 0, t_roll = 0, t0 = 0;
  const bool timing = (phase_ticks != nullptr) & (threadIdx.x == 0);
  const long long c_begin = timing ? clock64() : 0, w_begin = timing ? wall_clock64() : 0;
  if (commit_pending) {  // accepts of an earlier launch that nobody has copied yet
    commit_tile_chunks<M>(v, model, commit_idx, tile);
    phase_barrier();
  }
  int it = 0;
  for (; it < n_iters; it++) {
    if (timing) t0 = wall_clock64();
    sweep_backward_hex<M, MFD, P>(v, model, fdm, sp, 1, force, tile, pairs, steps, role);
    phase_barrier();  // the tile's gains, lambda, status are in memory for its rollout wavefronts
    if (timing) {
      const long long t1 = wall_clock64();
      t_sweep += t1 - t0;
      t0 = t1;
    }
    rollout_tile<M, true, true, 4, true, true>(v, model, alphas, NALPHA, v.cost_c, 1, sp, commit_idx, tile, lds_cost, /*count_running=*/it == n_iters - 1, rwave,
                                               pairs[role & 3].ring);
    if (threadIdx.x == 0) tile_running = 0;
    phase_barrier();  // candidates, costs, status, commit indices are in memory
    commit_tile_chunks<M>(v, model, commit_idx, tile);
    phase_barrier();  // the nominal trajectory is the accepted one
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
    phase_ticks[5 * tile + 3] += clock64() - c_begin;
    phase_ticks[5 * tile + 4] += wall_clock64() - w_begin;
  }
}

}  // namespace ilqr
