// backward_hex.hpp -- the backward pass of the metric batch (B <= 16 x #CU, nx = 4, nu = 1) on the MATRIX CORES:
// v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4 x 4 blocks per instruction, one element per lane -- the Riccati step's
// products for four trajectories at once, with the "row of one operand meets column of the other" traffic done inside the
// matrix unit.
//
// backward_quad runs one dependent chain per 16-trajectory tile on ONE wavefront: ~430 instructions per step, of which 64 are
// the FMAs of the 4 x 4 algebra and 48 the DPP moves that exchange columns, while three SIMDs of the CU wait.  Here the tile is
// four SUB-TILES of four trajectories; each has a chain wavefront of its own on its own SIMD (lane = 16 r + 4 t + c: element
// (r, c) of trajectory t).  A quantity "in D layout" (lane holds M[r][c]) is the matrix unit's result layout, IS its B operand
// layout, and read as the A operand it is the transpose -- and Vxx is symmetric to the bit.  So
//     W   = mfma(Vxx, fx)          Vxx' fx                          T  = mfma(fx, W)       fx' W      (Qxx  = cxx  + T)
//     Tt  = mfma(W, fx)            (fx' W)'                          (Qxx' = cxx' + Tt: the symmetrisation needs no exchange)
//     wv  = mfma(Vxx, R(fu))       Vxx' fu, replicated over c        Quu = cuu + mfma(R(fu), wv)      (every lane)
//     Qu  = cu + mfma(R(fu), R(Vx))                                   Qx[r] = cx[r] + mfma(fx, R(Vx))
//     Qux[c] = cxu[c] + mfma(R(fu), W)                                Qux[r] = cxu[r] + mfma(W, R(fu))
// (R(v): lane holds v[r]) -- nine matrix instructions, no DPP, no LDS exchange, a few dozen element-wise instructions, and the
// scalar box-QP replicated over the 16 lanes of a trajectory with SIXTEEN Armijo step sizes 0.6^k tested per pass (exactly the
// reference's sequential loop, boxqp.cpp:156-173: the first passing k wins).  The matrix unit accumulates k = 0..3 in order,
// one fused multiply-add per term from a zero accumulator (scripts/ubench/mfma4.hip) -- the sums of backward_quad, term for
// term -- so the route leaves the bits of every other route (tests/test_gpu_fused_sweep.py, scripts/soak.py).
// Each chain wavefront has its OWN producer wavefront (same SIMD, lower priority) and its own LDS ring of derivative records:
// four independent (chain, producer) pairs per tile, each the protocol of solve_tile.hpp with one producer of 16 knots x 4
// trajectories per round; the accepted candidates are committed right after the line search (commit_tile_chunks).
#pragma once
#include "solve_tile.hpp"

namespace ilqr {

#ifndef ILQR_COMMIT_ROLLED
#define ILQR_COMMIT_ROLLED 1
#endif

constexpr int HT = 4;  // trajectories per sub-tile (= per chain wavefront)
#ifndef ILQR_HEX_SLOT_PAD
#define ILQR_HEX_SLOT_PAD 8
#endif
#ifndef ILQR_HEX_PAIR_PERM
#define ILQR_HEX_PAIR_PERM 1
#endif
#ifndef ILQR_HEX_CAND_T
#define ILQR_HEX_CAND_T 1   // the kernel's candidates in groups per trajectory (rollout.hpp: CANDT); 0: one element per trajectory, the stage kernels' layout (A/B)
#endif
constexpr bool kHexCandT = ILQR_HEX_CAND_T != 0;

// One pair's ring: a slot holds one knot of the sub-tile's 4 trajectories, pair-interleaved like the HBM records:
// [pair][trajectory lp][2].
template <int NX, int NU, class real, int SLOTS_>
struct HexRing {
  static constexpr int US = Rec<NX, NU>::SIZE;
  static constexpr int PAIRS = (Rec<NX, NU>::SIZE + NU + 1) / 2;
  static constexpr int ROW = 2 * HT;             // `real`s from one element pair to the next
  static constexpr int PAD = ROW - 2 * TW;       // what derivatives_of_knot adds to its 16-trajectory row (negative: a narrower row)
  // SLOT_PAD: a producer round writes 16 knots = 16 consecutive slots with one store instruction per pair row; unpadded the slot
  // stride (1536 bytes in double) is a whole number of bank rows and the knots of a lane group land on the same banks (a 2-way
  // conflict on every one of a round's 24 stores); 8 `real`s further on each, the two (fp32: four) knots of a group sit side by side.
  static constexpr int ELEMS = PAIRS * ROW + ILQR_HEX_SLOT_PAD;
  static constexpr int SLOTS = SLOTS_;
  static constexpr bool PERM = (ILQR_HEX_PAIR_PERM != 0) && NX == 4 && NU == 1;
  static constexpr int pos(int pair) { return PERM ? hex_pair_pos(pair) : pair; }
};

constexpr int kHexKnotsPerRound = 64 / HT;       // a producer round: 16 knots x 4 trajectories
constexpr int kHexSlots = 24;
constexpr int kHexLead = 7;                      // a round may be written once its first knot is at most this far ahead of the consumer
static_assert(kHexSlots > kHexLead + kHexKnotsPerRound - 1, "a round must not reach a slot the chain has not left");

template <class real, int NX, int NU>
struct HexPair {  // LDS of one (chain, producer) pair
  using RS = HexRing<NX, NU, real, kHexSlots>;
  alignas(16) real ring[RS::SLOTS * RS::ELEMS];  // (16-byte aligned: the rollouts read it in row pairs, rollout.hpp)
  int rounds_done;                    // rounds (counted across passes) whose records are in the ring
  int consumer_at;                    // running index of the knot the chain waits for (everything below is consumed)
  int passes_started;                 // backward passes begun; -1 once the chain is through
  unsigned long long pass_lanes;      // exec mask of the chain wavefront in the current pass (bit 4 t = trajectory t)
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
// lanes of a trajectory: lane j = 4 r + c evaluates step 0.6^(base + j) with the loop's own expressions; the loop would have stopped
// at the FIRST k whose trial passes the test -- or, before that, at the first k >= 1 whose step is below minStep or whose trial
// lands on x itself (failure: the shortcut of qp1_backtrack_seq, the outcome of the reference's ~100 trips).  A trial on x has
// value == old value and fails the test, so "passes" and "stuck" never meet on one k and the order of the two is the loop's.
// The lanes of trajectory t sit at 16 r + 4 t + c, ascending with j: the lowest set bit of the ballot under the trajectory's lane
// mask IS the first k.  The winner's step comes back from the table and every lane recomputes the
// winner's trial point and value (same expression, same operands: the winner's bits).
// position 16 r + c of the first candidate of trajectory t whose bit is set in the wavefront's ballot; 63 (>= 52) if none.
// The trajectory's lanes are 16 r + 4 t + c, ascending with the candidate index j = 4 r + c: shifted down by 4 t they are the
// bits 0x000F000F000F000F, the highest of them 51 -- bit 63 is free to be the "none" mark.
__device__ __forceinline__ int hex_first(unsigned long long bal, int t4) {
  return __builtin_ctzll(((bal >> t4) & 0x000F000F000F000Full) | 0x8000000000000000ull);
}
__device__ __forceinline__ void qp1_search_hex(QP1StateT<double>& q, int j, int t4, double step_j0, const double* __restrict__ lds_steps) {
  // First pass, k = j, straight-line for every lane (an early exit's lanes compute along: the caller's selects ignore their
  // x1 / v1): the reference's loop ends inside it for all but a few per cent of the QPs.  A trial that lands on x itself ("stuck":
  // failure, qp1_backtrack_seq) has value == old value and fails the test, and so does every shorter step after it: a passing k
  // is never preceded by a stuck one, so the first set bit of the "passes" ballot is the loop's answer whenever there is one.
  const double t_x1 = clamp_of(q.x + step_j0 * q.search, q.lo, q.hi);   // qp1_trial
  const double t_v1 = qp1_value(q, t_x1);
  const bool t_pass = !qp1_armijo_fails(q, t_v1, step_j0);
  const int pp = hex_first(__ballot(t_pass), t4);
  const bool success = pp < 52;            // a set bit
  const int jw = success ? (((pp >> 4) << 2) | (pp & 3)) : 0;
#ifdef ILQR_EXP_NO_STEP_LDS  // timing experiment (wrong results for jw >= 2): what the step table's LDS round trip costs the chain
  q.step = jw == 0 ? 1.0 : 0.6;
#else
  q.step = lds_steps[jw];
#endif
  q.x1 = clamp_of(q.x + q.step * q.search, q.lo, q.hi);   // the winner's trial point and value, by the winner's expressions
  q.v1 = qp1_value(q, q.x1);
  bool more = p_and(!success, !q.early);
  if (__builtin_expect(__ballot(more) == 0ull, 1)) return;  // (wave-uniform)
  // No k <= 15 passes for some trajectory of this wavefront (the search direction is rounding noise: x is the optimum already).
  // The loop as written now needs its "stuck" and minStep exits (boxqp.cpp:167-171).  k = 1 .. 15 first, from the trial points
  // at hand (every step size here is above minStep): a stuck trial among them ends the search, no k <= 15 having passed.
  {
    const bool stuck = p_and(j >= 1, t_x1 == q.x);
    const int qs = hex_first(__ballot(p_and(stuck, more)), t4);
    if (p_and(more, qs < 52)) {
      q.ls_failed = true;
      more = false;
    }
    if (__ballot(more) == 0ull) return;
  }
  for (int base = 16;; base += 16) {  // 16 step sizes per pass
    const int kn = base + j;
    const double my_step = lds_steps[kn < 104 ? kn : 103];  // (k = 100 is below minStep: the loop never gets further)
    const double my_x1 = qp1_trial(q, my_step);
    const double my_v1 = qp1_value(q, my_x1);
    const bool pass = !qp1_armijo_fails(q, my_v1, my_step);
    const bool stuck = p_or(my_step < kMinStep, my_x1 == q.x);
    const int qp = hex_first(__ballot(p_and(pass, more)), t4), qs = hex_first(__ballot(p_and(stuck, more)), t4);
    if (more) {
      if (qp < qs) {
        q.step = lds_steps[base + (((qp >> 4) << 2) | (qp & 3))];
        q.x1 = qp1_trial(q, q.step);
        q.v1 = qp1_value(q, q.x1);
        more = false;
      } else if (qs < 52) {
        q.ls_failed = true;
        more = false;
      }
    }
    if (__ballot(more) == 0ull) return;
  }
}

// RS::pos for a pair index that depends on the lane (computed once per wavefront: the addresses are loop invariants)
template <class RS>
__device__ __forceinline__ int hex_pos_rt(int pair) { return RS::PERM ? hex_pair_pos(pair) : pair; }

template <class real>
struct HexStep {  // what lane (r, t, c) needs of one derivative record, as loaded (widened to the chain's arithmetic when used)
  typedef real pair_t __attribute__((ext_vector_type(2)));
  real F, cxx_rc, cxx_cr, fu_r, cx_r, cxu_r, cxu_c;   // fx[r][c], cxx[r][c], cxx[c][r], fu[r], cx[r], cxu[r], cxu[c]
  pair_t tail, uw;                                    // (cu, cuu), (us, 1 / (|us| + 1))
};

// One sub-tile's backward pass: run by ONE wavefront, lane = 16 r + 4 t + c.  Records from the pair's ring.
template <class M, class Gate, class RS>
__device__ __forceinline__ void backward_hex(const BatchViewT<typename M::real>& v, const M& model, const SolverParams& sp, int mode, int tile, int sub,
                                             int lane, const double* __restrict__ lds_steps, Gate& gate, const typename M::real* __restrict__ ring) {
  using real = typename M::real;   // what is stored per knot
  using creal = double;            // what the recursion computes in (backward_quad.hpp)
  static_assert(M::NX == 4 && M::NU == 1, "hex chain: nx = 4, nu = 1");
  using R = Rec<4, 1>;
  typedef real real2_t __attribute__((ext_vector_type(2)));
  const int r_ = lane >> 4, t_ = (lane >> 2) & 3, c_ = lane & 3, j16 = 4 * r_ + c_;
  const int t4 = 4 * t_;  // this trajectory's lanes in a ballot: 0x000F000F000F000F << t4
  const int l = sub * HT + t_;
  const int b = tile * TW + l;
  if (b >= v.B) return;                        // uniform over the trajectory's lanes
  if (mode == 1 && v.status[b] != 0) return;
  const int T = v.T;
  double lambda = v.lambda[b], dlambda = v.dlambda[b];
  const real* __restrict__ ust = v.us + tidx(tile, 0, 0, l, T, 1);
  real* __restrict__ kt = v.kff + tidx(tile, 0, 0, l, T, 1);
  real* __restrict__ Kt = v.Kfb + tidx(tile, 0, 0, l, T, 4);
  typedef const __attribute__((address_space(3))) real lds_cd;
  typedef const __attribute__((address_space(3))) real2_t lds_cd2;
  const creal step_j0 = lds_steps[j16];
  auto mm = [](creal a, creal bb) __attribute__((always_inline)) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, 0.0, 0, 0, 0); };

  auto load = [&](int t, HexStep<real>& d) __attribute__((always_inline)) {
    gate.wait(t);
    lds_cd* r = (lds_cd*)(ring + gate.slot(t) * RS::ELEMS + t_ * 2);
    auto pair = [&](int e) { return *(lds_cd2*)(r + RS::pos(e >> 1) * RS::ROW); };   // (constant e)
    auto one = [&](int e) { return r[hex_pos_rt<RS>(e >> 1) * RS::ROW + (e & 1)]; };  // (e depends on the lane)
    d.F = one(R::FX + r_ + 4 * c_);
    d.cxx_rc = one(R::CXX + r_ + 4 * c_);
    d.cxx_cr = one(R::CXX + c_ + 4 * r_);
    d.fu_r = one(R::FU + r_);
    d.cx_r = one(R::CX + r_);
    d.cxu_r = one(R::CXU + r_);
    d.cxu_c = one(R::CXU + c_);
    d.tail = pair(R::CU);
    d.uw = pair(RS::US);
  };

  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0, gacc = 0;
  auto one_pass = [&]() __attribute__((always_inline)) {
    gate.begin_pass();
    creal V, RVx, kprev;   // Vxx'[r][c] (D layout), Vx'[r] (replicated over c)
    const creal lam_r = (creal)lambda;
    {
      gate.wait(T);
      lds_cd* r = (lds_cd*)(ring + gate.slot(T) * RS::ELEMS + t_ * 2);
      RVx = (creal)r[hex_pos_rt<RS>((R::CX + r_) >> 1) * RS::ROW + ((R::CX + r_) & 1)];                        // :353
      V = (creal)r[hex_pos_rt<RS>((R::CXX + r_ + 4 * c_) >> 1) * RS::ROW + ((R::CXX + r_ + 4 * c_) & 1)];    // :354
    }
    kprev = (creal)kt[(unsigned)((T - 1) * TW)];
    // (gfx950 counts stores in vmcnt too: with this load still "in flight" at the loop header the compiler made every step wait for
    //  vmcnt(0) before its first use of kprev -- i.e. for the previous step's gain stores to be acknowledged by the L2, a few hundred
    //  cycles of a 1500-cycle step.  Waited for here, once per pass, no step waits for memory at all.)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    dV0 = dV1 = 0;
    diverge = 0;
    gacc = 0;
    real* __restrict__ Kt_i = Kt + (unsigned)(((T - 1) * 4 + c_) * TW);  // K[c] of step i
    real* __restrict__ kt_i = kt + (unsigned)((T - 1) * TW);

    // one Riccati step from the record `raw` of knot i; `nxt` is loaded with knot i - 1 on the way
    auto step = [&](int i, const HexStep<real>& raw, HexStep<real>& nxt) -> bool {
      const creal F = (creal)raw.F, Rfu = (creal)raw.fu_r;
      const creal cu = (creal)raw.tail.x, cuu = (creal)raw.tail.y, us = (creal)raw.uw.x, usw = (creal)raw.uw.y;
      // the matrix unit: every sum runs k = 0..3 from a zero accumulator, as backward_quad's do
      const creal W = mm(V, F);               // (Vxx' fx)[r][c]
      const creal wv = mm(V, Rfu);            // (Vxx' fu)[r]
      const creal aQu = mm(Rfu, RVx);         // fu' Vx'
      const creal aQx = mm(F, RVx);           // (fx' Vx')[r]
      const creal Tm = mm(F, W);              // (fx' W)[r][c]
      const creal Tt = mm(W, F);              // (fx' W)[c][r]
      const creal aQuu = mm(Rfu, wv);         // fu' Vxx' fu
      const creal aQux_c = mm(Rfu, W);        // (fu' W)[c]
      const creal aQux_r = mm(W, Rfu);        // (fu' W)[r]
      const creal Qxx_rc = (creal)raw.cxx_rc + Tm, Qxx_cr = (creal)raw.cxx_cr + Tt;   // :361
      const creal Qx_r = (creal)raw.cx_r + aQx;                                       // :359
      const creal Qu = cu + aQu;                                                      // :360
      const creal Qux_c = (creal)raw.cxu_c + aQux_c, Qux_r = (creal)raw.cxu_r + aQux_r;  // :362/:366
      const creal Quu = cuu + aQuu, QuuF = (cuu + lam_r) + aQuu;                      // :363, :367
      const creal lo = (creal)model.u_min[0] - us, hi = (creal)model.u_max[0] - us;
      // the next step's record: issued here, it lands under the box-QP
      __builtin_amdgcn_sched_barrier(0);
      if (i >= 1) load(i - 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      // :369  box-QP, replicated over the 16 lanes of the trajectory; 16 step sizes per pass of the Armijo search
      creal x;
      int free0;
      creal minv;
      QP1StateT<creal> q1;
      qp1_begin<false>(QuuF, Qu, kprev, lo, hi, q1, false);
      qp1_search_hex(q1, j16, t4, step_j0, lds_steps);
      bool goes_on;
      bool ok = qp1_finish_ok(q1, x, free0, minv, goes_on);
      if (goes_on)
        ok = qp1_continue(
                 q1, [&](QP1StateT<creal>& qs) __attribute__((always_inline)) { qp1_search_hex(qs, j16, t4, step_j0, lds_steps); }, x, free0) >= 1;
      if (!ok) diverge = i;
      // :373-385  K = -(R^-1 R^-T) Qux on a free control, 0 on a clamped one
      const creal k_scale = free0 ? -minv : creal(0);
      const creal K_r = k_scale * Qux_r, K_c = k_scale * Qux_c;
      // :388-389  (added to dV below, with everything else a successful step leaves behind: one predicated block)
      creal d0 = 0, d1 = 0;
      {
        d0 += x * Qu;
        creal rq = 0;
        rq += (creal(0.5) * x) * Quu;
        d1 += rq * x;
      }
      // :391-393
      creal T1_r, T1_c;
      {
        creal acc = 0;
        acc += K_r * Quu;
        T1_r = acc;
      }
      {
        creal acc = 0;
        acc += K_c * Quu;
        T1_c = acc;
      }
      creal Vxn_r, Vn_rc, Vn_cr;
      {
        creal t1 = 0, t2 = 0, t3 = 0;
        t1 += T1_r * x;
        t2 += K_r * Qu;
        t3 += Qux_r * x;
        Vxn_r = ((Qx_r + t1) + t2) + t3;
      }
      {
        creal u1 = 0, u2 = 0, u3 = 0;
        u1 += T1_r * K_c;
        u2 += K_r * Qux_c;
        u3 += Qux_r * K_c;
        Vn_rc = ((Qxx_rc + u1) + u2) + u3;
      }
      {
        creal u1 = 0, u2 = 0, u3 = 0;
        u1 += T1_c * K_r;
        u2 += K_c * Qux_r;
        u3 += Qux_c * K_r;
        Vn_cr = ((Qxx_cr + u1) + u2) + u3;
      }
      // 0.5 (V + V'): on the diagonal the two are the same value and 0.5 (x + x) = x exactly
      V = creal(0.5) * (Vn_rc + Vn_cr);
      RVx = Vxn_r;
      // :405-412 term of the gradient norm; :396-397 the gains
      if (ok) {
        dV0 += (double)d0;
        dV1 += (double)d1;
        gacc += (double)(abs_of(x) * usw);
        kprev = (creal)(real)x;  // the stored gain, as the reference reads k[i + 1] back (:369)
        // K[c] is the same bits on the four lanes r of (t, c), k on all sixteen of t (products of replicated operands: the matrix
        // unit runs the same sum for each): every lane stores -- four / sixteen writes of one value to one address -- and the
        // step has no lane predicates (each cost an exec save / restore around its store, their masks two v_readlane each)
        Kt_i[0] = (real)K_c;
        kt_i[0] = (real)x;
      }
      Kt_i -= 4 * TW;
      kt_i -= TW;
      return ok;
    };

    HexStep<real> A, Bd;
    int i = T - 1;
    load(i, A);
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
                                                   int force, int tile, P* pairs, const double* __restrict__ lds_steps, int role) {
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
    backward_hex<M, decltype(gate), RS>(v, model, sp, mode, tile, role, lane, lds_steps, gate, sh.ring);
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
      const bool mine = (lanes >> (4 * lp)) & 1ull;  // does trajectory lp take part in this pass?  (its chain lanes are 16 r + 4 lp + c)
      for (int r = 0; r < nrounds; r++) {
        const int j0 = r * kHexKnotsPerRound, G0 = pass * N + j0;
        while (G0 > __hip_atomic_load(&sh.consumer_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + kHexLead) __builtin_amdgcn_s_sleep(4);
        const int started = __hip_atomic_load(&sh.passes_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((started < 0) | (started > pass + 1)) break;  // the chain has left this pass behind
        const int t = T - (j0 + ks);
        if (t >= 0 && (pass == 0 || mine))
          derivatives_of_knot<M, true, MFD, RS::PAD, RS::PERM>(v, model, fdm, force, nullptr, tile, t, l, sh.ring + ((G0 + ks) % RS::SLOTS) * RS::ELEMS + lp * 2, true);
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
// CANDT: the candidates were left by this kernel's own rollouts, in groups per trajectory (rollout.hpp): a task reads its checkpoint as one
// 32-byte piece and its eight controls as two -- whole memory sectors of what it needs.  (false: the plane
// layout of every other producer of candidates -- the accepts of an earlier launch or of a stage call that this kernel finds pending.)
template <class M, bool CANDT = false>
__device__ __forceinline__ void commit_tile_chunks(const BatchViewT<typename M::real>& v, const M& model, const int* commit_of_lane /* [TW], LDS or global */, int tile) {
  using real = typename M::real;
  constexpr int NX = M::NX, NU = M::NU;
  static_assert(!CANDT || (NX == 4 && NU == 1 && CT == 8 && CG == 4), "grouped candidates: nx = 4, nu = 1");
  const int T = v.T;
  const real dt = (real)v.dt;
  const int ntask = v.nch * TW;
  for (int task = threadIdx.x; task < ntask; task += blockDim.x) {
    const int l = task & (TW - 1), c = task >> 4;
    const int b = tile * TW + l;
    const int ci = (b < v.B) ? commit_of_lane[l] : -1;
    if (ci < 0) continue;
    const int ta = ci * v.ntiles + tile;
    real x[NX], uq[CT][NU];  // the checkpoint and the chunk's controls: one memory round trip, then the steps
    if constexpr (CANDT) {
      typedef real real4v __attribute__((ext_vector_type(4)));
      const real4v xv = *reinterpret_cast<const real4v*>(v.cand_x + cand_g_x(ta, c, l, v.nch, NX));
      x[0] = xv.x;
      x[1] = xv.y;
      x[2] = xv.z;
      x[3] = xv.w;
      // (a plane row holds whole chunks past T - 1: the last chunk reads up to seven slots beyond it; they feed Euler steps whose
      //  results are never stored)
      const real4v u0 = *reinterpret_cast<const real4v*>(v.cand_u + cand_g_u(ta, 2 * c, l, T));
      const real4v u1 = *reinterpret_cast<const real4v*>(v.cand_u + cand_g_u(ta, 2 * c + 1, l, T));
      uq[0][0] = u0.x;
      uq[1][0] = u0.y;
      uq[2][0] = u0.z;
      uq[3][0] = u0.w;
      uq[4][0] = u1.x;
      uq[5][0] = u1.y;
      uq[6][0] = u1.z;
      uq[7][0] = u1.w;
    } else {
#pragma unroll
    for (int i = 0; i < NX; i++) x[i] = v.cand_x[tidx(ta, c, i, l, v.nch, NX)];
#pragma unroll
    for (int q = 0; q < CT; q++) {
      const int tq = (c * CT + q < T) ? c * CT + q : T - 1;
#pragma unroll
      for (int jj = 0; jj < NU; jj++) uq[q][jj] = v.cand_u[tidx(ta, tq, jj, l, T, NU)];
    }
    }
#if ILQR_COMMIT_ROLLED
    // ONE copy of the step in the instruction stream (a rolled loop; the chunk's controls move down a register per trip so that
    // every index stays static): 1 KB instead of 7 KB of code that runs once per iteration between two long loops.  Same time as the
    // unrolled form (measured; the commit's 17 us are two rounds of HBM latency + 7 Euler steps + 52 memory instructions per wavefront).
#pragma unroll 1
    for (int q = 0; q < CT; q++) {
      const int t = c * CT + q;
      if (t > T) break;
#pragma unroll
      for (int i = 0; i < NX; i++) v.xs[tidx(tile, t, i, l, T + 1, NX)] = x[i];
      if (t < T) {
#pragma unroll
        for (int jj = 0; jj < NU; jj++) v.us[tidx(tile, t, jj, l, T, NU)] = uq[0][jj];
        if (q + 1 < CT) {
          real x1[NX];
          integrate_dynamics(model, x, uq[0], dt, x1);
#pragma unroll
          for (int i = 0; i < NX; i++) x[i] = x1[i];
        }
      }
#pragma unroll
      for (int qq = 0; qq + 1 < CT; qq++)
#pragma unroll
        for (int jj = 0; jj < NU; jj++) uq[qq][jj] = uq[qq + 1][jj];
    }
#else
#pragma unroll
    for (int q = 0; q < CT; q++) {  // (no early exit: the loop must unroll for uq to stay in registers)
      const int t = c * CT + q;
      if (t <= T) {
#pragma unroll
        for (int i = 0; i < NX; i++) v.xs[tidx(tile, t, i, l, T + 1, NX)] = x[i];
        if (t < T) {
#pragma unroll
          for (int jj = 0; jj < NU; jj++) v.us[tidx(tile, t, jj, l, T, NU)] = uq[q][jj];
          if (q + 1 < CT) {
            real x1[NX];
            integrate_dynamics(model, x, uq[q], dt, x1);
#pragma unroll
            for (int i = 0; i < NX; i++) x[i] = x1[i];
          }
        }
      }
    }
#endif
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
  __shared__ double steps[104];  // (the chain runs in double for every handle)
  __shared__ double lds_cost[NALPHA * TW];
  __shared__ int tile_running;
  __shared__ int simd_count[4];
  __shared__ int lds_commit[TW];  // the accepted alpha of each trajectory of the tile, from accept_one to the commit
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
#ifdef ILQR_HEX_SECTIONS
  long long t_sweep_last = 0;
#endif
  const bool timing = (phase_ticks != nullptr) & (threadIdx.x == 0);
  const long long c_begin = timing ? clock64() : 0, w_begin = timing ? wall_clock64() : 0;
  if (commit_pending) {  // accepts of an earlier launch that nobody has copied yet
    commit_tile_chunks<M>(v, model, commit_idx + tile * TW, tile);
    phase_barrier();
  }
  int it = 0;
  for (; it < n_iters; it++) {
    if (timing) t0 = wall_clock64();
    sweep_backward_hex<M, MFD, P>(v, model, fdm, sp, 1, force, tile, pairs, steps, role);
    phase_barrier();  // the tile's gains, lambda, status are in memory for its rollout wavefronts
    if (timing) {
      const long long t1 = wall_clock64();
#ifdef ILQR_HEX_SECTIONS
      t_sweep_last = t1 - t0;
#endif
      t_sweep += t1 - t0;
      t0 = t1;
    }
    rollout_tile<M, true, true, kDeepPrefetch<M>, true, true, true, 1, kHexCandT>(v, model, alphas, NALPHA, v.cost_c, 1, sp, commit_idx, tile, lds_cost, /*count_running=*/it == n_iters - 1, rwave,
                                               pairs[role & 3].ring);
#ifdef ILQR_HEX_SECTIONS  // experiment builds (scripts/hex_sections.sh): the "backward" clock runs up to mark ILQR_HEX_SECTIONS, the "rollout" clock from there
#define ILQR_HEX_MARK(n)                                   \
    if (ILQR_HEX_SECTIONS == n && timing) {                \
      const long long t1 = wall_clock64();                 \
      t_sweep = t_sweep - t_sweep_last + (t1 - t0);        \
      t0 = t1;                                             \
    }
#else
#define ILQR_HEX_MARK(n)
#endif
    ILQR_HEX_MARK(1)  // rollouts + accept
    // (accept_one ran in threads 0 .. TW-1 at the end of rollout_tile: each hands its trajectory's accepted alpha on through LDS;
    //  the candidates themselves were stored by this block's rollout wavefronts and are waited for)
    if (threadIdx.x < TW) lds_commit[threadIdx.x] = commit_idx[tile * TW + threadIdx.x];
    if (threadIdx.x == 0) tile_running = 0;
    phase_barrier();  // candidates, costs, status are in memory
    ILQR_HEX_MARK(2)
    commit_tile_chunks<M, kHexCandT>(v, model, lds_commit, tile);
    ILQR_HEX_MARK(3)
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
