// backward_hex.hpp -- the nx = 4, nu = 1 backward pass with SIXTEEN lanes per trajectory (experiment, opt-in).
//
// backward_quad (kernels.hpp) gives a trajectory four lanes, each owning a column of the 4 x 4 quantities and a full
// copy of Vxx: 464 instructions per Riccati step, issued at ~0.9 of what one wavefront can issue, so
// the only lever left is the instruction count.  Here lane (r, c) of a 16-lane DPP row owns ELEMENT [r, c]:
//   * row_ror:4k rotates whole quads: lane (r, c) sees Vxx[r, (c - k) & 3] -- a row of Vxx in three moves;
//   * quad_perm broadcasts inside a quad: a column of W = Vxx fx in four;
//   * row_newbcast:n hands one lane's value to the whole row: the scalars that feed the box-QP (Quu, Qu, wv, Vx) are
//     computed ONCE and broadcast, so the sixteen redundant evaluations of the QP see bit-identical inputs;
//   * Vxx is exactly symmetric after (V + V')/2, so fu'Vxx is a sum over a quad's own lanes (two DPP add stages);
//   * the transpose for the symmetrisation is one ds_bpermute pair.
// Products are 4 FMAs instead of 16-32, there is no 32-move all-gather of Vxx: ~330 instructions per step.
// The box-QP is backward_quad's (qp1_*, the quad Armijo search), evaluated by every quad of the row.
// Sums run in rotated / tree order, so results differ from backward_quad's in the last bits (not bit-identical;
// tests/test_gpu_hex_backward.py compares the two to rounding and this one against the CPU restatement with the
// usual per-knot criteria).
//
// Stage call only (records in HBM, ILQR_FLAG_BACKWARD_LANE_GROUP): grid = 4 x tiles, one wavefront = 4 trajectories.
#pragma once
#include "kernels.hpp"

namespace ilqr {

template <int CTRL>
__device__ __forceinline__ double dpp_ctl(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_ctl(float x) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}

template <class real>
struct HexStep {  // what lane (r, c) needs of one record, as loaded (pairs stay pairs, see QuadStep)
  typedef real pair_t __attribute__((ext_vector_type(2)));
  pair_t fxc[2];  // fx[0..3, c]
  pair_t fxr[2];  // fx[0..3, r]
  pair_t fu[2];   // fu[0..3]
  pair_t tail;    // cu, cuu
  real fxs[4];    // fx[(c - k) & 3, c], k = 0..3: the partners of the rotated Vxx
  real fur;       // fu[r]
  real cxc, cxuc, cxur, cxx, us;
};

// ---- the ring protocol for FOUR backward wavefronts per tile (see RingGate in kernels.hpp for the one-consumer form) ----
// The producers keep treating the tile as one stream of knots, pass after pass.  What changes on the consumer side:
//   * consumer w writes its position to consumer_at4[w]; the producers' lead is measured against the minimum;
//   * passes are tile-wide, as they are with one backward wavefront: before every pass (the first included) the four
//     wavefronts VOTE -- each adds to needs[pass & 1] / ORs its trajectories into lanes16[pass & 1] if any of its
//     trajectories takes part, then arrives; when all four have arrived, wavefront 0 publishes the pass to the producers
//     (pass_lanes, passes_started) or the end of the tile (passes_started = -1), and everybody goes on or leaves together.
//     A wavefront without a trajectory in the pass skips it but votes again.  Counters are monotonic and double-buffered
//     by pass parity: a slot is written again two votes later, i.e. after everybody has passed the vote in between.
constexpr int kHexRingPad = 2;  // `real`s per pair row of a ring slot (RingSlot): rows of one trajectory on different banks
template <class real, int NX, int NU, int kProd, int RING_KB>
struct SweepSharedH : SweepShared<real, NX, NU, kProd, RING_KB, kHexRingPad> {
  alignas(16) int consumer_at4[4];
  int arrived;
  int needs[2];
  unsigned lanes16[2];
};

template <class SH, int kProd>
struct RingGateH {
  static constexpr bool kRing = true;
  static constexpr int kKnotsPerRound = 4 * kProd;
  SH& sh;
  const int T, nrounds, N, w;
  int pass = -1, have = 0, seen[2] = {0, 0};
  __device__ __forceinline__ RingGateH(SH& s, int T_, int w_)
      : sh(s), T(T_), nrounds((T_ + 1 + kKnotsPerRound - 1) / kKnotsPerRound), N(nrounds * kKnotsPerRound), w(w_) {}
  // vote for the next pass; need: this wavefront has trajectories in it (mask16: which of the tile's 16).  Returns
  // whether the tile runs the pass.  Called by all lanes of the wavefront (wave-uniform arguments).
  __device__ __forceinline__ bool begin_pass(bool need, unsigned mask16) {
    const int P = pass + 1, slot = P & 1;
    if ((threadIdx.x & 63) == 0) {
      if (need) {
        __hip_atomic_fetch_or(&sh.lanes16[slot], mask16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sh.needs[slot], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __hip_atomic_fetch_add(&sh.arrived, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    while (__hip_atomic_load(&sh.arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (P + 1)) __builtin_amdgcn_s_sleep(2);
    const int n = __hip_atomic_load(&sh.needs[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const bool any = n != seen[slot];
    seen[slot] = n;
    if (w == 0 && (threadIdx.x & 63) == 0) {
      if (any) {
        const unsigned m = __hip_atomic_exchange(&sh.lanes16[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&sh.pass_lanes, (unsigned long long)m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&sh.passes_started, P + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        __hip_atomic_store(&sh.passes_started, -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    pass = P;
    have = P * N;
    if (!any || !need)  // not in this pass: never hold the producers back
      __hip_atomic_store(&sh.consumer_at4[w], any ? (P + 1) * N : 0x3fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return any;
  }
  __device__ __forceinline__ void end_pass() {  // through with (or out of) the pass: the next thing read is the next pass's first knot
    __hip_atomic_store(&sh.consumer_at4[w], (pass + 1) * N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ int slot(int t) const { return (pass * N + (T - t)) % SH::RS::SLOTS; }
  __device__ __forceinline__ void wait(int t) {
    const int j = T - t, G = pass * N + j;
    if (G < have) return;
    const int round = pass * nrounds + j / kKnotsPerRound, pw = (j % kKnotsPerRound) / 4;
    __hip_atomic_store(&sh.consumer_at4[w], G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(&sh.rounds_done[pw], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= round) __builtin_amdgcn_s_sleep(2);
    have = pass * N + (j / kKnotsPerRound) * kKnotsPerRound + (pw + 1) * 4;
  }
};
struct NoGateH {  // records in HBM, one wavefront on its own: a pass happens iff this wavefront needs it
  static constexpr bool kRing = false;
  __device__ __forceinline__ bool begin_pass(bool need, unsigned) { return need; }
  __device__ __forceinline__ void end_pass() {}
  __device__ __forceinline__ void wait(int) {}
  __device__ __forceinline__ int slot(int) const { return 0; }
};

// The backward pass of the four trajectories 4 sub .. 4 sub + 3 of a tile, run by ONE wavefront (lane = 16 j + 4 c + r).
// Records from HBM (NoGateH) or from the producers' LDS ring (RingGateH).  Every lane of the wavefront must call this
// (the votes of RingGateH are wave-level); trajectories beyond the batch / not running simply take no part.
template <class M, class Gate, int RING_KB = ILQR_RING_KB>
__device__ __forceinline__ void backward_hex(const BatchViewT<typename M::real>& v, const M& model, const SolverParams& sp, int mode, int tile,
                                             int sub, int lane, const typename M::real* __restrict__ lds_steps, Gate& gate,
                                             const typename M::real* ring = nullptr) {
  using real = typename M::real;
  static_assert(M::NX == 4 && M::NU == 1, "16 lanes per trajectory: 4 x 4 state matrices, scalar control");
  using R = Rec<4, 1>;
  constexpr bool RP = Gate::kRing;
  using RS = RingSlot<4, 1, real, RING_KB, RP ? kHexRingPad : 0>;
  typedef real real2_t __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(3))) char lds_cc;
  const int j = lane >> 4, c = (lane >> 2) & 3, r = lane & 3, s = r;
  const int l = 4 * sub + j;  // trajectory inside the tile
  const int b = tile * TW + l;
  const bool exists = b < v.B;
  const int bc = exists ? b : 0;
  const bool takes_part = exists && !(mode == 1 && v.status[bc] != 0);  // row-uniform
  bool in_loop = takes_part;
  const int T = v.T;
  double lambda = v.lambda[bc], dlambda = v.dlambda[bc];
  // Addressing: a wave-uniform base (the tile's records in HBM: an SGPR pair; a ring slot: an LDS address) + 32-bit
  // BYTE offsets per lane and element -- per-lane 64-bit pointers cost 27 address instructions per step.
  const char* __restrict__ Dtile = RP ? nullptr : reinterpret_cast<const char*>(v.D + didx(tile, 0, 0, 0, T + 1, R::SIZE));
  const real* __restrict__ ust = v.us + tidx(tile, 0, 0, l, T, 1);
  real* __restrict__ kt = v.kff + tidx(tile, 0, 0, l, T, 1);
  real* __restrict__ Kt = v.Kfb + tidx(tile, 0, 0, l, T, 4);
  constexpr unsigned kStepBytes = (R::SIZE / 2) * 2 * TW * sizeof(real);
  constexpr unsigned kSlotBytes = RS::ELEMS * sizeof(real);
  auto off = [&](int e) { return (unsigned)(((e >> 1) * RS::ROW + (e & 1) + 2 * l) * sizeof(real)); };
  const unsigned o_fxc0 = off(R::FX + 4 * c), o_fxc1 = off(R::FX + 4 * c + 2);
  const unsigned o_fxr0 = off(R::FX + 4 * r), o_fxr1 = off(R::FX + 4 * r + 2);
  unsigned o_fxs[4];
#pragma unroll
  for (int k = 0; k < 4; k++) o_fxs[k] = off(R::FX + ((c - k) & 3) + 4 * c);
  const unsigned o_fur = off(R::FU + r), o_cxc = off(R::CX + c), o_cxx = off(R::CXX + r + 4 * c),
                 o_cxuc = off(R::CXU + c), o_cxur = off(R::CXU + r);
  const unsigned o_fu0 = off(R::FU), o_fu1 = off(R::FU + 2), o_tail = off(R::CU), o_us = off(RS::US);
  const int tlane = (lane & ~15) + 4 * r + c;  // the lane holding element [c, r]

  auto fill = [&](auto pair, auto one, HexStep<real>& d) __attribute__((always_inline)) {
    d.fxc[0] = pair(o_fxc0);
    d.fxc[1] = pair(o_fxc1);
    d.fxr[0] = pair(o_fxr0);
    d.fxr[1] = pair(o_fxr1);
    d.fu[0] = pair(o_fu0);
    d.fu[1] = pair(o_fu1);
    d.tail = pair(o_tail);
#pragma unroll
    for (int k = 0; k < 4; k++) d.fxs[k] = one(o_fxs[k]);
    d.fur = one(o_fur);
    d.cxc = one(o_cxc);
    d.cxuc = one(o_cxuc);
    d.cxur = one(o_cxur);
    d.cxx = one(o_cxx);
  };
  auto load = [&](int t, HexStep<real>& d) __attribute__((always_inline)) {
    gate.wait(t);
    if constexpr (RP) {
      lds_cc* q = (lds_cc*)((const char*)ring + (unsigned)gate.slot(t) * kSlotBytes);
      auto pair = [&](unsigned o) { return *(const __attribute__((address_space(3))) real2_t*)(q + o); };
      auto one = [&](unsigned o) { return *(const __attribute__((address_space(3))) real*)(q + o); };
      fill(pair, one, d);
      d.us = one(o_us);
    } else {
      const unsigned tb = (unsigned)t * kStepBytes;
      auto pair = [&](unsigned o) { return *reinterpret_cast<const real2_t*>(Dtile + (tb + o)); };
      auto one = [&](unsigned o) { return *reinterpret_cast<const real*>(Dtile + (tb + o)); };
      fill(pair, one, d);
      d.us = ust[(unsigned)(t * TW)];
    }
  };

  constexpr int kWaitAll = (7 << 4) | (15 << 8);  // s_waitcnt vmcnt(0) only
  constexpr int kWaitLds = 0xC07F;                // s_waitcnt lgkmcnt(0) only
  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0, gacc = 0;
  auto one_pass = [&]() __attribute__((always_inline)) {
    real Vxx, Vx[4], kprev;
    const real lam_r = (real)lambda;
    {
      gate.wait(T);
      if constexpr (RP) {
        lds_cc* q = (lds_cc*)((const char*)ring + (unsigned)gate.slot(T) * kSlotBytes);
        Vxx = *(const __attribute__((address_space(3))) real*)(q + o_cxx);  // :354
#pragma unroll
        for (int i = 0; i < 4; i++) Vx[i] = *(const __attribute__((address_space(3))) real*)(q + off(R::CX + i));  // :353
      } else {
        const unsigned tb = (unsigned)T * kStepBytes;
        Vxx = *reinterpret_cast<const real*>(Dtile + (tb + o_cxx));  // :354
#pragma unroll
        for (int i = 0; i < 4; i++) Vx[i] = *reinterpret_cast<const real*>(Dtile + (tb + off(R::CX + i)));  // :353
      }
    }
    kprev = kt[(size_t)(T - 1) * TW];
    dV0 = dV1 = 0;
    diverge = 0;
    gacc = 0;
    // raw: this step's record (landed); nxt: where the next step's record goes.  From HBM the prefetch is issued by the
    // caller before the step (a whole step of latency to cover); from the ring it is issued INSIDE the step, after the
    // box-QP's own LDS operations (step-table read, shuffles): LDS operations complete in issue order, so a shuffle
    // issued behind 17 record reads waits for all of them.
    auto step = [&](int i, const HexStep<real>& raw, HexStep<real>& nxt) -> bool {
      const real fxc[4] = {raw.fxc[0].x, raw.fxc[0].y, raw.fxc[1].x, raw.fxc[1].y};
      const real fxr[4] = {raw.fxr[0].x, raw.fxr[0].y, raw.fxr[1].x, raw.fxr[1].y};
      const real fu[4] = {raw.fu[0].x, raw.fu[0].y, raw.fu[1].x, raw.fu[1].y};
      const real cu = raw.tail.x, cuu = raw.tail.y;
      // W[r, c] = sum_q Vxx[r, q] fx[q, c], q = c, c-1, c-2, c-3 (the rotations' order)
      const real V1 = dpp_ctl<0x124>(Vxx), V2 = dpp_ctl<0x128>(Vxx), V3 = dpp_ctl<0x12C>(Vxx);
      real W = Vxx * raw.fxs[0];
      W += V1 * raw.fxs[1];
      W += V2 * raw.fxs[2];
      W += V3 * raw.fxs[3];
      // Qxx[r, c] = cxx[r, c] + sum_q fx[q, r] W[q, c]      :361
      real Wq[4];
      quad_gather(W, Wq);
      real Qxx;
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fxr[q] * Wq[q];
        Qxx = raw.cxx + acc;
      }
      // Qx[c] = cx[c] + fx[:, c]'Vx      :359
      real Qxc;
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fxc[q] * Vx[q];
        Qxc = raw.cxc + acc;
      }
      // wv = Vxx fu: Vxx is exactly symmetric, so wv[c] = sum_r Vxx[r, c] fu[r] over the quad's own lanes
      real wvc = Vxx * raw.fur;
      wvc += dpp_ctl<0xB1>(wvc);  // quad_perm [1,0,3,2]
      wvc += dpp_ctl<0x4E>(wvc);  // quad_perm [2,3,0,1]: every lane of quad c holds wv[c], the same bits
      const real wv[4] = {dpp_ctl<0x150>(wvc), dpp_ctl<0x154>(wvc), dpp_ctl<0x158>(wvc), dpp_ctl<0x15C>(wvc)};  // row_newbcast:0,4,8,12
      // replicated on all 16 lanes from identical operands in one order: Qu, Quu, QuuF     :360, :363, :367
      real Qu, Quu, QuuF;
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fu[q] * Vx[q];
        Qu = cu + acc;
        acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fu[q] * wv[q];
        Quu = cuu + acc;
        QuuF = (cuu + lam_r) + acc;
      }
      // Qux[c] and Qux[r]      :362
      real QuxC, QuxR;
      {
        real acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fxc[q] * wv[q];
        QuxC = raw.cxuc + acc;
        acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fxr[q] * wv[q];
        QuxR = raw.cxur + acc;
      }
      // :369 box-QP, evaluated by every quad of the row (identical inputs)
      const real lo = model.u_min[0] - raw.us, hi = model.u_max[0] - raw.us;
      real x;
      int free0;
      real minv;
      QP1StateT<real> q1;
      qp1_begin<false>(QuuF, Qu, kprev, lo, hi, q1, (sp.fixes & 2) != 0);
      if (__builtin_expect(!qp1_search_quad(q1, s, lane, lds_steps), 0)) {
        q1.step = 1;
        q1.x1 = qp1_trial(q1, real(1));
        q1.v1 = qp1_value(q1, q1.x1);
        qp1_backtrack_seq(q1);
      }
      int result = qp1_finish(q1, x, free0, minv);
      if (result == kQpGoesOn)
        result = qp1_continue(
            q1,
            [&](QP1StateT<real>& qs) __attribute__((always_inline)) {
              if (__builtin_expect(!qp1_search_quad(qs, s, lane, lds_steps), 0)) qp1_line_search_seq(qs);
            },
            x, free0);
      const bool ok = result >= 1;
      if (!ok) diverge = i;
      if constexpr (RP) {
        __builtin_amdgcn_sched_barrier(0);
        if (i >= 1) load(i - 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
      }
      const real Kc = free0 ? -minv * QuxC : real(0);  // :373-385
      const real Kr = free0 ? -minv * QuxR : real(0);
      // :388-389
      {
        const real d0 = x * Qu;
        if (ok) dV0 += (double)d0;
        const real d1 = ((real(0.5) * x) * Quu) * x;
        if (ok) dV1 += (double)d1;
      }
      // :391 Vx[c], then to every lane
      const real T1c = Kc * Quu, T1r = Kr * Quu;
      const real Vxc = ((Qxc + T1c * x) + Kc * Qu) + QuxC * x;
      Vx[0] = dpp_ctl<0x150>(Vxc);
      Vx[1] = dpp_ctl<0x154>(Vxc);
      Vx[2] = dpp_ctl<0x158>(Vxc);
      Vx[3] = dpp_ctl<0x15C>(Vxc);
      // :392 Vn[r, c] ; :393 (Vn + Vn')/2 -- the diagonal stays as it is
      const real Vn = ((Qxx + T1r * Kc) + Kr * QuxC) + QuxR * Kc;
      const real VnT = __shfl(Vn, tlane, 64);
      Vxx = (r == c) ? Vn : real(0.5) * (Vn + VnT);
      // :405-412 term of the gradient norm
      {
        const real val = abs_of(x) * recip(abs_of(raw.us) + 1);
        if (ok) gacc += (double)val;
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (RP)
        __builtin_amdgcn_s_waitcnt(kWaitLds);
      else
        __builtin_amdgcn_s_waitcnt(kWaitAll);
      __builtin_amdgcn_sched_barrier(0);
      // :396-397
      if (ok) {
        kprev = x;
        if (r == 0) Kt[(unsigned)((i * 4 + c) * TW)] = Kc;
        if ((lane & 15) == 0) kt[(unsigned)(i * TW)] = x;
      }
      return ok;
    };
    {
      HexStep<real> A, Bd;
      int i = T - 1;
      load(i, A);
      __builtin_amdgcn_s_waitcnt(kWaitAll & kWaitLds);
      while (true) {
        __builtin_amdgcn_sched_barrier(0);
        if (!RP && i >= 1) load(i - 1, Bd);
        __builtin_amdgcn_sched_barrier(0);
        if (!step(i, A, Bd)) break;
        if (--i < 0) break;
        __builtin_amdgcn_sched_barrier(0);
        if (!RP && i >= 1) load(i - 1, A);
        __builtin_amdgcn_sched_barrier(0);
        if (!step(i, Bd, A)) break;
        if (--i < 0) break;
      }
    }
  };

  // the pass loop (ilqr_core.cpp:136-150), tile-wide under RingGateH
  while (true) {
    const unsigned long long bal = __ballot(in_loop);
    unsigned mask16 = 0;
#pragma unroll
    for (int jj = 0; jj < 4; jj++) mask16 |= ((bal >> (16 * jj)) & 1ull) ? (1u << (4 * sub + jj)) : 0u;
    if (!gate.begin_pass(bal != 0ull, mask16)) break;
    if (in_loop) {
      one_pass();
      if (mode == 0) {
        done = (diverge == 0);
        in_loop = false;
      } else if (diverge != 0) {  // :142-148
        dlambda = fmax(dlambda * sp.lambda_factor, sp.lambda_factor);
        lambda = fmax(lambda * dlambda, sp.lambda_min);
        if (lambda > sp.lambda_max) in_loop = false;
      } else {
        done = true;
        in_loop = false;
      }
    }
    gate.end_pass();
  }
  if (!takes_part) return;
  double acc = gacc;
  if (!done) {  // abandoned: k[0..T) is a mix of old and new gains, re-read (as backward_quad)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    acc = 0;
    for (int t = 0; t < T; t++) acc += (double)(abs_of(kt[(size_t)t * TW]) / (abs_of(ust[(size_t)t * TW]) + 1));
  }
  const double gnorm = acc / T;
  if ((lane & 15) == 0) {
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

// stage call / records in HBM: grid = 4 x tiles, block = 64
template <class M>
__global__ __launch_bounds__(64) void k_backward_h(BatchViewT<typename M::real> v, M model, SolverParams sp, int mode) {
  using real = typename M::real;
  __shared__ real lds_steps[104];
  load_step_table(lds_steps);
  NoGateH gate;
#ifdef ILQR_PHASE_TIMING
  const long long c0 = clock64(), w0 = wall_clock64();
#endif
  backward_hex<M, NoGateH>(v, model, sp, mode, (int)blockIdx.x >> 2, (int)blockIdx.x & 3, (int)threadIdx.x, lds_steps, gate);
#ifdef ILQR_PHASE_TIMING
  if (v.dbg && blockIdx.x == 0 && threadIdx.x == 0) {  // shader cycles (s_memtime) and 100 MHz wall ticks of one pass: the clock the chip ran at
    v.dbg[920] = clock64() - c0;
    v.dbg[921] = wall_clock64() - w0;
  }
#endif
}

// Phase 1 of a tile with FOUR backward wavefronts (w = 0..3, trajectories 4 w .. 4 w + 3) and kProd producer wavefronts
// (w = 4 ..): sweep_backward_tile's protocol with the consumer side of RingGateH.  Block = 64 (4 + kProd) threads.
template <class M, int kProd, int RING_KB, class MFD, class SH>
__device__ __forceinline__ void sweep_backward_tile_h(const BatchViewT<typename M::real>& v, const M& model, const MFD& fdm, const SolverParams& sp,
                                                      int mode, int force, const int* __restrict__ commit_idx, int tile, SH& sh) {
  constexpr int kKnotsPerRound = 4 * kProd;
  constexpr int kLeadKnots = ILQR_LEAD_ROUNDS * kKnotsPerRound;
  using RS = typename SH::RS;
  static_assert(RS::SLOTS >= kLeadKnots + 4, "the ring must hold the producers' lead plus the four knots in production");
  if (threadIdx.x < kProd) sh.rounds_done[threadIdx.x] = 0;
  if (threadIdx.x >= 64 && threadIdx.x < 68) sh.consumer_at4[threadIdx.x - 64] = 0;
  if (threadIdx.x == 128) {
    sh.passes_started = 0;
    sh.arrived = 0;
    sh.needs[0] = sh.needs[1] = 0;
    sh.lanes16[0] = sh.lanes16[1] = 0;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int T = v.T;
#ifdef ILQR_PHASE_TIMING
  if (v.dbg && tile < 2 && lane == 0) v.dbg[900 + tile * 8 + wave] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  // HW_ID
#endif
  if (wave < 4) {
    __builtin_amdgcn_s_setprio(3);
    RingGateH<SH, kProd> gate(sh, T, wave);
    backward_hex<M, decltype(gate), RING_KB>(v, model, sp, mode, tile, wave, lane, sh.steps, gate, sh.ring);
    __builtin_amdgcn_s_setprio(0);
  } else {
    const int w = wave - 4;
    const int l = lane & (TW - 1), sub = lane >> 4;
    const int nrounds = (T + 1 + kKnotsPerRound - 1) / kKnotsPerRound, N = nrounds * kKnotsPerRound;
    if (w < kProd)
    for (int pass = 0;; pass++) {
      unsigned long long lanes = ~0ull;
      {  // every pass, the first included, exists only once the backward wavefronts have voted for it
        int started;
        while ((started = __hip_atomic_load(&sh.passes_started, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= 0 && started <= pass)
          __builtin_amdgcn_s_sleep(8);
        if (started < 0 && (pass > 0 || commit_idx == nullptr)) break;  // (the first pass still owes the commit of every knot)
        if (pass > 0) lanes = __hip_atomic_load(&sh.pass_lanes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      const bool mine = (lanes >> l) & 1ull;  // does trajectory l take part in this pass?
      for (int r = 0; r < nrounds; r++) {
        const int j0 = r * kKnotsPerRound + w * 4, G0 = pass * N + j0;
        while (true) {
          typedef int int4_t __attribute__((ext_vector_type(4)));
          const int4_t at = *reinterpret_cast<volatile int4_t*>(sh.consumer_at4);
          int mn = at.x < at.y ? at.x : at.y;
          const int m2 = at.z < at.w ? at.z : at.w;
          mn = mn < m2 ? mn : m2;
          if (G0 <= mn + kLeadKnots) break;
          __builtin_amdgcn_s_sleep(8);
        }
        const int started = __hip_atomic_load(&sh.passes_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const bool moved_on = (started < 0) | (started > pass + 1);
        if (moved_on && (pass > 0 || commit_idx == nullptr)) break;
        const int t = T - (j0 + sub);
        if (t >= 0 && (pass == 0 || mine))
          derivatives_of_knot<M, true, MFD, kHexRingPad>(v, model, fdm, force, pass == 0 ? commit_idx : nullptr, tile, t, l,
                                            sh.ring + ((G0 + sub) % RS::SLOTS) * RS::ELEMS + l * 2, !moved_on);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        if (lane == 0) __hip_atomic_store(&sh.rounds_done[w], pass * nrounds + r + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
}

template <class M, int kProd = kProducers, int RING_KB = ILQR_RING_KB, class MFD = M>
__global__ __launch_bounds__(64 * (4 + kProd)) void k_sweep_backward_h(BatchViewT<typename M::real> v, M model, MFD fdm, SolverParams sp, int mode,
                                                                        int force, const int* __restrict__ commit_idx) {
  using real = typename M::real;
  __shared__ SweepSharedH<real, M::NX, M::NU, kProd, RING_KB> sh;
  load_step_table(sh.steps);  // (barrier)
  if (blockIdx.x == 0 && threadIdx.x == 64) *v.n_running = 0;  // k_accept of this iteration recounts
  sweep_backward_tile_h<M, kProd, RING_KB, MFD>(v, model, fdm, sp, mode, force, commit_idx, (int)blockIdx.x, sh);
}

}  // namespace ilqr
