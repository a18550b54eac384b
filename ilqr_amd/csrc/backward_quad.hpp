// backward_quad.hpp -- the backward pass with one QUAD of lanes per trajectory (nx = 4): DPP exchanges, the scalar box-QP
// fast path and its quad-parallel Armijo search, records from HBM (k_backward_q) or from the LDS ring (solve_tile.hpp).
#pragma once
#include "derivatives.hpp"

namespace ilqr {

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
  const bool sane = p_and(p_and(q.Q != real(0), p_and(thr > 0.f, thr < 1.f)), p_and(kg >= 1, kg <= 96));
  const int k1 = p_and(sane, kg > 2) ? kg - 1 : 1;
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
  const bool wok = p_and(p_and(w != 0u, p_or(wwin > 1, k1 == 1)), p_or(sane, k1 == 1));
  const bool ok = p_or(unit, wok);
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
  // (the unit-step trial is lane 0's candidate: bit 0 of s4)
  // The same once the search direction is rounding noise (late in a solve Quu reaches 1e12+ and x
  // sits on the optimum to an ulp: search ~ 1e-19): steps 1 and 0.6 still move x by an ulp, from
  // 0.36 on the trial IS x.  With the window at k = 1, 2, 3 every k <= 3 has been tested exactly;
  // if none passes and the k = 3 trial equals x, no later k can pass either.
  const unsigned int s4 = (unsigned int)(__ballot(my_x1 == q.x) >> (lane & ~3)) & 0xFu;
  const bool stuck = p_and((s4 & 1u) != 0u, !q.early);
  const bool dead = p_and(p_and(k1 == 1, m4 == 0u), p_and((s4 & 8u) != 0u, !q.early));
  q.ls_failed = p_or(q.ls_failed, p_or(stuck, dead));
  return p_or(p_or(ok, q.early), p_or(stuck, dead));
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
// Arithmetic.  `real` (M::real) is what is STORED per knot: records, controls, gains.  The recursion itself -- Vx, Vxx, the Q
// blocks, the box-QP, the value update -- runs in `creal` = double for every handle: an fp32 handle's chain is the fp64 chain
// on float records (the mixed mode of DESIGN.md 3.6: float cannot resolve Quu = cuu + fu'Vxx fu once lambda has reached 0;
// the chain is bound by instruction issue, not by bytes, so the price is the conversions).  The stored gains are the
// roundings of the double ones; the warm start of the next box-QP is the STORED k (as the reference reads k[i+1] back).
template <class M, class Gate, int RING_KB = ILQR_RING_KB, bool FIXES = true, bool ONESET = false>
__device__ __forceinline__ void backward_quad(const BatchViewT<typename M::real>& v, const M& model, const SolverParams& sp, int mode,
                                              int tile, int lane, const double* __restrict__ lds_steps, Gate& gate,
                                              const typename M::real* ring = nullptr) {
  using real = typename M::real;
  using creal = double;
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
    creal Vx[4], Vxx[16], kprev[NU];
    const creal lam_r = (creal)lambda;  // the regularisation of this pass (:367)
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
    // (gfx950 counts stores in vmcnt too: with this load still "in flight" at the loop header the compiler made every step wait for
    //  vmcnt(0) before its first use of kprev -- i.e. for the previous step's gain stores to be acknowledged by the L2, a few hundred
    //  cycles of a 1500-cycle step.  Waited for here, once per pass, no step waits for memory at all.)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    dV0 = dV1 = 0;
    diverge = 0;
    gacc = 0;

    // one Riccati step; returns false if the box-QP reports failure (ilqr_core.cpp:371)
    real* __restrict__ Kt_i = Kt + (unsigned)(((T - 1) * NU * NX + NU * s) * TW);  // this lane's K(:, s) and k of step i
    real* __restrict__ kt_i = kt + (unsigned)((T - 1) * NU * TW);
    auto step = [&](int i, const QuadStep<NU, real>& raw) -> bool {
      struct {  // the record, unpacked (register renames: the loads have landed, see QuadStep) and widened to the chain's arithmetic
        creal fx[16], fxc[4], fu[4 * NU], cu[NU], cuu[NU * NU], us[NU], cx, cxx[4], cxu[NU];
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
        creal tail[NU + NU * NU];
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
      creal W[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        creal acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += Vxx[r + 4 * q] * d.fxc[q];
        W[r] = acc;
      }
      // Qxx[:, s] = cxx[:, s] + fx' W      :361
      creal Qxxc[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        creal acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += d.fx[q + 4 * r] * W[q];
        Qxxc[r] = d.cxx[r] + acc;
      }
      // Qx[s] = cx[s] + fx[:, s]' Vx'      :359
      creal Qxs;
      {
        creal acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += d.fxc[q] * Vx[q];
        Qxs = d.cx + acc;
      }
      // Qux[:, s] = cxu[s, :]' + fu' W     :362/:366
      creal Quxc[NU];
#pragma unroll
      for (int a = 0; a < NU; a++) {
        creal acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += d.fu[q + 4 * a] * W[q];
        Quxc[a] = d.cxu[a] + acc;
      }
      // replicated: Qu, wv = Vxx' fu, Quu, QuuF     :360, :363, :367
      creal Qu[NU], Quu[NU * NU], QuuF[NU * NU];
#pragma unroll
      for (int a = 0; a < NU; a++) {
        creal acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += d.fu[q + 4 * a] * Vx[q];
        Qu[a] = d.cu[a] + acc;
      }
#pragma unroll
      for (int c = 0; c < NU; c++) {
        creal wv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += Vxx[r + 4 * q] * d.fu[q + 4 * c];
          wv[r] = acc;
        }
#pragma unroll
        for (int a = 0; a < NU; a++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += d.fu[q + 4 * a] * wv[q];
          Quu[a + NU * c] = d.cuu[a + NU * c] + acc;
          QuuF[a + NU * c] = (d.cuu[a + NU * c] + ((a == c) ? lam_r : creal(0))) + acc;
        }
      }
      // opt-in (sp.fixes & 4, see k_backward_t): Quu_reg = Quu + lambda fu'fu, Qux_reg[:, s] = Qux[:, s] + lambda fu'fx[:, s]
      creal Quxr[NU];
#pragma unroll
      for (int a = 0; a < NU; a++) Quxr[a] = Quxc[a];
      const bool reg_vxx = FIXES && (sp.fixes & 4) != 0;
      if (reg_vxx) {
#pragma unroll
        for (int a = 0; a < NU; a++) {
#pragma unroll
          for (int c = 0; c < NU; c++) {
            creal acc = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) acc += d.fu[q + 4 * a] * d.fu[q + 4 * c];
            QuuF[a + NU * c] = Quu[a + NU * c] + lam_r * acc;
          }
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += d.fu[q + 4 * a] * d.fxc[q];
          Quxr[a] = Quxc[a] + lam_r * acc;
        }
      }
      // :369  box-QP (replicated in the quad)
      creal lo[NU], hi[NU];
#pragma unroll
      for (int a = 0; a < NU; a++) {
        lo[a] = model.u_min[a] - d.us[a];
        hi[a] = model.u_max[a] - d.us[a];
      }
      // :371  a failed QP ends the pass.  No early return: the rest of the step is computed
      // anyway (its results are discarded) so that the vmcnt wait below sits on every path.
      struct { creal x[NU]; } qp;
      creal Kc[NU];
      bool ok;
      creal k_scale = 0;    // (NU == 1: what K is scaled from, see the exchange below)
      if constexpr (NU == 1) {
        int free0;
        creal minv;
        QP1StateT<creal> q1;
        qp1_begin<false>(QuuF[0], Qu[0], kprev[0], lo[0], hi[0], q1, FIXES && (sp.fixes & 2) != 0);
        if (__builtin_expect(!qp1_search_quad(q1, s, lane, lds_steps), 0)) {  // fallback: rare, out of line
          q1.step = 1;
          q1.x1 = qp1_trial(q1, creal(1));
          q1.v1 = qp1_value(q1, q1.x1);
          qp1_backtrack_seq(q1);
        }
        bool goes_on;
        ok = qp1_finish_ok(q1, qp.x[0], free0, minv, goes_on);
        if (goes_on)  // the QP goes on (rare early in a solve, a quarter of the steps of some tiles later)
          ok = qp1_continue(
                   q1,
                   [&](QP1StateT<creal>& qs) __attribute__((always_inline)) {
                     if (__builtin_expect(!qp1_search_quad(qs, s, lane, lds_steps), 0)) {
                       qp1_line_search_seq(qs);
                     }
                   },
                   qp.x[0], free0) >= 1;
        // :373-385  K = -(R^-1 R^-T) Qux on a free control, 0 on a clamped one: one scale for the row (0 x Qux = +-0)
        k_scale = free0 ? -minv : creal(0);
        Kc[0] = k_scale * Quxr[0];
      } else if constexpr (NU == 2) {
        // m = 2: the scalarised solver (boxqp.hpp: box_qp2); K[:, s] = -(R^-1 R^-T) Qux[free, s] scattered to the free rows (:373-385)
        BoxQP2Result<creal> r;
        box_qp2(QuuF, Qu, kprev, lo, hi, r, FIXES && (sp.fixes & 2) != 0);
        ok = r.result >= 1;
        qp.x[0] = r.x[0];
        qp.x[1] = r.x[1];
        const bool both = r.free0 & r.free1;
        const creal q0 = r.free0 ? Quxr[0] : Quxr[1];  // rows_w_ind(Qux_reg, v_free)(:, s), by rank
        const creal kA = (r.nfR == 2) ? (-r.m00 * q0 + -r.m01 * Quxr[1]) : -r.m00 * q0;  // rank 0 (the second term only if both are free)
        const creal kB = -r.m01 * Quxr[0] + -r.m11 * Quxr[1];                              // rank 1
        Kc[0] = r.free0 ? kA : creal(0);
        Kc[1] = r.free1 ? (both ? kB : kA) : creal(0);
      } else {
        BoxQPResult<NU, creal> r;
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
          creal Minv[NU * NU], qf[NU];
          rinv_rinvT<NU>(r.nfR, r.R, Minv);
          const int nuse = (nf < r.nfR) ? nf : r.nfR;
#pragma unroll
          for (int a = 0; a < NU; a++) {
            creal val = 0;
#pragma unroll
            for (int j = 0; j < NU; j++)
              if (r.v_free[j] && rank[j] == a) val = Quxr[j];
            qf[a] = val;
          }
#pragma unroll
          for (int j = 0; j < NU; j++)
            if (r.v_free[j] && rank[j] < nuse) {
              creal acc = 0;
#pragma unroll
              for (int a = 0; a < NU; a++)
                if (a < nuse) {
                  creal mrow = 0;
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
        creal d0 = 0;
#pragma unroll
        for (int a = 0; a < NU; a++) d0 += qp.x[a] * Qu[a];
        if (ok) dV0 += (double)d0;
        creal d1 = 0;
#pragma unroll
        for (int c = 0; c < NU; c++) {
          creal r = 0;
#pragma unroll
          for (int a = 0; a < NU; a++) r += (creal(0.5) * qp.x[a]) * Quu[a + NU * c];
          d1 += r * qp.x[c];
        }
        if (ok) dV1 += (double)d1;
      }
      // T1s[c] = (K' Quu)[s, c]
      creal T1s[NU];
#pragma unroll
      for (int c = 0; c < NU; c++) {
        creal acc = 0;
#pragma unroll
        for (int q = 0; q < NU; q++) acc += Kc[q] * Quu[q + NU * c];
        T1s[c] = acc;
      }
      // :391  Vx[s]
      creal Vxs;
      {
        creal t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
        for (int c = 0; c < NU; c++) {
          t1 += T1s[c] * qp.x[c];
          t2 += Kc[c] * Qu[c];
          t3 += Quxc[c] * qp.x[c];
        }
        Vxs = ((Qxs + t1) + t2) + t3;
      }
      // exchange K, Qux, K'Quu columns inside the quad
      creal Kall[NU][4], Qall[NU][4], T1all[NU][4];
#pragma unroll
      for (int a = 0; a < NU; a++) quad_gather(Quxc[a], Qall[a]);  // (does not wait for the box-QP)
      if (NU == 1 && !reg_vxx) {
        // K[0, r] = -minv Qux[0, r] in lane r; the same product of the same operands here: no second exchange
#pragma unroll
        for (int r = 0; r < 4; r++) Kall[0][r] = k_scale * Qall[0][r];
      } else {
#pragma unroll
        for (int a = 0; a < NU; a++) quad_gather(Kc[a], Kall[a]);
      }
#pragma unroll
      for (int c = 0; c < NU; c++)  // (K'Quu)[r, c] for every r, from the gathered K (no third exchange)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < NU; q++) acc += Kall[q][r] * Quu[q + NU * c];
          T1all[c][r] = acc;
        }
      // :392  Vn[r, s] = Qxx[r,s] + (K'Quu)[r,:] K[:,s] + K[:,r]' Qux[:,s] + Qux[:,r]' K[:,s]
      creal Vn[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        creal t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
        for (int q = 0; q < NU; q++) {
          t1 += T1all[q][r] * Kc[q];
          t2 += Kall[q][r] * Quxc[q];
          t3 += Qall[q][r] * Kc[q];
        }
        Vn[r] = ((Qxxc[r] + t1) + t2) + t3;
      }
      // all-gather, then :393 symmetrise (every lane keeps the full matrix)
      creal Vf[16];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        creal col[4];
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
          const creal sym = creal(0.5) * (Vf[r + 4 * c] + Vf[c + 4 * r]);
          Vxx[r + 4 * c] = sym;
          Vxx[c + 4 * r] = sym;
        }
      }
      quad_gather(Vxs, Vx);
      // :405-412 term of the gradient norm for this step (summed here in descending t)
      {
        creal mx = 0;
#pragma unroll
        for (int a = 0; a < NU; a++) {
          // (the weight 1 / (|u| + 1) in the STORED arithmetic on every route: the ring's producers computed it there)
          const creal val = abs_of(qp.x[a]) * ((RP && NU == 1) ? (creal)raw.usw : (creal)recip((real)abs_of(d.us[a]) + real(1)));
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
          kprev[a] = (creal)(real)qp.x[a];  // the stored gain, as the reference reads k[i + 1] back (:369)
          Kt_i[a * TW] = (real)Kc[a];
        }
        if (s == 0) {
#pragma unroll
          for (int a = 0; a < NU; a++) kt_i[a * TW] = (real)qp.x[a];
        }
      }
      // (running per-lane pointers, not base + i * stride: the scalar index cost two SGPRs whose zero high word the
      //  register allocator kept reloading from its spill lanes inside this block)
      Kt_i -= NU * NX * TW;
      kt_i -= NU * TW;
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

// (the chain runs in double for every handle: the double table)
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
  __shared__ double lds_steps[104];  // backtracking step sizes (per-lane indexed -> LDS, not constant cache)
  load_step_table(lds_steps);
  NoGate gate;
  if (sp.fixes)
    backward_quad<M, NoGate, ILQR_RING_KB, true>(v, model, sp, mode, (int)blockIdx.x, (int)threadIdx.x, lds_steps, gate);
  else
    backward_quad<M, NoGate, ILQR_RING_KB, false>(v, model, sp, mode, (int)blockIdx.x, (int)threadIdx.x, lds_steps, gate);
}

}  // namespace ilqr
