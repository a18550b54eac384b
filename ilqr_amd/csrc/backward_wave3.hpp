// backward_wave3.hpp -- the generic backward pass (n <= 32, m <= 16, one wavefront per trajectory) with the fp64 pipe's
// cycles counted: k_backward_w2's register layout, minus the work that layout does not need.
//
// What bounds k_backward_w2 (profiles/r05a_w2_sections.txt, scripts/ubench/coissue.hip): on gfx950 a v_mfma_f64_16x16x4_f64
// occupies the SIMD's fp64 datapath for 64.8 cycles -- exactly the 16 v_fma_f64 issue slots its 1024 multiply-adds are worth --
// and fp64 VALU instructions of the OTHER wavefront on that SIMD do not issue meanwhile (one MFMA + one FMA wavefront per SIMD:
// the FMA loop waits for the MFMA loop to end).  A step's cost is therefore the SUM of its MFMAs x 64.8 and its VALU
// instructions x ~4.5 cycles, whatever overlaps in time: 216 MFMAs + 3650 VALU + 690 SALU instructions per step in
// k_backward_w2<2>.  This kernel removes pipe cycles, not latency:
//   * matrix-VECTOR products (Qx, Qu, the Vx update: 48 MFMAs whose B operand was one vector in 16 identical columns) are
//     per-lane FMA chains over the natural registers + one cross-row-group reduction through LDS: ~80 VALU instructions;
//   * the box-QP's factorisation (Cholesky row by row over v_readlane broadcasts, triangular inverse, R^-1 R^-T: ~1450 VALU
//     instructions on 16 of 64 lanes, 28 % of a step) is replaced, when the free set is the previous step's, by a
//     NEWTON-SCHULZ REFINEMENT OF THE PREVIOUS STEP'S INVERSE ON THE MATRIX CORES: X <- X + X (I - M X), two 16x16x16
//     products = 8 MFMAs per iteration, everything in natural registers (M and X are symmetric: a natural register set is its
//     own A operand), two or three iterations from a neighbour's inverse.  Quu changes by a few per cent from one knot to the
//     next, the residual I - M X measures exactly how much, and anything that does not contract (a changed free set, a
//     non-positive-definite block, an ill-conditioned one) takes the literal path (w_box_qp: Eigen's unblocked LLT, partial
//     factors, stale factors, boxqp.cpp:80-119 as written), after which the inverse is re-seeded.  Convergence from a positive
//     definite neighbour with ||I - M X|| < 1/2 implies M positive definite, so the cases where the reference's LLT stops
//     early are never taken here.  The result differs from R^-1 R^-T by rounding (both are M^-1 to ~1e-16 cond(M));
//   * free / clamped sets as MASKS (clamped rows and columns of M replaced by the identity, of the inverse by zeros): no
//     compaction, no scatter -- masked-out terms add exact zeros to the same ascending sums;
//   * FULL (n = 16 NT, m = 16: BASELINE configs[4]): no bounds predicates on loads, sums or stores;
//   * LQF (LQ model with exact derivatives): the record's constant blocks come from const_rec as before, and cx = cxx x_t,
//     cu = cuu u_t are formed here from the knot -- per-lane FMAs on the cxx / cuu registers the step loads anyway, folded
//     into the same reduction as fx'Vx -- so no record array exists at all for that mode (44 GB at configs[4]) and no sweep
//     kernel runs; knot T's cxx (= sym(Qf)) is a second constant record.
// Everything else (operand maps, transposed products, the symmetrisation through LDS, the lambda-retry loop, the gradient
// norm) is k_backward_w2's; gains agree with it to rounding, not to the bit (tests/test_gpu_generic_backward.py).
#pragma once
#include "backward_wave2.hpp"

namespace ilqr {

constexpr int kQpBail = -100;        // the fast box-QP hands the QP to the literal path
constexpr double kNsStart = 0.03;    // max |I - M X| entry a warm start may have: ||I - M X||_2 <= 16 x that < 1/2
constexpr double kNsDone = 1e-9;     // ... below which one more update leaves a residual of ~(16 x 1e-9)^2
constexpr double kNsExact = 4e-15;   // ... and below which the update would change nothing (|I - M X| at rounding level)
constexpr int kNsMaxIter = 6;

#ifdef ILQR_W2_TIMING
__device__ long long g_w3_counts[4];  // QPs on the fast path, QPs handed to the literal path, Newton-Schulz iterations, (free)
#define ILQR_W3COUNT(k, n_) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd((unsigned long long*)&g_w3_counts[k], (unsigned long long)(n_)); }
#else
#define ILQR_W3COUNT(k, n_)
#endif

// max over the 64 lanes (result in all lanes); NaN-propagating enough for a guard: a NaN anywhere makes `rho < bound` false
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  auto row = [&](int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
  };
  return fmax(fmax(row(0), row(16)), fmax(row(32), row(48)));
}

// The warm start of the inverse and the state that says what it is the inverse of.
struct NsState {
  unsigned mask = 0;   // free set (bit i = control i free) the stored X belongs to
  bool valid = false;
};

// Newton-Schulz refinement of the stored inverse for the free set `fmask` of Q (m x m in LDS, ld LDM).  M~ = Q on the free
// rows / columns, the identity elsewhere.  On success: Xm = the masked inverse (zeros on clamped rows / columns) in natural
// registers (row 4 r + g, column p), also written to L.Minv(); the unmasked X back to Xw.  False: did not contract.
template <class LDS>
__device__ __forceinline__ bool ns_refine(LDS& L, const double* Q, double* Xw, unsigned fmask, int lane, double (&Xm)[4]) {
  const int g = lane >> 4, p = lane & 15;
  const bool fp = (fmask >> p) & 1u;
  double negM[4], X[4], id[4];
  bool fa[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int a = 4 * r + g;
    fa[r] = (fmask >> a) & 1u;
    id[r] = (a == p) ? 1.0 : 0.0;
    const double q = Q[a + LDM * p];
    negM[r] = (fa[r] && fp) ? -q : -id[r];
    X[r] = Xw[64 * r + lane];
  }
  bool done = false;
  int its = 0;
#pragma unroll 1
  for (int it = 0; it < kNsMaxIter; it++) {
    double4_t R = {id[0], id[1], id[2], id[3]};
#pragma unroll
    for (int ks = 0; ks < 4; ks++) R = __builtin_amdgcn_mfma_f64_16x16x4f64(negM[ks], X[ks], R, 0, 0, 0);  // I - M X
    const double rho = wave_max(fmax(fmax(fabs(R[0]), fabs(R[1])), fmax(fabs(R[2]), fabs(R[3]))));
    if (!(rho < kNsStart)) break;  // not a contraction (or NaN): the literal factorisation
    if (rho < kNsExact) {          // X is M's inverse to rounding already (a recursion that has reached its fixed point)
      done = true;
      break;
    }
    double4_t Xn = {X[0], X[1], X[2], X[3]};
#pragma unroll
    for (int ks = 0; ks < 4; ks++) Xn = __builtin_amdgcn_mfma_f64_16x16x4f64(X[ks], R[ks], Xn, 0, 0, 0);  // X + X R
#pragma unroll
    for (int r = 0; r < 4; r++) X[r] = Xn[r];
    its++;
    if (rho < kNsDone) {
      done = true;
      break;
    }
  }
  ILQR_W3COUNT(2, its)
  if (!done) return false;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    Xw[64 * r + lane] = X[r];
    Xm[r] = (fa[r] && fp) ? X[r] : 0.0;
    L.Minv()[(4 * r + g) + LDM * p] = Xm[r];
  }
  return true;
}

// src/boxqp.cpp:26-139 for one trajectory per wavefront, free / clamped sets as masks, the factorisation replaced by
// ns_refine.  Inputs in LDS as w_box_qp's (QuuF, Qu, kprev, lo, hi).  Returns the reference's result code -- then L.x holds the
// solution, free_out the free set at the exit, Xm the masked inverse of the last factor (all zero if nothing is free) -- or
// kQpBail: the caller runs w_box_qp on the same inputs.  The vector arithmetic is w_box_qp's expression for expression (same
// LDS operands, same sums); what it skips are recomputations of values it already holds (the gradient inside the slope, the
// old value after the first iteration, Q (x .* clamped) when nothing is clamped).
template <class LDS>
__device__ int w3_box_qp_fast(int m, LDS& L, int lane, double* Xw, NsState& ns, double (&Xm)[4], unsigned& free_out) {
  const double* Q = L.QuuF();
  const double* c = L.Qu;
  const bool mine = lane < m;
  double x = 0, lo = 0, hi = 0, cc = 0;
  if (mine) {
    lo = L.lo[lane];
    hi = L.hi[lane];
    cc = c[lane];
    const double k0 = L.kprev[lane];
    const double a = (k0 < lo) ? lo : k0;  // :35
    x = (hi < a) ? hi : a;
    L.x[lane] = x;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) Xm[r] = 0.0;
  lds_sync();
  double val;  // :36 x'Qx + x.c (no 1/2)
  {
    double part = 0, lin = 0;
    if (mine) {
      const double r = dot_padded([&](int i) { return L.x[i]; }, [&](int i) { return Q[i + LDM * lane]; });
      part = r * x;
      lin = x * cc;
    }
    val = wave_sum_row0(part) + wave_sum_row0(lin);
  }
  double oldvalue = 0;
  int result = 0;
  unsigned clamped_mask = 0, fmask_factor = 0;
  for (int iter = 0; iter <= kQpMaxIter; iter++) {
    if (iter > 0 && (oldvalue - val) < kMinRelImprove * fabs(oldvalue)) {  // :54-57
      result = 4;
      break;
    }
    // :58 grad = Qx + c ; :62-71 clamped set
    double gr = 0;
    bool isc = false;
    if (mine) {
      const double s = dot_padded([&](int j) { return Q[lane + LDM * j]; }, [&](int j) { return L.x[j]; });
      gr = s + cc;
      isc = (fabs(x - lo) < kClampTol && gr > 0) || (fabs(x - hi) < kClampTol && gr < 0);
    }
    oldvalue = val;
    const unsigned free_mask = (unsigned)__ballot(mine && !isc);
    const unsigned new_clamped = (unsigned)__ballot(mine && isc);
    const bool count_changed = __popc(clamped_mask) != __popc(new_clamped);  // sum(old_clamped - clamped) != 0, :80
    clamped_mask = new_clamped;
    if (free_mask == 0) {  // :74-77
      result = 6;
      break;
    }
    if (iter == 0 || count_changed) {  // :80
      if (!(ns.valid && ns.mask == free_mask)) return kQpBail;
      if (!ns_refine(L, Q, Xw, free_mask, lane, Xm)) {
        ns.valid = false;
        return kQpBail;
      }
      fmask_factor = free_mask;
      lds_sync();
    } else if (fmask_factor != free_mask) {
      return kQpBail;  // a stale factor of another set of equal size (:80): the literal path knows how the reference misuses it
    }
    // :93-97
    {
      const double gn2 = wave_sum_row0((mine && !isc) ? gr * gr : 0.0);
      if (grad_norm_below_min(gn2)) {
        result = 5;
        break;
      }
    }
    // :100 grad_clamped = Q (x .* clamped) + c
    double gcv = cc;
    if (new_clamped != 0) {
      if (mine) L.tmp[lane] = x * (isc ? 1.0 : 0.0);
      lds_sync();
      if (mine) {
        const double s = dot_padded([&](int j) { return Q[lane + LDM * j]; }, [&](int j) { return L.tmp[j]; });
        gcv = s + cc;
      }
    }
    if (mine) L.gc[lane] = gcv;
    lds_sync();
    // :103-119 search(free) = -(R^-1 R^-T) gc(free) - x(free), 0 elsewhere: the masked inverse has zero rows / columns there
    double srch = 0;
    if (mine) {
      const double s = dot_padded([&](int l2) { return -L.Minv()[lane + LDM * l2]; }, [&](int l2) { return L.gc[l2]; });
      srch = isc ? 0.0 : s - x;
      L.search[lane] = srch;
    }
    lds_sync();
    // :121 quadclamp_line_search (src/boxqp.cpp:143-178)
    bool failed = false;
    double v = 0, xcv = 0;
    {
      const double slope = wave_sum_row0(mine ? srch * gr : 0.0);  // search . (Q x + c): the gradient above, same expression
      if (slope >= 0) {
        failed = true;
      } else {
        double step = 1;
        if (mine) {
          const double xr = x + step * srch;
          const double a = (xr < lo) ? lo : xr;
          xcv = (hi < a) ? hi : a;
          L.xc[lane] = xcv;
        }
        lds_sync();
        v = w_quad_cost(m, Q, c, L.xc, lane);
        const double old_v = (iter == 0) ? w_quad_cost(m, Q, c, L.x, lane) : val;  // (iter > 0: val IS quadCost(x), the last search's)
        while ((v - old_v) > kArmijo * (step * slope)) {
          step *= kStepDec;
          lds_sync();
          if (mine) {
            const double xr = x + step * srch;
            const double a = (xr < lo) ? lo : xr;
            xcv = (hi < a) ? hi : a;
            L.xc[lane] = xcv;
          }
          lds_sync();
          v = w_quad_cost(m, Q, c, L.xc, lane);
          if (step < kMinStep) {
            failed = true;
            break;
          }
        }
      }
    }
    if (failed) {  // :122-125, x not updated
      result = 2;
      break;
    }
    lds_sync();
    if (mine) {
      x = xcv;  // :133-134
      L.x[lane] = x;
    }
    val = v;
    lds_sync();
  }
  lds_sync();
  free_out = ((1u << m) - 1u) & ~clamped_mask;
  if (free_out == 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) Xm[r] = 0.0;
  }
  return result;
}

// m <= 2 (what most models outside configs[4] have: one or two controls): the box-QP as the thread-level solvers of boxqp.hpp that the
// nx = 4 kernels use (box_qp_scalar / box_qp2: registers, no LDS round trips), every lane on the same values; the outputs go where
// w_box_qp leaves its own (L.x, L.vfree, the compact R^-1 R^-T in L.Minv()), so the step continues exactly as after the literal path.
// A sixteen-lane wave-level QP for a 1 x 1 or 2 x 2 block was 10-15 K cycles of a 30 K-cycle step at n = 6, m = 2.
template <class LDS>
__device__ __forceinline__ int w3_box_qp_small(int m, LDS& L, int lane, int& nfR_out, int fixes) {
  const double* Qm = L.QuuF();
  int result, nfR;
  if (m == 1) {
    double x;
    int fr;
    double minv;
    const double Q1 = Qm[0], c1 = L.Qu[0], k0 = L.kprev[0], lo1 = L.lo[0], hi1 = L.hi[0];
    result = box_qp_scalar(Q1, c1, k0, lo1, hi1, x, fr, minv);
    if (fixes & 2) {  // opt-in: a failed factorisation ends the QP (boxqp.cpp:85-88 with info() checked).  The 1 x 1 block is factored in
      // iteration 0 unless the start is clamped there (:62-77)
      const double xs0 = (hi1 < ((k0 < lo1) ? lo1 : k0)) ? hi1 : ((k0 < lo1) ? lo1 : k0), g0 = Q1 * xs0 + c1;
      const bool clamped0 = (fabs(xs0 - lo1) < kClampTol && g0 > 0) || (fabs(xs0 - hi1) < kClampTol && g0 < 0);
      if (!clamped0 && !(Q1 > 0.0)) result = -1;
    }
    nfR = 1;  // (the factor the solver holds at its exit is 1 x 1 whenever one was formed; nothing free: K = 0 below)
    lds_sync();
    for (int e = lane; e < LDM * WM; e += 64) L.Minv()[e] = 0.0;  // (the tile is read whole by the K product: S's leftovers are not zeros)
    lds_sync();
    if (lane == 0) {
      L.x[0] = x;
      L.vfree[0] = fr;
      L.Minv()[0] = minv;
    }
  } else {
    const double Q2[4] = {Qm[0], Qm[1], Qm[LDM], Qm[1 + LDM]};
    const double c2[2] = {L.Qu[0], L.Qu[1]}, x02[2] = {L.kprev[0], L.kprev[1]}, lo2[2] = {L.lo[0], L.lo[1]}, hi2[2] = {L.hi[0], L.hi[1]};
    BoxQP2Result<double> r;
    box_qp2(Q2, c2, x02, lo2, hi2, r, (fixes & 2) != 0);
    result = r.result;
    nfR = r.nfR;
    lds_sync();
    for (int e = lane; e < LDM * WM; e += 64) L.Minv()[e] = 0.0;
    lds_sync();
    if (lane == 0) {
      L.x[0] = r.x[0];
      L.x[1] = r.x[1];
      L.vfree[0] = r.free0 ? 1 : 0;
      L.vfree[1] = r.free1 ? 1 : 0;
      L.Minv()[0] = r.m00;
      L.Minv()[1] = r.m01;
      L.Minv()[LDM] = r.m01;
      L.Minv()[1 + LDM] = r.m11;
    }
  }
  lds_sync();
  nfR_out = nfR;
  return result;
}

// n <= 16 NT, m <= 16.  Arguments as k_backward_w2; LQF: const_rec holds TWO records (the constant blocks of the knots t < T,
// then knot T's) and v.D is not touched.
// REGV (ILQR_FLAG_REGULARIZE_VXX, opt-in): lambda regularises Vxx' ([Tassa 2012] eq. 10) instead of Quu -- QuuF = Quu + lambda fu'fu and the
// gains' Qux_reg = Qux + lambda fu'fx, two more transposed products per 16-column block on the operands the step holds anyway; the value
// update keeps Quu, Qux (as backward_thread.hpp does for the tiled kernels).  Instantiated without FULL / LQF.
// Wavefronts per SIMD.  n <= 16 (NT = 1): TWO since round 6 -- at three (168 registers) the kernel kept 308 bytes of scratch per lane and its
// step went through memory: the pendulum chain's pass (n = 16, m = 4, B = 4096) 7.4 -> 4.8 ms with 254 registers and no scratch (four: the
// compiler gives up on the bound).  n > 16 (NT = 2): two, 256 registers and 170-260 bytes of scratch; one wavefront per SIMD with the
// accumulation registers as spill space was measured too (ILQR_W3_NT2_WAVES=1: profiles/README.md, round 6).
#ifndef ILQR_W3_NT1_WAVES
#define ILQR_W3_NT1_WAVES 2
#endif
#ifndef ILQR_W3_NT2_WAVES
#define ILQR_W3_NT2_WAVES 2
#endif
template <int NT, bool FULL, bool LQF, bool REGV = false>
__global__ __launch_bounds__(64, NT == 2 ? ILQR_W3_NT2_WAVES : ILQR_W3_NT1_WAVES) void k_backward_w3(BatchView v, int n, int m, const double* __restrict__ u_min,
                                                                    const double* __restrict__ u_max, SolverParams sp, int mode,
                                                                    const double* __restrict__ const_rec) {
  __shared__ Wave2Lds<NT> L;
  constexpr int N = 16 * NT;
  constexpr int LDX = Wave2Lds<NT>::LD;
  constexpr int RS = N + 16;  // stride of one row group's partial sums in the reduction scratch
  static_assert(4 * RS <= LDM * N && 3 * 4 * N <= LDM * N, "reduction scratch lives in Kbuf");
  static_assert(256 <= LDM * N, "the inverse's warm start lives in Tbuf");
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  if (LQF && b == 0 && lane == 0) *v.n_running = 0;  // (no sweep kernel on this route: k_accept of this iteration recounts)
  if (b >= v.B) return;
  if (mode == 1 && v.status[b] != 0) return;
  const int T = v.T;
  const int REC = 2 * n * n + 2 * n * m + n + m + m * m;
  const int oFX = 0, oFU = oFX + n * n, oCX = oFU + n * m, oCXX = oCX + n, oCXU = oCXX + n * n, oCU = oCXU + n * m,
            oCUU = oCU + m;
  const double* __restrict__ Db = LQF ? const_rec : v.D + (size_t)b * (T + 1) * REC;
  const double* __restrict__ xsb = v.xs + (size_t)b * (T + 1) * n;
  const double* __restrict__ usb = v.us + (size_t)b * T * m;
  double* __restrict__ kb = v.kff + (size_t)b * T * m;
  double* __restrict__ Kb = v.Kfb + (size_t)b * T * m * n;
  double lambda = v.lambda[b], dlambda = v.dlambda[b];
  const int g = lane >> 4, p = lane & 15;
  {
    double* z = reinterpret_cast<double*>(&L);
    const int nz = (int)(sizeof(Wave2Lds<NT>) / sizeof(double));
    for (int e = lane; e < nz; e += 64) z[e] = 0.0;
  }
  lds_sync();
  double* const red = L.Kbuf;  // partial sums of the matrix-vector products, [row group][column]
  double* const Xw = L.Tbuf;   // the inverse's warm start, natural layout [r][lane]

  auto mfma = [](double a, double b2, double4_t c) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, c, 0, 0, 0);
  };
  const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  // natural registers: X[ti][tj][r] = X(16 ti + 4 r + g, 16 tj + p)
  unsigned lb_nn = (unsigned)(g + n * p);
  unsigned lb_tn = (unsigned)(p + n * g);
  unsigned lb_mm = (unsigned)(g + m * p);
  auto ldm = [](const double* r, bool in, unsigned off) __attribute__((always_inline)) {
    if (FULL) return r[off];
    const double val = r[in ? off : 0u];
    return in ? val : 0.0;
  };
  auto row_in = [&](int a0) __attribute__((always_inline)) { return FULL || a0 + g < n; };
  auto col_in = [&](int tj) __attribute__((always_inline)) { return FULL || 16 * tj + p < n; };
  auto mrow_in = [&](int rr) __attribute__((always_inline)) { return FULL || 4 * rr + g < m; };
  const bool mcol_in = FULL || p < m;

  // the four-way sum over the row groups of one column of the reduction scratch, group 0 first
  auto red4 = [&](int col) __attribute__((always_inline)) { return ((red[col] + red[RS + col]) + red[2 * RS + col]) + red[3 * RS + col]; };

  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0;
  NsState ns;
#ifdef ILQR_W2_TIMING
  W2Clock clk;
  clk.start();
#endif
  while (true) {
    double Vxx[NT][NT][4];
    {  // :353-354
      const double* r = LQF ? const_rec + REC : Db + (size_t)T * REC;
#pragma unroll
      for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int tj = 0; tj < NT; tj++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int a0 = 16 * ti + 4 * rr;
            Vxx[ti][tj][rr] = ldm(r, row_in(a0) && col_in(tj), lb_nn + (unsigned)(oCXX + a0 + n * 16 * tj));
          }
      if (LQF) {  // Vx = cx[T] = cxx[T] x_T
        for (int e = lane; e < n; e += 64) L.cx[e] = xsb[(size_t)T * n + e];
        lds_sync();
#pragma unroll
        for (int tj = 0; tj < NT; tj++) {
          double part = 0;
#pragma unroll
          for (int ti = 0; ti < NT; ti++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) part = __builtin_fma(Vxx[ti][tj][rr], L.cx[16 * ti + 4 * rr + g], part);
          red[g * RS + 16 * tj + p] = part;
        }
        lds_sync();
        if (lane < N) L.Vx[lane] = (FULL || lane < n) ? red4(lane) : 0.0;
      } else {
        for (int e = lane; e < n; e += 64) L.Vx[e] = r[oCX + e];
      }
      if (lane < m) L.kprev[lane] = kb[(size_t)(T - 1) * m + lane];
    }
    dV0 = dV1 = 0;
    diverge = 0;
    ns.valid = false;
    lds_sync();
    for (int i = T - 1; i >= 0; i--) {
      ILQR_W2MARK(7)
      const double* rk = LQF ? const_rec : Db + (size_t)i * REC;        // this knot's record (cx, cu, and the matrices unless const_rec has them)
      const double* rm = (LQF || const_rec) ? const_rec : rk;           // ... its matrix blocks
      double fx[NT][NT][4], fu[NT][4];
      double kx = 0, ku = 0;  // non-LQF: cx on lanes < n, cu on lanes N .. N + m - 1
      {
        asm volatile("" : "+v"(lb_nn));
#pragma unroll
        for (int ti = 0; ti < NT; ti++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int a0 = 16 * ti + 4 * rr;
            const bool ain = row_in(a0);
#pragma unroll
            for (int tj = 0; tj < NT; tj++) fx[ti][tj][rr] = ldm(rm, ain && col_in(tj), lb_nn + (unsigned)(oFX + a0 + n * 16 * tj));
            fu[ti][rr] = ldm(rm, ain && mcol_in, lb_nn + (unsigned)(oFU + a0));
          }
        const double us_l = (lane < m) ? usb[(size_t)i * m + lane] : 0.0;
        if (LQF) {
          const double xk = (lane < n) ? xsb[(size_t)i * n + lane] : 0.0;
          if (lane < n) L.cx[lane] = xk;   // the knot, for cx = cxx x_t
          if (lane < m) L.tmp[lane] = us_l;  // ... and cu = cuu u_t
        } else {
          kx = (lane < n) ? rk[oCX + lane] : 0.0;
          ku = (lane >= N && lane - N < m) ? rk[oCU + lane - N] : 0.0;
        }
        if (lane < m) {
          L.lo[lane] = u_min[lane] - us_l;  // :369
          L.hi[lane] = u_max[lane] - us_l;
        }
      }
      lds_sync();
      // :359-360 the partial sums of fx'Vx and fu'Vx over this lane's rows (16 ti + 4 r + g); reduced over g below
      double px[NT], pu = 0;
      {
        double vxr[NT][4];
#pragma unroll
        for (int ti = 0; ti < NT; ti++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) vxr[ti][rr] = L.Vx[16 * ti + 4 * rr + g];
#pragma unroll
        for (int tj = 0; tj < NT; tj++) px[tj] = 0;
#pragma unroll
        for (int ti = 0; ti < NT; ti++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
#pragma unroll
            for (int tj = 0; tj < NT; tj++) px[tj] = __builtin_fma(fx[ti][tj][rr], vxr[ti][rr], px[tj]);
            pu = __builtin_fma(fu[ti][rr], vxr[ti][rr], pu);
          }
      }
      ILQR_W2MARK(0)
      // A1' = Vxx' fx (n x n), A2' = Vxx' fu (n x m)
      double4_t a1t[NT][NT], a2t[NT];
#pragma unroll
      for (int ti = 0; ti < NT; ti++) {
        a2t[ti] = zero4;
#pragma unroll
        for (int tj = 0; tj < NT; tj++) a1t[ti][tj] = zero4;
      }
#pragma unroll
      for (int ks = 0; ks < 4 * NT; ks++) {
#pragma unroll
        for (int ti = 0; ti < NT; ti++) {
#pragma unroll
          for (int tj = 0; tj < NT; tj++) a1t[ti][tj] = mfma(Vxx[ks >> 2][ti][ks & 3], fx[ks >> 2][tj][ks & 3], a1t[ti][tj]);
          a2t[ti] = mfma(Vxx[ks >> 2][ti][ks & 3], fu[ks >> 2][ks & 3], a2t[ti]);
        }
      }
      // :361 Qxx = cxx + A1 fx ; :362 Qux = cxu' + A2 fx ; :363/:367 Quu, QuuF = cuu (+ lambda I) + A2 fu
      ILQR_W2MARK(1)
      double Qxx[NT][NT][4], Qux[NT][4], quu_nat[4];
      double Quxr[REGV ? NT : 1][4];  // REGV: Qux + lambda fu'fx, what the gains are solved from
#pragma unroll
      for (int tj = 0; tj < NT; tj++) {  // one 16-column block of the outputs at a time (registers)
        __builtin_amdgcn_sched_barrier(0);
        double cxx[NT][4], cxu[4], cuu[4];
        {
          asm volatile("" : "+v"(lb_nn), "+v"(lb_tn), "+v"(lb_mm));
#pragma unroll
          for (int ti = 0; ti < NT; ti++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
              const int a0 = 16 * ti + 4 * rr;
              cxx[ti][rr] = ldm(rm, row_in(a0) && col_in(tj), lb_nn + (unsigned)(oCXX + a0 + n * 16 * tj));
            }
#pragma unroll
          for (int rr = 0; rr < 4; rr++)  // Qux(a, c) starts from cxu(c, a): offset c + n a   (LQ: cxu = 0)
            cxu[rr] = LQF ? 0.0 : ldm(rm, mrow_in(rr) && col_in(tj), lb_tn + (unsigned)(oCXU + 16 * tj + n * 4 * rr));
          if (tj == 0) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) cuu[rr] = ldm(rm, mrow_in(rr) && mcol_in, lb_mm + (unsigned)(oCUU + 4 * rr));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (LQF) {  // cx = cxx x_t, cu = cuu u_t: partial sums over this lane's rows, into the sums they are added to
#pragma unroll
          for (int ti = 0; ti < NT; ti++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) px[tj] = __builtin_fma(cxx[ti][rr], L.cx[16 * ti + 4 * rr + g], px[tj]);
          if (tj == 0) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) pu = __builtin_fma(cuu[rr], L.tmp[4 * rr + g], pu);
          }
        }
        double4_t qxx[NT], qux = zero4, quu = zero4;
#pragma unroll
        for (int ti = 0; ti < NT; ti++) qxx[ti] = zero4;
#pragma unroll
        for (int ks = 0; ks < 4 * NT; ks++) {
#pragma unroll
          for (int ti = tj; ti < NT; ti++) qxx[ti] = mfma(a1t[ks >> 2][ti][ks & 3], fx[ks >> 2][tj][ks & 3], qxx[ti]);  // (tiles on and below the diagonal: see Vn)
          qux = mfma(a2t[ks >> 2][ks & 3], fx[ks >> 2][tj][ks & 3], qux);
          if (tj == 0) quu = mfma(a2t[ks >> 2][ks & 3], fu[ks >> 2][ks & 3], quu);
        }
#pragma unroll
        for (int ti = tj; ti < NT; ti++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const double val = cxx[ti][rr] + qxx[ti][rr];
            Qxx[ti][tj][rr] = (FULL || (row_in(16 * ti + 4 * rr) && col_in(tj))) ? val : 0.0;
            asm volatile("" : "+v"(Qxx[ti][tj][rr]));  // (computed HERE: sunk below the box-QP, both addends stay live across it)
          }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const double val = LQF ? qux[rr] : cxu[rr] + qux[rr];
          Qux[tj][rr] = (FULL || (mrow_in(rr) && col_in(tj))) ? val : 0.0;
          asm volatile("" : "+v"(Qux[tj][rr]));
        }
        double4_t fufu = zero4;
        if constexpr (REGV) {  // fu'fx, fu'fu: rows and columns outside the model are zero in fu, fx already
          double4_t fufx = zero4;
#pragma unroll
          for (int ks = 0; ks < 4 * NT; ks++) {
            fufx = mfma(fu[ks >> 2][ks & 3], fx[ks >> 2][tj][ks & 3], fufx);
            if (tj == 0) fufu = mfma(fu[ks >> 2][ks & 3], fu[ks >> 2][ks & 3], fufu);
          }
#pragma unroll
          for (int rr = 0; rr < 4; rr++) Quxr[tj][rr] = Qux[tj][rr] + lambda * fufx[rr];
        }
        if (tj == 0) {
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int a = 4 * rr + g, c = p;
            const bool in = FULL || (mrow_in(rr) && mcol_in);
            const double cu2 = in ? cuu[rr] : 0.0;
            quu_nat[rr] = in ? cu2 + quu[rr] : 0.0;
            L.Quu()[a + LDM * c] = quu_nat[rr];
            if constexpr (REGV)
              L.QuuF()[a + LDM * c] = in ? quu_nat[rr] + lambda * fufu[rr] : 0.0;
            else
              L.QuuF()[a + LDM * c] = in ? (cu2 + ((a == c) ? lambda : 0.0)) + quu[rr] : 0.0;
          }
        }
      }
      // Qx = cx + fx'Vx, Qu = cu + fu'Vx: the four row groups' partial sums through the scratch
#pragma unroll
      for (int tj = 0; tj < NT; tj++) red[g * RS + 16 * tj + p] = px[tj];
      red[g * RS + N + p] = pu;
      lds_sync();
      if (lane < N) {
        const double s = red4(lane);
        L.Qx[lane] = (FULL || lane < n) ? (LQF ? s : kx + s) : 0.0;
      } else if (lane < N + 16) {
        const double s = red4(lane);
        L.Qu[lane - N] = (lane - N < m) ? (LQF ? s : ku + s) : 0.0;
      }
      lds_sync();
      ILQR_W2MARK(2)
      double Xm[4];
      unsigned free_mask = 0;
      int nfR = 0, nfact = 0;
      bool slow = false;
      int result;
      if (!FULL && m <= 2) {  // one or two controls: the scalar solvers (w3_box_qp_small); K by the literal path's code below
        slow = true;
        nfact = -1;  // (no warm start for the matrix-core refinement is kept on this route)
        result = w3_box_qp_small(m, L, lane, nfR, sp.fixes);
        free_mask = (unsigned)__ballot(lane < m && L.vfree[lane]);
      } else if ((result = w3_box_qp_fast(m, L, lane, Xw, ns, Xm, free_mask)) == kQpBail) {
        slow = true;
        ns.valid = false;
        ILQR_W3COUNT(1, 1)
        result = w_box_qp(m, L, lane, nfR ILQR_W2CLOCK_PASS, &nfact, sp.fixes);
        free_mask = (unsigned)__ballot(lane < m && L.vfree[lane]);
      } else {
        ILQR_W3COUNT(0, 1)
      }
      ILQR_W2MARK(3)
      if (result < 1) {  // :371
        diverge = i;
        break;
      }
      // :373-385  K rows of free dims, natural registers K[tj][r] = K(4 r + g, 16 tj + p)
      double K[NT][4];
      const int nf = __popc(free_mask);
      auto quxk = [&](int tj, int rr) -> double {  // what the gains are solved from
        if constexpr (REGV) return Quxr[tj][rr]; else return Qux[tj][rr];
      };
      if (!slow) {
        // K = -(masked inverse) Qux: clamped rows of the inverse are zero, so those rows of K are, and its zero columns add
        // exact zeros to the k-ordered sums over the free dims
#pragma unroll
        for (int tj = 0; tj < NT; tj++) {
          double4_t acc = zero4;
#pragma unroll
          for (int ks = 0; ks < WM / 4; ks++) acc = mfma(Xm[ks], quxk(tj, ks), acc);
#pragma unroll
          for (int rr = 0; rr < 4; rr++) K[tj][rr] = -acc[rr];
        }
      } else {
        const unsigned long long fm64 = free_mask;
        if (nf > 0 && nf == nfR) {
          double* MF = L.Qf();
          if (nf == m) {
            MF = L.Minv();
          } else {
            if (lane < m && L.vfree[lane]) L.idx[__popcll(fm64 & ((1ull << lane) - 1ull))] = lane;
            for (int e = lane; e < LDM * WM; e += 64) MF[e] = 0.0;
            lds_sync();
            for (int e = lane; e < nf * nf; e += 64) {
              const int a = e % nf, b2 = e / nf;
              MF[L.idx[a] + LDM * L.idx[b2]] = L.Minv()[a + LDM * b2];
            }
            lds_sync();
          }
          double aM[WM / 4];
          ld_operand<WM / 4>([&](int i2, int k) { return MF[i2 + LDM * k]; }, lane, aM);
#pragma unroll
          for (int tj = 0; tj < NT; tj++) {
            double4_t acc = zero4;
#pragma unroll
            for (int ks = 0; ks < WM / 4; ks++) acc = mfma(aM[ks], quxk(tj, ks), acc);
#pragma unroll
            for (int rr = 0; rr < 4; rr++) K[tj][rr] = -acc[rr];
          }
        } else {  // nothing free, or a stale factor of another size (:80): through LDS, as k_backward_w does
          if (lane < m && L.vfree[lane]) L.idx[__popcll(fm64 & ((1ull << lane) - 1ull))] = lane;
          for (int c = lane >> 4; c < n; c += 4) L.K()[(lane & 15) + LDM * c] = 0;
#pragma unroll
          for (int tj = 0; tj < NT; tj++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) L.Tbuf[(4 * rr + g) + LDM * (16 * tj + p)] = quxk(tj, rr);
          lds_sync();
          if (nf > 0) {
            const int nuse = (nf < nfR) ? nf : nfR;
            for (int e = lane; e < nuse * n; e += 64) {
              const int rr = e % nuse, c = e / nuse;
              double acc = 0;
              for (int l2 = 0; l2 < nuse; l2++) acc += -L.Minv()[rr + LDM * l2] * L.Tbuf[L.idx[l2] + LDM * c];
              L.K()[L.idx[rr] + LDM * c] = acc;
            }
          }
          lds_sync();
#pragma unroll
          for (int tj = 0; tj < NT; tj++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) K[tj][rr] = L.K()[(4 * rr + g) + LDM * (16 * tj + p)];
        }
        // re-seed the inverse's warm start from the literal factor, if that was a complete one for this free set:
        // X = R^-1 R^-T scattered to the free rows / columns, the identity elsewhere
        lds_sync();
        if (nf > 0 && nf == nfR && nfact == nfR) {
          const bool fp = (free_mask >> p) & 1u;
          const int rp = __popc(free_mask & ((1u << p) - 1u));
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int a = 4 * rr + g;
            const bool fa = (free_mask >> a) & 1u;
            const int ra = __popc(free_mask & ((1u << a) - 1u));
            const double mv = L.Minv()[((fa && fp) ? ra : 0) + LDM * ((fa && fp) ? rp : 0)];
            Xw[64 * rr + lane] = (fa && fp) ? mv : ((a == p) ? 1.0 : 0.0);
          }
          ns.mask = free_mask;
          ns.valid = true;
        }
        lds_sync();
      }
      ILQR_W2MARK(4)
      // :388-389
      {
        const double d0 = wave_sum_row0(lane < m ? L.x[lane] * L.Qu[lane] : 0.0);
        double part = 0;
        if (lane < m) {
          const double rr = dot_padded([&](int a) { return 0.5 * L.x[a]; }, [&](int a) { return L.Quu()[a + LDM * lane]; });
          part = rr * L.x[lane];
        }
        dV0 += d0;
        dV1 += wave_sum_row0(part);
      }
      // T1' = Quu' K (m x n): Quu's natural registers are its A operand
      double4_t t1t[NT];
#pragma unroll
      for (int tj = 0; tj < NT; tj++) {
        t1t[tj] = zero4;
#pragma unroll
        for (int ks = 0; ks < WM / 4; ks++) t1t[tj] = mfma(quu_nat[ks], K[tj][ks], t1t[tj]);
      }
      // :391 Vx = ((Qx + T1 k) + K'Qu) + Qux'k: per-lane partial sums over the rows 4 r + g, three sums kept apart
      {
        double xq[4], qq[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          xq[rr] = L.x[4 * rr + g];
          qq[rr] = L.Qu[4 * rr + g];
        }
#pragma unroll
        for (int tj = 0; tj < NT; tj++) {
          double s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            s1 = __builtin_fma(t1t[tj][rr], xq[rr], s1);
            s2 = __builtin_fma(K[tj][rr], qq[rr], s2);
            s3 = __builtin_fma(Qux[tj][rr], xq[rr], s3);
          }
          red[(0 * 4 + g) * N + 16 * tj + p] = s1;
          red[(1 * 4 + g) * N + 16 * tj + p] = s2;
          red[(2 * 4 + g) * N + 16 * tj + p] = s3;
        }
        lds_sync();
        if (lane < N) {
          auto sum4 = [&](int q) __attribute__((always_inline)) {
            return ((red[(q * 4 + 0) * N + lane] + red[(q * 4 + 1) * N + lane]) + red[(q * 4 + 2) * N + lane]) + red[(q * 4 + 3) * N + lane];
          };
          const double vx = ((L.Qx[lane] + sum4(0)) + sum4(1)) + sum4(2);
          L.Vx[lane] = (FULL || lane < n) ? vx : 0.0;
        }
      }
      lds_sync();  // (Quu, x, Qu have been read: S may take Vn)
      ILQR_W2MARK(5)
      // :392 Vn = ((Qxx + T1 K) + K'Qux) + Qux'K ; :393 Vxx = (Vn + Vn')/2 through S.  Vn is symmetric up to rounding (Qxx = cxx + fx'Vxx fx,
      // K'Quu K, and K'Qux + Qux'K as a pair), so only its 16 x 16 tiles on and below the diagonal are formed: the tile above the
      // diagonal of the new Vxx is the transpose of the one below it (read back transposed from S), where the reference averages two
      // roundings of the same number -- 20 of a step's 188 MFMAs at n = 32.
#pragma unroll
      for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int tj = 0; tj <= ti; tj++) {
          double4_t p1 = zero4, p2 = zero4, p3 = zero4;
#pragma unroll
          for (int ks = 0; ks < WM / 4; ks++) {
            p1 = mfma(t1t[ti][ks], K[tj][ks], p1);
            p2 = mfma(K[ti][ks], Qux[tj][ks], p2);
            p3 = mfma(Qux[ti][ks], K[tj][ks], p3);
          }
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const double vn = ((Qxx[ti][tj][rr] + p1[rr]) + p2[rr]) + p3[rr];
            Vxx[ti][tj][rr] = vn;
            L.S[(16 * ti + 4 * rr + g) + LDX * (16 * tj + p)] = vn;
          }
        }
      lds_sync();
#pragma unroll
      for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int tj = 0; tj < NT; tj++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const double tr = L.S[(16 * tj + p) + LDX * (16 * ti + 4 * rr + g)];  // Vn(column, row)
            if (ti == tj) Vxx[ti][tj][rr] = 0.5 * (Vxx[ti][tj][rr] + tr);
            else if (ti < tj) Vxx[ti][tj][rr] = tr;   // (ti > tj: Vn itself)
          }
      // :396-397
      if (lane < m) {
        kb[(size_t)i * m + lane] = L.x[lane];
        L.kprev[lane] = L.x[lane];
      }
#pragma unroll
      for (int tj = 0; tj < NT; tj++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          if (FULL || (mrow_in(rr) && col_in(tj))) Kb[(size_t)i * m * n + (4 * rr + g) + m * (16 * tj + p)] = K[tj][rr];
        }
      lds_sync();
      ILQR_W2MARK(6)
    }
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
#ifdef ILQR_W2_TIMING
  clk.flush();
#endif
  // :153 / :405-412 gradient norm: mean_t max_j |k_j| / (|u_j| + 1), ascending t
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  double acc = 0;
  for (int t = 0; t < T; t++) {
    double val = -1.0;
    if (lane < m) val = fabs(kb[(size_t)t * m + lane]) / (fabs(usb[(size_t)t * m + lane]) + 1);
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) val = fmax(val, __shfl_xor(val, off, 64));
    acc += __shfl(val, 0, 64);
  }
  const double gnorm = acc / T;
  if (lane == 0) {
    v.dV[b] = dV0;
    v.dV[v.Bp + b] = dV1;
    v.diverge[b] = diverge;
    v.backpass_done[b] = done ? 1 : 0;
    v.gnorm[b] = gnorm;
    if (mode == 1) {
      v.lambda[b] = lambda;
      v.dlambda[b] = dlambda;
      if (!sp.fixed_work && gnorm < sp.tol_grad && lambda < 1e-5) {
        v.status[b] = 1;
        v.iters[b] += 1;
      }
    }
  }
}

}  // namespace ilqr
