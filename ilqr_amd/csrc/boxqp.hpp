// boxqp.hpp -- per-thread box-constrained QP (projected Newton) for small compile-time M.
//
// Device counterpart of src/boxqp.cpp:26-178 + include/boxqp.h + the mask helpers of
// include/eigen_helpers.h:15-61.  All loops are over the compile-time dimension M and fully
// unrolled, so every array lives in registers; the order-preserving compaction of the free set
// (subvec_w_ind / extract_bool_rowsandcols / rows_w_ind) is done with rank-matched selects
// instead of dynamic indexing.  Behaviour that leaks into results is kept exactly:
//   - initial value has no 1/2 (boxqp.cpp:36), loop bound is inclusive (:50);
//   - refactor only when the NUMBER of clamped dims changed (:80) -> a stale factor of equal
//     size can be applied to a different free set;
//   - Eigen's unblocked LLT (Cholesky/LLT.h:302-325) stops at a non-positive pivot and leaves
//     the remaining columns untouched; boxqp.cpp:85-88 never checks info(), so that partial
//     factor is used;
//   - a failed line search returns result 2 without updating x (:121-125).
#pragma once
#include "common.hpp"

// host+device so the very same source can be unit-tested on the CPU (tests/native)
#define ILQR_HD __host__ __device__ __forceinline__

namespace ilqr {

// 1/a for a normal-range a: v_rcp_f64 + two Newton steps (5 dependent instructions, <= 1 ulp)
// instead of the 11-instruction IEEE division sequence; the host build (unit tests) divides.
ILQR_HD double recip(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(a);
  double e = __builtin_fma(-a, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-a, r, 1.0);
  return __builtin_fma(r, e, r);
#else
  return 1.0 / a;
#endif
}
// fp32: v_rcp_f32 (1 ulp) + one Newton step
ILQR_HD float recip(float a) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_rcpf(a);
  const float e = __builtin_fmaf(-a, r, 1.0f);
  return __builtin_fmaf(r, e, r);
#else
  return 1.0f / a;
#endif
}
// 1/sqrt(a) and sqrt(a) for a normal-range a > 0: v_rsq_f64 + two Newton steps, then one correction of the root (each
// <= 1 ulp; 11 instructions for the pair instead of ~20 for the IEEE square root plus ~12 per IEEE division by it).
ILQR_HD void rsqrt_and_sqrt(double a, double& rs, double& s) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(a);
  double e = __builtin_fma(-a * y, y, 1.0);  // 1 - a y^2
  y = __builtin_fma(0.5 * y, e, y);
  e = __builtin_fma(-a * y, y, 1.0);
  y = __builtin_fma(0.5 * y, e, y);
  double r = a * y;
  const double d = __builtin_fma(-r, r, a);
  r = __builtin_fma(0.5 * y, d, r);
  rs = y;
  s = r;
#else
  s = __builtin_sqrt(a);
  rs = 1.0 / s;
#endif
}
// and / or of predicates as SELECTS on i1: `a | b` on bool is integer arithmetic on the promoted operands, which the GPU back
// end tends to carry out in vector registers (0 / 1 materialised per predicate, 16-bit logic, a compare to get a lane mask
// back); `||` would be a branch.  These stay lane masks (s_or_b64 / s_and_b64).
ILQR_HD bool p_or(bool a, bool b) { return a ? true : b; }
ILQR_HD bool p_and(bool a, bool b) { return a ? b : false; }
// type-directed math (an unqualified fabs / fmin / fmax / sqrt on a float silently picks the double
// function in a host pass and wherever only the C declarations are visible)
ILQR_HD double sqrt_of(double a) { return __builtin_sqrt(a); }
ILQR_HD float sqrt_of(float a) { return __builtin_sqrtf(a); }
ILQR_HD double abs_of(double a) { return __builtin_fabs(a); }
ILQR_HD float abs_of(float a) { return __builtin_fabsf(a); }
ILQR_HD double min_of(double a, double b) { return __builtin_fmin(a, b); }
ILQR_HD float min_of(float a, float b) { return __builtin_fminf(a, b); }
ILQR_HD double max_of(double a, double b) { return __builtin_fmax(a, b); }
ILQR_HD float max_of(float a, float b) { return __builtin_fmaxf(a, b); }
// min(max(v, lo), hi).  On the device as the two instructions: the builtins first canonicalise every operand the compiler cannot
// prove quiet (one v_max_f64 v, v, v each -- three per box-QP in the backward chains) to quiet signalling NaNs, which nothing
// here produces; for every other input, -0 / +0 and quiet NaNs included, the result is the same bits.
ILQR_HD double clamp_of(double v, double lo, double hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  double t, r;
  asm("v_max_f64 %0, %1, %2" : "=v"(t) : "v"(v), "v"(lo));
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(t), "v"(hi));
  return r;
#else
  return min_of(max_of(v, lo), hi);
#endif
}
ILQR_HD float clamp_of(float v, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  float t, r;
  asm("v_max_f32 %0, %1, %2" : "=v"(t) : "v"(v), "v"(lo));
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(hi));
  return r;
#else
  return min_of(max_of(v, lo), hi);
#endif
}

template <int M, class real>
ILQR_HD void clamp_to_limits(const real* x, const real* lo, const real* hi, real* out) {
#pragma unroll
  for (int i = 0; i < M; i++) {  // include/boxqp.h:48-51  upper.cwiseMin(x.cwiseMax(lower))
    const real a = (x[i] < lo[i]) ? lo[i] : x[i];
    out[i] = (hi[i] < a) ? hi[i] : a;
  }
}

// sqrt(gn2) < minGrad (boxqp.cpp:93-97, Eigen's norm()) without the square root: for a correctly rounded sqrt the
// test is monotone in gn2, so it equals gn2 < y* with y* the smallest value whose root rounds to >= minGrad -- one ulp
// below fl(1e-8^2) in double, found by search in float (tests/test_device_boxqp_on_host.py checks both against sqrt).
ILQR_HD bool grad_norm_below_min(double gn2) { return gn2 < 0x1.cd2b297d889bcp-54; }
ILQR_HD bool grad_norm_below_min(float gn2) { return gn2 < 0x1.cd2b28p-54f; }

template <int M, class real>
ILQR_HD real quad_cost(const real* Q, const real* c, const real* x) {
  real quad = 0, lin = 0;  // include/boxqp.h:53-55   ((0.5 x')Q) x + x.c
#pragma unroll
  for (int j = 0; j < M; j++) {
    real r = 0;
#pragma unroll
    for (int i = 0; i < M; i++) r += (real(0.5) * x[i]) * Q[i + M * j];
    quad += r * x[j];
  }
#pragma unroll
  for (int i = 0; i < M; i++) lin += x[i] * c[i];
  return quad + lin;
}

template <int M, class real>
ILQR_HD void matvec(const real* Q, const real* x, real* y) {
#pragma unroll
  for (int i = 0; i < M; i++) {
    real s = 0;
#pragma unroll
    for (int j = 0; j < M; j++) s += Q[i + M * j] * x[j];
    y[i] = s;
  }
}

// src/boxqp.cpp:143-178.  Returns failed; x_opt/v_opt are written unless the direction is not
// a descent direction (:151-154).
template <int M, class real>
ILQR_HD bool quadclamp_line_search(const real* x0, const real* dir, const real* Q,
                                                      const real* c, const real* lo, const real* hi,
                                                      real* x_opt, real& v_opt) {
  real grad[M], xr[M], xc[M];
  matvec<M>(Q, x0, grad);
  real slope = 0;
#pragma unroll
  for (int i = 0; i < M; i++) slope += dir[i] * (grad[i] + c[i]);
  if (slope >= 0) return true;
  real step = 1;
#pragma unroll
  for (int i = 0; i < M; i++) xr[i] = x0[i] + step * dir[i];
  clamp_to_limits<M>(xr, lo, hi, xc);
  real v = quad_cost<M>(Q, c, xc);
  const real old_v = quad_cost<M>(Q, c, x0);
  bool failed = false;
  // (the reference's (v - old_v) / (step * slope) < armijo without the division: step * slope < 0 here, so the test is
  //  (v - old_v) > armijo * (step * slope) up to the rounding of the quotient -- the scalar fast path's form, see below)
  while ((v - old_v) > real(kArmijo) * (step * slope)) {
    step *= real(kStepDec);
#pragma unroll
    for (int i = 0; i < M; i++) xr[i] = x0[i] + step * dir[i];
    clamp_to_limits<M>(xr, lo, hi, xc);
    v = quad_cost<M>(Q, c, xc);
    // A trial that lands on x0 itself (the direction is rounding noise, or points out of the box from a bound) has
    // v == old_v, and so has every shorter step: the reference's loop keeps failing the test down to minStep (:167-171)
    // -- unless step * slope underflows first, which ends it "accepting" the same point; either way x and the free set
    // stay as they are (result 2 or, one iteration later, 4).  Same outcome, up to 100 trips earlier.
    bool stuck = true;
#pragma unroll
    for (int i = 0; i < M; i++) stuck = p_and(stuck, xc[i] == x0[i]);
    if (step < real(kMinStep) || stuck) {
      failed = true;
      break;
    }
  }
#pragma unroll
  for (int i = 0; i < M; i++) x_opt[i] = xc[i];
  v_opt = v;
  return failed;
}

// Eigen 3.3.4 llt_inplace<Lower>::unblocked on the leading nf x nf block (ld = M).
template <int M, class real>
ILQR_HD bool llt_lower(int nf, real* A) {  // returns true if a pivot was not positive (Eigen: info() != Success)
  bool stop = false;
#pragma unroll
  for (int k = 0; k < M; k++) {
    if (k < nf && !stop) {
      real x = A[k + M * k];
      real sq = 0;
#pragma unroll
      for (int j = 0; j < M; j++)
        if (j < k) sq += A[k + M * j] * A[k + M * j];
      if (k > 0) x -= sq;
      if (x <= real(0)) {
        stop = true;
      } else {
        x = sqrt_of(x);
        A[k + M * k] = x;
#pragma unroll
        for (int i = 0; i < M; i++)
          if (i > k && i < nf) {
            real s = 0;
#pragma unroll
            for (int j = 0; j < M; j++)
              if (j < k) s += A[i + M * j] * A[k + M * j];
            real v = A[i + M * k];
            if (k > 0) v -= s;
            A[i + M * k] = v / x;
          }
      }
    }
  }
  return stop;
}

// Minv = R^-1 R^-T for the upper-triangular leading nf x nf block of R (ld = M).
// (The reference: two PartialPivLU inverses and a product, boxqp.cpp:105-112, ilqr_core.cpp:379.)
template <int M, class real>
ILQR_HD void rinv_rinvT(int nf, const real* R, real* Minv) {
  real Ri[M * M];
#pragma unroll
  for (int e = 0; e < M * M; e++) Ri[e] = 0;
#pragma unroll
  for (int j = 0; j < M; j++) {
    if (j < nf) {
      Ri[j + M * j] = real(1) / R[j + M * j];
#pragma unroll
      for (int ii = 0; ii < M; ii++) {
        const int i = j - 1 - ii;  // i = j-1 .. 0
        if (i >= 0) {
          real s = 0;
#pragma unroll
          for (int l = 0; l < M; l++)
            if (l > i && l <= j) s += R[i + M * l] * Ri[l + M * j];
          Ri[i + M * j] = -s / R[i + M * i];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < M; i++)
#pragma unroll
    for (int j = 0; j < M; j++) {
      real s = 0;
#pragma unroll
      for (int l = 0; l < M; l++)
        if (l < nf) s += Ri[i + M * l] * Ri[j + M * l];
      Minv[i + M * j] = s;
    }
}

template <int M, class real = double>
struct BoxQPResult {
  int result;
  real x[M];
  int v_free[M];  // 0/1 mask of free dims at exit
  real R[M * M]; // compact upper factor: leading nfR x nfR block, ld = M
  int nfR;
};

// src/boxqp.cpp:26-139
template <int M, class real>
ILQR_HD void box_qp(const real* Q, const real* c, const real* x0, const real* lo,
                                       const real* hi, BoxQPResult<M, real>& res, bool detect_indefinite = false) {
  real x[M], grad[M], gc[M], search[M], tmp[M];
  real clamped[M], old_clamped[M];
  clamp_to_limits<M>(x0, lo, hi, x);  // :35
  real val;
  {  // :36  x'Qx + x.c (no 1/2)
    real quad = 0, lin = 0;
#pragma unroll
    for (int j = 0; j < M; j++) {
      real r = 0;
#pragma unroll
      for (int i = 0; i < M; i++) r += x[i] * Q[i + M * j];
      quad += r * x[j];
    }
#pragma unroll
    for (int i = 0; i < M; i++) lin += x[i] * c[i];
    val = quad + lin;
  }
  real oldvalue = 0;
  int result = 0;
  int nfR = 0;
#pragma unroll
  for (int i = 0; i < M; i++) {
    clamped[i] = 0;
    old_clamped[i] = 0;
    res.v_free[i] = 0;
  }
#pragma unroll
  for (int e = 0; e < M * M; e++) res.R[e] = 0;

  real Minv[M * M];  // (R^-1 R^-T) of the factor held in res.R: the reference inverts R in every iteration (:105-112); R only
#pragma unroll       // changes when the free set does, so the product is formed there and kept -- same values
  for (int e = 0; e < M * M; e++) Minv[e] = 0;
#pragma unroll 2  // (most QPs end in their second iteration: the loop-carried copies then sit on a back edge that is rarely taken)
  for (int iter = 0; iter <= kQpMaxIter; iter++) {  // :50
    if (iter > 0 && (oldvalue - val) < real(kMinRelImprove) * abs_of(oldvalue)) {  // :54-57
      result = 4;
      break;
    }
    matvec<M>(Q, x, grad);
#pragma unroll
    for (int i = 0; i < M; i++) grad[i] += c[i];
    oldvalue = val;

    bool all_clamped = true;
    real dsum = 0;
    int rank[M];
    int nf = 0;
#pragma unroll
    for (int i = 0; i < M; i++) {  // :62-71
      old_clamped[i] = clamped[i];
      const bool cl = (abs_of(x[i] - lo[i]) < real(kClampTol) && grad[i] > 0) || (abs_of(x[i] - hi[i]) < real(kClampTol) && grad[i] < 0);
      clamped[i] = cl ? real(1) : real(0);
      res.v_free[i] = cl ? 0 : 1;
      all_clamped = all_clamped && cl;
      dsum += old_clamped[i] - clamped[i];
      rank[i] = nf;
      nf += cl ? 0 : 1;
    }
    if (all_clamped) {  // :74-77
      result = 6;
      break;
    }

    if (iter == 0 || dsum != 0) {  // :80
      real Qf[M * M];
#pragma unroll
      for (int e = 0; e < M * M; e++) Qf[e] = 0;
      // extract_bool_rowsandcols (eigen_helpers.h:46-61): Qf[rank[i]][rank[j]] = Q[i][j] for free i,j
      if constexpr (M == 2) {  // (the same selection written out: four selects instead of the 64 of the loops below)
        const bool f0 = res.v_free[0] != 0, both = f0 && res.v_free[1] != 0;
        Qf[0] = f0 ? Q[0] : Q[3];
        Qf[1] = both ? Q[1] : real(0);
        Qf[2] = both ? Q[2] : real(0);
        Qf[3] = both ? Q[3] : real(0);
      } else {
#pragma unroll
      for (int a = 0; a < M; a++)
#pragma unroll
        for (int b = 0; b < M; b++) {
          real v = 0;
#pragma unroll
          for (int i = 0; i < M; i++)
#pragma unroll
            for (int j = 0; j < M; j++)
              if (res.v_free[i] && res.v_free[j] && rank[i] == a && rank[j] == b) v = Q[i + M * j];
          Qf[a + M * b] = v;
        }
      }
      const bool indefinite = llt_lower<M>(nf, Qf);  // :85 (info() ignored ...
      if (detect_indefinite && indefinite) {       // ... unless the caller opted into the fix: result -1)
        result = -1;
        break;
      }
#pragma unroll
      for (int a = 0; a < M; a++)
#pragma unroll
        for (int b = 0; b < M; b++) res.R[a + M * b] = (a <= b && b < nf) ? Qf[b + M * a] : real(0);  // :86-88
      nfR = nf;
      rinv_rinvT<M>(nfR, res.R, Minv);
    }

    real gn2 = 0;  // :93-97
#pragma unroll
    for (int i = 0; i < M; i++)
      if (res.v_free[i]) gn2 += grad[i] * grad[i];
    if (grad_norm_below_min(gn2)) {
      result = 5;
      break;
    }

    // :100  grad_clamped = Q (x .* clamped) + c
#pragma unroll
    for (int i = 0; i < M; i++) tmp[i] = x[i] * clamped[i];
    matvec<M>(Q, tmp, gc);
#pragma unroll
    for (int i = 0; i < M; i++) gc[i] += c[i];

    // :103-119  search(free) = -(R^-1 R^-T) gc(free) - x(free)
    real gfree[M], xfree[M], sfree[M];
    if constexpr (M == 2) {
      const bool f0 = res.v_free[0] != 0, both = f0 && res.v_free[1] != 0;
      gfree[0] = f0 ? gc[0] : gc[1];
      xfree[0] = f0 ? x[0] : x[1];
      gfree[1] = both ? gc[1] : real(0);
      xfree[1] = both ? x[1] : real(0);
    } else {
#pragma unroll
    for (int a = 0; a < M; a++) {
      real g = 0, xx = 0;
#pragma unroll
      for (int i = 0; i < M; i++)
        if (res.v_free[i] && rank[i] == a) {
          g = gc[i];
          xx = x[i];
        }
      gfree[a] = g;
      xfree[a] = xx;
    }
    }
#pragma unroll
    for (int a = 0; a < M; a++) {
      real s = 0;
#pragma unroll
      for (int l = 0; l < M; l++)
        if (l < nfR) s += -Minv[a + M * l] * gfree[l];
      sfree[a] = s - xfree[a];
    }
    if constexpr (M == 2) {
      const bool f0 = res.v_free[0] != 0, f1 = res.v_free[1] != 0;
      search[0] = f0 ? sfree[0] : real(0);
      search[1] = f1 ? (f0 ? sfree[1] : sfree[0]) : real(0);
    } else {
#pragma unroll
    for (int i = 0; i < M; i++) {
      real s = 0;
#pragma unroll
      for (int a = 0; a < M; a++)
        if (res.v_free[i] && rank[i] == a) s = sfree[a];
      search[i] = s;
    }
    }

    real lx[M], lv = 0;
    const bool failed = quadclamp_line_search<M>(x, search, Q, c, lo, hi, lx, lv);  // :121
    if (failed) {  // :122-125
      result = 2;
      break;
    }
#pragma unroll
    for (int i = 0; i < M; i++) x[i] = lx[i];  // :133-134
    val = lv;
  }
#pragma unroll
  for (int i = 0; i < M; i++) res.x[i] = x[i];
  res.result = result;
  res.nfR = nfR;
}


// ------------------------------------------------------------------------------------------
// M = 2 (the double integrator, include/double_integrator.h:16-17): box_qp<2> written out for two scalars.
// The generic solver spends most of a 2 x 2 QP in Eigen's recipe for (R^-1 R^-T) -- Cholesky, two triangular
// inverses, a product: two square roots and five IEEE divisions, formed in the loop and once more by the caller for
// K (:373-385) -- and in rank-matched selects that emulate the mask helpers.  Here the free set is one of three cases
// (both, only 0, only 1), the compaction is a pair of selects, and (R^-1 R^-T) of a positive definite free block is its
// inverse by the adjugate: 1 / Q_ff for one free dimension, adj(Q_ff) / det for two (one rcp + Newton; rounding-level
// difference from the Cholesky route, as for M = 1 below).  What leaks into results is kept:
//   - initial value without 1/2, inclusive loop bound, stall / all-clamped / gradient / line-search exits in the
//     reference's order (boxqp.cpp:50-125);
//   - refactor only when the NUMBER of clamped dimensions changed (:80): with one free dimension the factor may belong
//     to the OTHER index -- Minv is kept compact (rank-indexed) exactly like R_free;
//   - Eigen's unchecked partial Cholesky (LLT.h:318-319) when the free block is not positive definite: that rare case
//     runs the generic llt_lower<2> / rinv_rinvT<2> on the compacted block, so its partial factor is the reference's.
// Outputs: x, the free mask, and Minv (what the caller needs for K) instead of R.
// ------------------------------------------------------------------------------------------
template <class real>
struct BoxQP2Result {
  int result;
  real x[2];
  bool free0, free1;
  real m00, m01, m11;  // compact (R^-1 R^-T) of the factor held at exit: leading nfR x nfR block
  int nfR;
};

template <class real>
ILQR_HD void box_qp2_loop(const real* Q, const real* c, const real* x0, const real* lo, const real* hi, BoxQP2Result<real>& res,
                          bool detect_indefinite = false) {
  const real q00 = Q[0], q10 = Q[1], q01 = Q[2], q11 = Q[3];
  real x[2];
  clamp_to_limits<2>(x0, lo, hi, x);  // :35
  real val = ((x[0] * q00 + x[1] * q10) * x[0] + (x[0] * q01 + x[1] * q11) * x[1]) + (x[0] * c[0] + x[1] * c[1]);  // :36 (no 1/2)
  real oldvalue = 0;
  int result = 0, nfR = 0;
  bool cl0 = false, cl1 = false;
  real m00 = 0, m01 = 0, m11 = 0;
#pragma unroll 2  // (most QPs end in their second iteration)
  for (int iter = 0; iter <= kQpMaxIter; iter++) {  // :50
    if (iter > 0 && (oldvalue - val) < real(kMinRelImprove) * abs_of(oldvalue)) {  // :54-57
      result = 4;
      break;
    }
    const real g0 = (q00 * x[0] + q01 * x[1]) + c[0], g1 = (q10 * x[0] + q11 * x[1]) + c[1];
    oldvalue = val;
    const int n_old = (int)cl0 + (int)cl1;
    cl0 = p_or(p_and(abs_of(x[0] - lo[0]) < real(kClampTol), g0 > 0), p_and(abs_of(x[0] - hi[0]) < real(kClampTol), g0 < 0));  // :62-71
    cl1 = p_or(p_and(abs_of(x[1] - lo[1]) < real(kClampTol), g1 > 0), p_and(abs_of(x[1] - hi[1]) < real(kClampTol), g1 < 0));
    if (p_and(cl0, cl1)) {  // :74-77
      result = 6;
      break;
    }
    const bool f0 = !cl0, both = p_and(f0, !cl1);
    if (iter == 0 || n_old != (int)cl0 + (int)cl1) {  // :80
      // the free block, compacted (eigen_helpers.h:46-61)
      const real a00 = f0 ? q00 : q11, a10 = both ? q10 : real(0), a11 = both ? q11 : real(0);
      const real det = a00 * a11 - a10 * a10;
      const bool pd = p_and(a00 > real(0), p_or(!both, det > real(0)));
      if (detect_indefinite && !pd) {  // opt-in fix: a failed factorisation ends the QP
        result = -1;
        break;
      }
      nfR = both ? 2 : 1;
      if (__builtin_expect(pd, 1)) {
        if (both) {
          const real rd = recip(det);
          m00 = a11 * rd;
          m01 = -a10 * rd;
          m11 = a00 * rd;
        } else {
          m00 = recip(a00);
          m01 = m11 = 0;
        }
      } else {  // Eigen's partial factor, literally
        real Qf[4] = {a00, a10, both ? q01 : real(0), a11}, R[4], Mi[4];
        llt_lower<2>(nfR, Qf);
        R[0] = Qf[0];
        R[1] = 0;
        R[2] = both ? Qf[1] : real(0);
        R[3] = both ? Qf[3] : real(0);
        rinv_rinvT<2>(nfR, R, Mi);
        m00 = Mi[0];
        m01 = Mi[1];
        m11 = Mi[3];
      }
    }
    real gn2 = 0;  // :93-97
    if (!cl0) gn2 += g0 * g0;
    if (!cl1) gn2 += g1 * g1;
    if (grad_norm_below_min(gn2)) {
      result = 5;
      break;
    }
    // :100  grad_clamped = Q (x .* clamped) + c
    const real t0 = cl0 ? x[0] : real(0), t1 = cl1 ? x[1] : real(0);
    const real gc0 = (q00 * t0 + q01 * t1) + c[0], gc1 = (q10 * t0 + q11 * t1) + c[1];
    // :103-119  search(free) = -(R^-1 R^-T) gc(free) - x(free), by rank
    const real gf0 = f0 ? gc0 : gc1, xf0 = f0 ? x[0] : x[1];
    real sf0, sf1 = 0;
    if (nfR == 2) {
      const real gf1 = both ? gc1 : real(0), xf1 = both ? x[1] : real(0);
      sf0 = (-m00 * gf0 + -m01 * gf1) - xf0;
      sf1 = (-m01 * gf0 + -m11 * gf1) - xf1;
    } else {
      sf0 = -m00 * gf0 - xf0;
    }
    real search[2];
    search[0] = f0 ? sf0 : real(0);
    search[1] = cl1 ? real(0) : (f0 ? sf1 : sf0);
    real lx[2], lv = 0;
    if (quadclamp_line_search<2>(x, search, Q, c, lo, hi, lx, lv)) {  // :121-125
      result = 2;
      break;
    }
    x[0] = lx[0];  // :133-134
    x[1] = lx[1];
    val = lv;
  }
  res.result = result;
  res.x[0] = x[0];
  res.x[1] = x[1];
  res.free0 = !cl0;
  res.free1 = !cl1;
  res.m00 = m00;
  res.m01 = m01;
  res.m11 = m11;
  res.nfR = nfR;
}

// What the double integrator's QPs actually do (counted over the bench workload on the CPU: 286 652 QPs): 13.5 %
// find both controls clamped in iteration 0 (result 6); 86 % take ONE projected-Newton step whose unit step passes the
// Armijo test (99.87 % of all line searches do) and leave iteration 1 through the gradient test (result 5); a few per
// mille do anything else.  box_qp2 therefore runs iteration 0 with the unit step only, and iteration 1 as far as its
// gradient test, straight-line -- the reference's exits as predicates in the reference's order, the expressions of
// box_qp2_loop -- and hands every QP that needs a backtracking search, an indefinite block's partial factor or a second
// step to box_qp2_loop from the start (the loop with its six divergent exits and the sequential line search was 49 % of
// the double integrator's backward pass: 0.183 -> 0.094 ms without any QP).
template <class real>
ILQR_HD void box_qp2(const real* Q, const real* c, const real* x0, const real* lo, const real* hi, BoxQP2Result<real>& res,
                     bool detect_indefinite = false) {
  const real q00 = Q[0], q10 = Q[1], q01 = Q[2], q11 = Q[3];
  real x[2];
  clamp_to_limits<2>(x0, lo, hi, x);  // :35
  const real val0 = ((x[0] * q00 + x[1] * q10) * x[0] + (x[0] * q01 + x[1] * q11) * x[1]) + (x[0] * c[0] + x[1] * c[1]);  // :36 (no 1/2)
  auto clamp_flag = [&](real xi, real gi, real l, real h) {  // :62-71
    return p_or(p_and(abs_of(xi - l) < real(kClampTol), gi > 0), p_and(abs_of(xi - h) < real(kClampTol), gi < 0));
  };
  struct Factor {
    real m00, m01, m11;
    bool pd, both;
  };
  auto factor = [&](bool c0, bool c1) {  // :80-91 for a positive definite free block (the adjugate; see box_qp2_loop)
    Factor f;
    const bool f0 = !c0;
    f.both = p_and(f0, !c1);
    const real a00 = f0 ? q00 : q11, a10 = f.both ? q10 : real(0), a11 = f.both ? q11 : real(0);
    const real det = a00 * a11 - a10 * a10;
    f.pd = p_and(a00 > real(0), p_or(!f.both, det > real(0)));
    const real rd = recip(f.both ? det : a00);
    f.m00 = f.both ? a11 * rd : rd;
    f.m01 = f.both ? -a10 * rd : real(0);
    f.m11 = f.both ? a00 * rd : real(0);
    return f;
  };
  // ---- iteration 0
  const real g0 = (q00 * x[0] + q01 * x[1]) + c[0], g1 = (q10 * x[0] + q11 * x[1]) + c[1];
  const bool c0 = clamp_flag(x[0], g0, lo[0], hi[0]), c1 = clamp_flag(x[1], g1, lo[1], hi[1]);
  const bool ex6 = p_and(c0, c1);  // :74-77
  const Factor F = factor(c0, c1);
  const bool ex_indef = p_and(!ex6, p_and(detect_indefinite, !F.pd));  // opt-in fix: a failed factorisation ends the QP
  bool slow = p_and(!ex6, p_and(!detect_indefinite, !F.pd));           // Eigen's partial factor: the loop has it
  const bool f0 = !c0;
  real gn2 = 0;  // :93-97
  if (!c0) gn2 += g0 * g0;
  if (!c1) gn2 += g1 * g1;
  const bool alive_f = p_and(!ex6, F.pd);
  const bool ex5 = p_and(alive_f, grad_norm_below_min(gn2));
  // :100-119
  const real t0 = c0 ? x[0] : real(0), t1 = c1 ? x[1] : real(0);
  const real gc0 = (q00 * t0 + q01 * t1) + c[0], gc1 = (q10 * t0 + q11 * t1) + c[1];
  const real gf0 = f0 ? gc0 : gc1, xf0 = f0 ? x[0] : x[1];
  const real gf1 = F.both ? gc1 : real(0), xf1 = F.both ? x[1] : real(0);
  // (one free dimension: m01 = 0 and gf1 = 0 make this the loop's -m00 * gf0 - xf0 exactly)
  const real sf0 = (-F.m00 * gf0 + -F.m01 * gf1) - xf0;
  const real sf1 = F.both ? (-F.m01 * gf0 + -F.m11 * gf1) - xf1 : real(0);
  real search[2];
  search[0] = f0 ? sf0 : real(0);
  search[1] = c1 ? real(0) : (f0 ? sf1 : sf0);
  // :143-178 with the unit step only
  const real slope = search[0] * g0 + search[1] * g1;
  const bool alive_s = p_and(alive_f, !ex5);
  const bool ex2 = p_and(alive_s, slope >= 0);
  real xr[2] = {x[0] + search[0], x[1] + search[1]}, x1[2];
  clamp_to_limits<2>(xr, lo, hi, x1);
  const real v1 = quad_cost<2>(Q, c, x1);
  const real old_v = quad_cost<2>(Q, c, x);
  const bool alive_l = p_and(alive_s, !(slope >= 0));
  slow = p_or(slow, p_and(alive_l, (v1 - old_v) > real(kArmijo) * (real(1) * slope)));  // backtracking: the loop's business
  const bool stepped = p_and(alive_l, !slow);
  // ---- iteration 1 as far as its gradient test, for the QPs that took the unit step
  const bool ex4 = p_and(stepped, (val0 - v1) < real(kMinRelImprove) * abs_of(val0));  // :54-57 (oldvalue = val0, val = v1)
  const real h0 = (q00 * x1[0] + q01 * x1[1]) + c[0], h1 = (q10 * x1[0] + q11 * x1[1]) + c[1];
  const bool d0 = clamp_flag(x1[0], h0, lo[0], hi[0]), d1 = clamp_flag(x1[1], h1, lo[1], hi[1]);
  const bool alive_1 = p_and(stepped, !ex4);
  const bool ex6b = p_and(alive_1, p_and(d0, d1));
  const bool refactor = ((int)c0 + (int)c1) != ((int)d0 + (int)d1);  // :80
  const Factor G = factor(d0, d1);
  const bool alive_2 = p_and(alive_1, !p_and(d0, d1));
  const bool ex_indef_b = p_and(alive_2, p_and(refactor, p_and(detect_indefinite, !G.pd)));
  slow = p_or(slow, p_and(alive_2, p_and(refactor, p_and(!detect_indefinite, !G.pd))));
  real hn2 = 0;
  if (!d0) hn2 += h0 * h0;
  if (!d1) hn2 += h1 * h1;
  const bool alive_3 = p_and(alive_2, p_or(!refactor, G.pd));
  const bool ex5b = p_and(alive_3, grad_norm_below_min(hn2));
  slow = p_or(slow, p_and(alive_3, !ex5b));  // a second step: the loop
  if (__builtin_expect(slow, 0)) {
    box_qp2_loop(Q, c, x0, lo, hi, res, detect_indefinite);
    return;
  }
  // what the loop leaves at each exit
  const bool in1 = stepped;                                  // left in iteration 1: x = x1
  const bool cl_b = alive_1;                                 // ... past :62-71 of iteration 1: the clamp flags are d
  const bool fac_b = p_and(p_and(alive_2, refactor), G.pd);  // ... having refactored
  const bool fac_a = alive_f;                                // iteration 0 got as far as its factor
  res.result = ex6 ? 6 : ex_indef ? -1 : ex5 ? 5 : ex2 ? 2 : ex4 ? 4 : ex6b ? 6 : ex_indef_b ? -1 : 5;
  res.x[0] = in1 ? x1[0] : x[0];
  res.x[1] = in1 ? x1[1] : x[1];
  res.free0 = !(cl_b ? d0 : c0);
  res.free1 = !(cl_b ? d1 : c1);
  res.m00 = fac_b ? G.m00 : (fac_a ? F.m00 : real(0));
  res.m01 = fac_b ? G.m01 : (fac_a ? F.m01 : real(0));
  res.m11 = fac_b ? G.m11 : (fac_a ? F.m11 : real(0));
  res.nfR = fac_b ? (G.both ? 2 : 1) : (fac_a ? (F.both ? 2 : 1) : 0);
}

// ------------------------------------------------------------------------------------------
// M = 1 fast path (acrobot, the headline configuration): the same projected-Newton iteration as
// box_qp<1>, written out for a scalar so that the dependent chain is short.  Differences from
// the literal restatement are all at rounding level and documented here:
//   - the factor R = sqrt(Q) is never formed: (R^-1 R^-T) = 1/Q for Q > 0, and 1/Q^2 for the
//     unchecked-LLT-failure case Q <= 0 where the reference ends up with R = Q;
//   - |g| < minGrad instead of sqrt(g*g) < minGrad;
//   - the Armijo test (v - old_v)/(step*slope) < 0.1 is evaluated without the division as
//     (v - old_v) > 0.1*(step*slope), valid because step*slope < 0 on that path;
//   - grad_clamped = Q*(x*0) + c is taken as c.
// Outputs: x (= k), free (v_free[0]), minv = (R^-1 R^-T) of the factor held at exit.
// ------------------------------------------------------------------------------------------
template <class real>
ILQR_HD int box_qp_scalar(real Q, real c, real x0, real lo, real hi, real& x_out, int& free_out,
                          real& minv_out) {
  real x = (x0 < lo) ? lo : x0;
  x = (hi < x) ? hi : x;
  real val = (x * Q) * x + x * c;  // boxqp.cpp:36 (no 1/2)
  real oldvalue = 0;
  const real minv = (Q > real(0)) ? real(1) / Q : real(1) / (Q * Q);
  int result = 0;
  int free_ = 0;
  for (int iter = 0; iter <= kQpMaxIter; iter++) {
    if (iter > 0 && (oldvalue - val) < real(kMinRelImprove) * abs_of(oldvalue)) {
      result = 4;
      break;
    }
    const real grad = Q * x + c;
    oldvalue = val;
    const bool cl = (abs_of(x - lo) < real(kClampTol) && grad > 0) || (abs_of(x - hi) < real(kClampTol) && grad < 0);
    if (cl) {
      free_ = 0;
      result = 6;
      break;
    }
    free_ = 1;
    if (abs_of(grad) < real(kMinGrad)) {
      result = 5;
      break;
    }
    const real search = -minv * c - x;
    const real slope = search * grad;
    if (slope >= 0) {
      result = 2;
      break;
    }
    real step = 1;
    real xc = x + step * search;
    xc = (xc < lo) ? lo : xc;
    xc = (hi < xc) ? hi : xc;
    real v = ((real(0.5) * xc) * Q) * xc + xc * c;
    const real old_v = ((real(0.5) * x) * Q) * x + x * c;
    bool failed = false;
    while ((v - old_v) > real(kArmijo) * (step * slope)) {
      step *= real(kStepDec);
      xc = x + step * search;
      xc = (xc < lo) ? lo : xc;
      xc = (hi < xc) ? hi : xc;
      v = ((real(0.5) * xc) * Q) * xc + xc * c;
      if (step < real(kMinStep)) {
        failed = true;
        break;
      }
    }
    if (failed) {
      result = 2;
      break;
    }
    x = xc;
    val = v;
  }
  x_out = x;
  free_out = free_;
  minv_out = minv;
  return result;
}


// Mostly straight-line evaluation of the first two projected-Newton iterations of box_qp_scalar.
// With one wavefront holding 16 trajectories, data-dependent loops cost every lane the worst
// lane's trip count plus an exec-mask round trip per branch, while on the acrobot workload every
// QP leaves within two iterations through one of:
//   A  all clamped at iter 0 (result 6, ~85 % with u in [-1.5,1.5])
//   B  |grad| < minGrad at iter 0 (5)       C  not a descent direction (2)
//   D  no improvement at iter 1 (4)         E  clamped at iter 1 (6)       F  |grad| < minGrad at iter 1 (5)
//   G  iter 1's direction is not a descent direction (2) -- the usual exit of an interior Newton step in float
//   H  iter 1's unit trial is x1 itself (2) -- the direction is below half an ulp: late in a solve, Quu ~ 1e12
// The evaluation is split in three so that the kernel can replace the sequential Armijo
// backtracking (a Newton step truncated by a bound to < ~10 % of its length fails the test at
// step 1 -- ~10 % of the QPs, i.e. most wavefronts) by a quad-parallel search:
//   qp1_begin  iteration 0 up to the unit-step trial point
//   qp1_backtrack_seq  the loop of boxqp.cpp:161-173 as written
//   qp1_finish iteration 1's exit tests and the result ladder
//   qp1_continue  iterations 1, 2, ... for the QPs that leave through none of the six exits
// box_qp_scalar_fast composes them sequentially (host tests).  qp1_finish returns kQpGoesOn when
// the QP has to go on: the caller then runs qp1_continue.
constexpr int kQpGoesOn = -100;  // qp1_finish: none of the fast path's exits applies (not a boxQP result code; -1 is: indefinite)

template <class real>
struct QP1StateT {
  real Q, c, lo, hi;
  real x, val0, g0, minv, search, slope, old_v;
  real x1, v1, step;
  bool clA, exB, exC, early, ls_failed;
  bool indef;  // opt-in fix: Q <= 0 with a free dimension -> the Cholesky factorisation fails -> result -1
};

template <class real>
ILQR_HD real qp1_value(const QP1StateT<real>& q, real xx) { return ((real(0.5) * xx) * q.Q) * xx + xx * q.c; }
template <class real>
ILQR_HD real qp1_trial(const QP1StateT<real>& q, real step) { return clamp_of(q.x + step * q.search, q.lo, q.hi); }
// Armijo test of boxqp.cpp:161 without the division (step*slope < 0 on this path)
template <class real>
ILQR_HD bool qp1_armijo_fails(const QP1StateT<real>& q, real v, real step) {
  return (v - q.old_v) > real(kArmijo) * (step * q.slope);
}

template <bool EVAL_UNIT = true, class real>
ILQR_HD void qp1_begin(real Q, real c, real x0, real lo, real hi, QP1StateT<real>& q, bool detect_indefinite = false) {
  q.Q = Q;
  q.c = c;
  q.lo = lo;
  q.hi = hi;
  q.x = clamp_of(x0, lo, hi);  // == clamp_to_limits for non-NaN input
  q.val0 = (q.x * Q) * q.x + q.x * c;  // boxqp.cpp:36 (no 1/2)
  q.g0 = Q * q.x + c;
  const real den = (Q > real(0)) ? Q : Q * Q;
  q.minv = recip(den);
  // (p_and / p_or: no short-circuit branches in the wavefront's instruction stream, no integer arithmetic on promoted bools)
  q.clA = p_or(p_and(abs_of(q.x - lo) < real(kClampTol), q.g0 > 0), p_and(abs_of(q.x - hi) < real(kClampTol), q.g0 < 0));
  q.exB = abs_of(q.g0) < real(kMinGrad);
  q.search = -q.minv * c - q.x;
  q.slope = q.search * q.g0;
  q.exC = q.slope >= 0;
  q.indef = p_and(detect_indefinite, !(Q > real(0)));
  q.early = p_or(p_or(q.clA, q.indef), p_or(q.exB, q.exC));
  q.step = 1;
  if (EVAL_UNIT) {
    q.x1 = qp1_trial(q, real(1));
    q.v1 = qp1_value(q, q.x1);
  } else {
    q.x1 = q.x;
    q.v1 = 0;
  }
  q.old_v = qp1_value(q, q.x);
  q.ls_failed = false;
}

template <class real>
ILQR_HD void qp1_backtrack_seq(QP1StateT<real>& q) {  // boxqp.cpp:161-173
  while (!q.early && qp1_armijo_fails(q, q.v1, q.step)) {
    q.step *= real(kStepDec);
    q.x1 = qp1_trial(q, q.step);
    q.v1 = qp1_value(q, q.x1);
    // A trial that lands on x itself (the step is below half an ulp of x: the search direction is
    // rounding noise) has v1 == old_v, and so has every shorter step: the reference's loop keeps
    // failing the test until step < minStep (:167-171).  Same outcome, ~90 trips earlier.
    if (q.step < real(kMinStep) || q.x1 == q.x) {
      q.ls_failed = true;
      break;
    }
  }
}

template <class real>
ILQR_HD int qp1_finish(const QP1StateT<real>& q, real& x_out, int& free_out, real& minv_out) {
  const bool exD = (q.val0 - q.v1) < real(kMinRelImprove) * abs_of(q.val0);
  const real g1 = q.Q * q.x1 + q.c;
  const bool clE = ((abs_of(q.x1 - q.lo) < real(kClampTol)) & (g1 > 0)) | ((abs_of(q.x1 - q.hi) < real(kClampTol)) & (g1 < 0));
  const bool exF = abs_of(g1) < real(kMinGrad);
  // G: iteration 1's own search direction is not a descent direction (boxqp.cpp:150-153 -> result 2, x kept).
  // After an interior Newton step x1 IS the optimum to rounding, so search = -minv c - x1 is 0 or an ulp of
  // either sign.  In fp64 exit F fires first (|g1| ~ 1e-16 |c|); in float |g1| ~ 1e-7 |c| never passes
  // minGrad = 1e-8, and without this exit every unclamped step paid the data-dependent continue loop.
  const real search1 = -q.minv * q.c - q.x1;
  const real slope1 = search1 * g1;
  const bool exG = slope1 >= real(0);
  // H: iteration 1's unit trial lands on x1 itself.  Late in a solve Quu reaches 1e12+, x1 sits on the optimum to an
  // ulp and |g1| ~ Quu ulp is still above minGrad; g1 = Q x1 + c and search1 = -c/Q - x1 then carry independent
  // rounding noise, so x1 on a limit can be "free" by the sign of g1 while search1 points out of the box (the common
  // case, measured), or search1 is below half an ulp of x1.  Every shorter step lands on x1 too, the Armijo ratio is 0
  // at every k, the reference's loop runs down to minStep and reports failure (boxqp.cpp:167-171 -> result 2, x
  // kept).  Same outcome without the continue loop: 100-iteration average 0.93 -> 0.91 ms (fp64), 0.84 -> 0.76 ms
  // (fp32), for 4 more instructions per step in the first iterations (0.566 -> 0.571 ms).
  const bool exH = clamp_of(q.x1 + search1, q.lo, q.hi) == q.x1;
  minv_out = q.minv;
  // the reference's order of tests, as selects (no branches)
  const bool stay = q.clA | q.indef | q.exB | q.exC | q.ls_failed;  // x is not updated
  const int inner = exD ? 4 : (clE ? 6 : (exF ? 5 : ((exG | exH) ? 2 : kQpGoesOn)));
  const int outer = q.clA ? 6 : (q.indef ? -1 : (q.exB ? 5 : 2));  // (the factorisation comes before the gradient test, boxqp.cpp:80-97)
  x_out = stay ? q.x : q.x1;
  free_out = (q.clA | (!stay & !exD & clE)) ? 0 : 1;
  return stay ? outer : inner;
}

// qp1_finish for the kernels: the same exit tests, but instead of the reference's result code (4 / 5 / 6 / 2 / -1 / "goes on") the
// two facts the backward pass uses -- success (code >= 1) and "goes on" -- as predicates: the code ladder was ten selects per step.
template <class real>
ILQR_HD bool qp1_finish_ok(const QP1StateT<real>& q, real& x_out, int& free_out, real& minv_out, bool& goes_on) {
  const bool exD = (q.val0 - q.v1) < real(kMinRelImprove) * abs_of(q.val0);
  const real g1 = q.Q * q.x1 + q.c;
  const bool clE = p_or(p_and(abs_of(q.x1 - q.lo) < real(kClampTol), g1 > 0), p_and(abs_of(q.x1 - q.hi) < real(kClampTol), g1 < 0));
  const bool exF = abs_of(g1) < real(kMinGrad);
  // G: iteration 1's own search direction is not a descent direction (boxqp.cpp:150-153 -> result 2, x kept).
  // After an interior Newton step x1 IS the optimum to rounding, so search = -minv c - x1 is 0 or an ulp of
  // either sign.  In fp64 exit F fires first (|g1| ~ 1e-16 |c|); in float |g1| ~ 1e-7 |c| never passes
  // minGrad = 1e-8, and without this exit every unclamped step paid the data-dependent continue loop.
  const real search1 = -q.minv * q.c - q.x1;
  const real slope1 = search1 * g1;
  const bool exG = slope1 >= real(0);
  // H: iteration 1's unit trial lands on x1 itself.  Late in a solve Quu reaches 1e12+, x1 sits on the optimum to an
  // ulp and |g1| ~ Quu ulp is still above minGrad; g1 = Q x1 + c and search1 = -c/Q - x1 then carry independent
  // rounding noise, so x1 on a limit can be "free" by the sign of g1 while search1 points out of the box (the common
  // case, measured), or search1 is below half an ulp of x1.  Every shorter step lands on x1 too, the Armijo ratio is 0
  // at every k, the reference's loop runs down to minStep and reports failure (boxqp.cpp:167-171 -> result 2, x
  // kept).  Same outcome without the continue loop: 100-iteration average 0.93 -> 0.91 ms (fp64), 0.84 -> 0.76 ms
  // (fp32), for 4 more instructions per step in the first iterations (0.566 -> 0.571 ms).
  const bool exH = clamp_of(q.x1 + search1, q.lo, q.hi) == q.x1;
  minv_out = q.minv;
  // the reference's order of tests, as selects (no branches)
  const bool stay = p_or(p_or(p_or(q.clA, q.indef), p_or(q.exB, q.exC)), q.ls_failed);  // x is not updated
  // what the kernels need of the result code: does the QP go on, and (if not) did it succeed (result >= 1)?  Every exit
  // code of the ladder in qp1_finish is >= 1 except -1 (indefinite free block with the opt-in fix, tested after "all clamped")
  x_out = stay ? q.x : q.x1;
  free_out = p_or(q.clA, p_and(p_and(!stay, !exD), clE)) ? 0 : 1;
  goes_on = p_and(!stay, !p_or(p_or(exD, clE), p_or(exF, p_or(exG, exH))));
  return p_or(!stay, p_or(q.clA, !q.indef));
}

// Iterations >= 1 of the loop of box_qp_scalar, continued from the state the two-iteration fast
// path leaves when none of its exits applies (qp1_finish returned -1): x = q.x1, val = q.v1,
// oldvalue = q.val0.  (Iteration 1's exit tests are evaluated again -- same expressions, same
// answers -- so the loop body is the reference's as written.)  `line_search(q)` performs
// boxqp.cpp:143-178 for the state's x / search / slope / old_v and leaves x1, v1, ls_failed: the
// kernel passes the quad-parallel Armijo search, the host tests the sequential loop.  QPs that
// need this are the ones whose trial point comes inside the box before the Armijo test passes on
// the bound: the Newton target stays outside, so they creep towards the bound over several
// iterations.  Restarting the literal loop from iteration 0 for them (as the first version did)
// made their wavefront the slowest of the launch once a solve had run ~20 iterations.
template <class real, class LineSearch>
ILQR_HD int qp1_continue(QP1StateT<real>& q, LineSearch line_search, real& x_out, int& free_out) {
  real x = q.x1, val = q.v1, oldvalue = q.val0;
  int result = 0, free_ = 1;
  {
    // The usual reason to be here late in a solve: x1 is the optimum to rounding, but with Quu ~ 1e10 its gradient
    // Quu * (an ulp of error) is still above minGrad and the Newton step of iteration 1, a few 1e-17, still moves x.
    // Iteration 1 then accepts the unit step and iteration 2 leaves through the improvement test (:54-57, result 4)
    // Those two iterations written out (same expressions, same order; taken only if that IS what happens): the loop
    // below with its line search cost ~6000 cycles of the wavefront for it, a quarter of the late-solve step.
    const real g1 = q.Q * x + q.c;
    const real s1 = -q.minv * q.c - x;
    const real slope1 = s1 * g1;
    const real x2 = clamp_of(x + real(1) * s1, q.lo, q.hi);  // qp1_trial at step 1
    const real v2 = qp1_value(q, x2);
    const bool unit_passes = !((v2 - val) > real(kArmijo) * (real(1) * slope1));  // qp1_armijo_fails, old_v = value(x1) = val
    const bool moved = x2 != x;  // (x2 == x is exit H's business: the caller has excluded it, but stay exact)
    const bool tiny = (val - v2) < real(kMinRelImprove) * abs_of(val);  // iteration 2's first test
    // ... or, three times in four (counted), through the gradient test (:93-97, result 5): the step was a real one and
    // x2 is the optimum to rounding.
    const real g2 = q.Q * x2 + q.c;
    const bool cl2 = p_or(p_and(abs_of(x2 - q.lo) < real(kClampTol), g2 > 0), p_and(abs_of(x2 - q.hi) < real(kClampTol), g2 < 0));
    const bool flat2 = abs_of(g2) < real(kMinGrad);
    // ... or -- float's usual one, where |g| < 1e-8 is out of reach -- because iteration 2's own direction is no descent
    // direction (:150-153, result 2, x2 kept).
    const real s2 = -q.minv * q.c - x2;
    const bool nodesc2 = (s2 * g2) >= real(0);
    const bool stepped = p_and(p_and(unit_passes, moved), slope1 < real(0));
    if (p_and(stepped, p_or(tiny, p_and(!cl2, p_or(flat2, nodesc2))))) {
      x_out = x2;
      free_out = 1;
      return tiny ? 4 : (flat2 ? 5 : 2);
    }
  }
  for (int iter = 1; iter <= kQpMaxIter; iter++) {
    if ((oldvalue - val) < real(kMinRelImprove) * abs_of(oldvalue)) {  // boxqp.cpp:54-57 (iter > 0 here)
      result = 4;
      break;
    }
    const real grad = q.Q * x + q.c;
    oldvalue = val;
    const bool cl = p_or(p_and(abs_of(x - q.lo) < real(kClampTol), grad > 0), p_and(abs_of(x - q.hi) < real(kClampTol), grad < 0));
    if (cl) {  // :74-77
      free_ = 0;
      result = 6;
      break;
    }
    free_ = 1;
    if (abs_of(grad) < real(kMinGrad)) {  // :93-97
      result = 5;
      break;
    }
    q.x = x;
    q.g0 = grad;
    q.search = -q.minv * q.c - x;
    q.slope = q.search * grad;
    if (q.slope >= 0) {  // :150-153
      result = 2;
      break;
    }
    q.old_v = qp1_value(q, x);
    q.early = false;
    q.ls_failed = false;
    q.step = 1;
    line_search(q);
    if (q.ls_failed) {  // :121-125
      result = 2;
      break;
    }
    x = q.x1;
    val = q.v1;
  }
  x_out = x;
  free_out = free_;
  return result;
}
template <class real>
ILQR_HD void qp1_line_search_seq(QP1StateT<real>& q) {
  q.step = 1;
  q.x1 = qp1_trial(q, real(1));
  q.v1 = qp1_value(q, q.x1);
  qp1_backtrack_seq(q);
}

template <class real>
ILQR_HD int box_qp_scalar_fast(real Q, real c, real x0, real lo, real hi, real& x_out, int& free_out,
                               real& minv_out) {
  QP1StateT<real> q;
  qp1_begin(Q, c, x0, lo, hi, q);
  qp1_backtrack_seq(q);
  int result = qp1_finish(q, x_out, free_out, minv_out);
  if (result == kQpGoesOn) result = qp1_continue(q, [](QP1StateT<real>& s) { qp1_line_search_seq(s); }, x_out, free_out);
  return result;
}

// step sizes of the backtracking loop, exactly as it produces them: s[0] = 1, s[k+1] = s[k]*0.6
template <class real>
struct StepTableT {
  real s[104];
  constexpr StepTableT() : s() {
    real v = 1;
    for (int k = 0; k < 104; k++) {
      s[k] = v;
      v = v * real(kStepDec);
    }
  }
};

}  // namespace ilqr
