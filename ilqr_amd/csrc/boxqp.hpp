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

template <int M>
ILQR_HD void clamp_to_limits(const double* x, const double* lo, const double* hi, double* out) {
#pragma unroll
  for (int i = 0; i < M; i++) {  // include/boxqp.h:48-51  upper.cwiseMin(x.cwiseMax(lower))
    const double a = (x[i] < lo[i]) ? lo[i] : x[i];
    out[i] = (hi[i] < a) ? hi[i] : a;
  }
}

template <int M>
ILQR_HD double quad_cost(const double* Q, const double* c, const double* x) {
  double quad = 0, lin = 0;  // include/boxqp.h:53-55   ((0.5 x')Q) x + x.c
#pragma unroll
  for (int j = 0; j < M; j++) {
    double r = 0;
#pragma unroll
    for (int i = 0; i < M; i++) r += (0.5 * x[i]) * Q[i + M * j];
    quad += r * x[j];
  }
#pragma unroll
  for (int i = 0; i < M; i++) lin += x[i] * c[i];
  return quad + lin;
}

template <int M>
ILQR_HD void matvec(const double* Q, const double* x, double* y) {
#pragma unroll
  for (int i = 0; i < M; i++) {
    double s = 0;
#pragma unroll
    for (int j = 0; j < M; j++) s += Q[i + M * j] * x[j];
    y[i] = s;
  }
}

// src/boxqp.cpp:143-178.  Returns failed; x_opt/v_opt are written unless the direction is not
// a descent direction (:151-154).
template <int M>
ILQR_HD bool quadclamp_line_search(const double* x0, const double* dir, const double* Q,
                                                      const double* c, const double* lo, const double* hi,
                                                      double* x_opt, double& v_opt) {
  double grad[M], xr[M], xc[M];
  matvec<M>(Q, x0, grad);
  double slope = 0;
#pragma unroll
  for (int i = 0; i < M; i++) slope += dir[i] * (grad[i] + c[i]);
  if (slope >= 0) return true;
  double step = 1;
#pragma unroll
  for (int i = 0; i < M; i++) xr[i] = x0[i] + step * dir[i];
  clamp_to_limits<M>(xr, lo, hi, xc);
  double v = quad_cost<M>(Q, c, xc);
  const double old_v = quad_cost<M>(Q, c, x0);
  bool failed = false;
  while ((v - old_v) / (step * slope) < kArmijo) {
    step *= kStepDec;
#pragma unroll
    for (int i = 0; i < M; i++) xr[i] = x0[i] + step * dir[i];
    clamp_to_limits<M>(xr, lo, hi, xc);
    v = quad_cost<M>(Q, c, xc);
    if (step < kMinStep) {
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
template <int M>
ILQR_HD void llt_lower(int nf, double* A) {
  bool stop = false;
#pragma unroll
  for (int k = 0; k < M; k++) {
    if (k < nf && !stop) {
      double x = A[k + M * k];
      double sq = 0;
#pragma unroll
      for (int j = 0; j < M; j++)
        if (j < k) sq += A[k + M * j] * A[k + M * j];
      if (k > 0) x -= sq;
      if (x <= 0.0) {
        stop = true;
      } else {
        x = sqrt(x);
        A[k + M * k] = x;
#pragma unroll
        for (int i = 0; i < M; i++)
          if (i > k && i < nf) {
            double s = 0;
#pragma unroll
            for (int j = 0; j < M; j++)
              if (j < k) s += A[i + M * j] * A[k + M * j];
            double v = A[i + M * k];
            if (k > 0) v -= s;
            A[i + M * k] = v / x;
          }
      }
    }
  }
}

// Minv = R^-1 R^-T for the upper-triangular leading nf x nf block of R (ld = M).
// (The reference: two PartialPivLU inverses and a product, boxqp.cpp:105-112, ilqr_core.cpp:379.)
template <int M>
ILQR_HD void rinv_rinvT(int nf, const double* R, double* Minv) {
  double Ri[M * M];
#pragma unroll
  for (int e = 0; e < M * M; e++) Ri[e] = 0;
#pragma unroll
  for (int j = 0; j < M; j++) {
    if (j < nf) {
      Ri[j + M * j] = 1.0 / R[j + M * j];
#pragma unroll
      for (int ii = 0; ii < M; ii++) {
        const int i = j - 1 - ii;  // i = j-1 .. 0
        if (i >= 0) {
          double s = 0;
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
      double s = 0;
#pragma unroll
      for (int l = 0; l < M; l++)
        if (l < nf) s += Ri[i + M * l] * Ri[j + M * l];
      Minv[i + M * j] = s;
    }
}

template <int M>
struct BoxQPResult {
  int result;
  double x[M];
  int v_free[M];  // 0/1 mask of free dims at exit
  double R[M * M]; // compact upper factor: leading nfR x nfR block, ld = M
  int nfR;
};

// src/boxqp.cpp:26-139
template <int M>
ILQR_HD void box_qp(const double* Q, const double* c, const double* x0, const double* lo,
                                       const double* hi, BoxQPResult<M>& res) {
  double x[M], grad[M], gc[M], search[M], tmp[M];
  double clamped[M], old_clamped[M];
  clamp_to_limits<M>(x0, lo, hi, x);  // :35
  double val;
  {  // :36  x'Qx + x.c (no 1/2)
    double quad = 0, lin = 0;
#pragma unroll
    for (int j = 0; j < M; j++) {
      double r = 0;
#pragma unroll
      for (int i = 0; i < M; i++) r += x[i] * Q[i + M * j];
      quad += r * x[j];
    }
#pragma unroll
    for (int i = 0; i < M; i++) lin += x[i] * c[i];
    val = quad + lin;
  }
  double oldvalue = 0;
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

  for (int iter = 0; iter <= kQpMaxIter; iter++) {  // :50
    if (iter > 0 && (oldvalue - val) < kMinRelImprove * fabs(oldvalue)) {  // :54-57
      result = 4;
      break;
    }
    matvec<M>(Q, x, grad);
#pragma unroll
    for (int i = 0; i < M; i++) grad[i] += c[i];
    oldvalue = val;

    bool all_clamped = true;
    double dsum = 0;
    int rank[M];
    int nf = 0;
#pragma unroll
    for (int i = 0; i < M; i++) {  // :62-71
      old_clamped[i] = clamped[i];
      const bool cl = (fabs(x[i] - lo[i]) < kClampTol && grad[i] > 0) || (fabs(x[i] - hi[i]) < kClampTol && grad[i] < 0);
      clamped[i] = cl ? 1.0 : 0.0;
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
      double Qf[M * M];
#pragma unroll
      for (int e = 0; e < M * M; e++) Qf[e] = 0;
      // extract_bool_rowsandcols (eigen_helpers.h:46-61): Qf[rank[i]][rank[j]] = Q[i][j] for free i,j
#pragma unroll
      for (int a = 0; a < M; a++)
#pragma unroll
        for (int b = 0; b < M; b++) {
          double v = 0;
#pragma unroll
          for (int i = 0; i < M; i++)
#pragma unroll
            for (int j = 0; j < M; j++)
              if (res.v_free[i] && res.v_free[j] && rank[i] == a && rank[j] == b) v = Q[i + M * j];
          Qf[a + M * b] = v;
        }
      llt_lower<M>(nf, Qf);  // :85 (info() ignored)
#pragma unroll
      for (int a = 0; a < M; a++)
#pragma unroll
        for (int b = 0; b < M; b++) res.R[a + M * b] = (a <= b && b < nf) ? Qf[b + M * a] : 0.0;  // :86-88
      nfR = nf;
    }

    double gn2 = 0;  // :93-97
#pragma unroll
    for (int i = 0; i < M; i++)
      if (res.v_free[i]) gn2 += grad[i] * grad[i];
    if (sqrt(gn2) < kMinGrad) {
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
    double Minv[M * M], gfree[M], xfree[M], sfree[M];
    rinv_rinvT<M>(nfR, res.R, Minv);
#pragma unroll
    for (int a = 0; a < M; a++) {
      double g = 0, xx = 0;
#pragma unroll
      for (int i = 0; i < M; i++)
        if (res.v_free[i] && rank[i] == a) {
          g = gc[i];
          xx = x[i];
        }
      gfree[a] = g;
      xfree[a] = xx;
    }
#pragma unroll
    for (int a = 0; a < M; a++) {
      double s = 0;
#pragma unroll
      for (int l = 0; l < M; l++)
        if (l < nfR) s += -Minv[a + M * l] * gfree[l];
      sfree[a] = s - xfree[a];
    }
#pragma unroll
    for (int i = 0; i < M; i++) {
      double s = 0;
#pragma unroll
      for (int a = 0; a < M; a++)
        if (res.v_free[i] && rank[i] == a) s = sfree[a];
      search[i] = s;
    }

    double lx[M], lv = 0;
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

}  // namespace ilqr
