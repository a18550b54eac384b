// backward_wave.hpp -- generic backward pass, ONE WAVEFRONT PER TRAJECTORY, runtime nx <= 32,
// nu <= 16, every matrix of the step held in LDS.
//
// This is the path for (a) models that exist only as host virtuals -- their finite differences
// are taken on the host and uploaded with ilqr_set_derivatives -- and (b) the synthetic LQ
// configuration n = 32, m = 16 of BASELINE.json.  The n = 4 shipped models use the quad kernel
// (kernels.hpp), where a whole wavefront per trajectory would idle 60 of 64 lanes.
//
// Layout for these handles ("AoS"): everything of one trajectory is contiguous,
//     D  [b][t][REC]   record = fx | fu | cx | cxx | cxu | cu | cuu (column-major blocks, Rec order)
//     us [b][t][nu]    kff [b][t][nu]    Kfb [b][t][nu*nx]
// so a wavefront streams its own trajectory with 512-byte coalesced accesses (27 KB per step at
// n = 32, m = 16).
//
// Work split: the O(n^3) products are spread over the 64 lanes by output element with the inner
// sum in index order (the order of the reference's loops); LDS matrices use odd leading dimensions (33 / 17) so
// that lanes walking different columns hit different banks.  One trajectory per wavefront makes
// all control flow of the box-QP (src/boxqp.cpp:26-178) wave-uniform: projected-Newton
// iterations, the factor-on-count-change rule, the Armijo loop run exactly as written, with
// per-dimension work on lanes 0..m-1 and wave reductions for the scalars.
#pragma once
#include "common.hpp"

namespace ilqr {

constexpr int WN = 32, WM = 16;   // maximum dimensions of this kernel
constexpr int LDN = WN + 1;       // leading dimension of LDS matrices with up to 32 rows
constexpr int LDM = WM + 1;       // ... with up to 16 rows

struct WaveLds {
  double Vxx[LDN * WN], fx[LDN * WN], A1[LDN * WN], Qxx[LDN * WN];
  double fu[LDN * WM], T1[LDN * WM];
  double A2[LDM * WN], Qux[LDM * WN], K[LDM * WN];
  double Quu[LDM * WM], QuuF[LDM * WM], Qf[LDM * WM], R[LDM * WM], Ri[LDM * WM], Minv[LDM * WM];
  double Vx[WN], Qx[WN], Vxn[WN];
  double Qu[WM], x[WM], grad[WM], gc[WM], search[WM], lo[WM], hi[WM], clamped[WM], oldcl[WM], xr[WM], xc[WM], tmp[WM],
      kprev[WM], gfree[WM], xfree[WM];
  int vfree[WM], idx[WM];
};

__device__ __forceinline__ double wave_sum(double v) {  // sum over the 64 lanes (result in all lanes)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ void lds_sync() {  // one wavefront per block: LDS ops are in order, only drain them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
}

// 0.5 x'Qx + x.c with Q m x m (ld LDM), include/boxqp.h:53-55, evaluated ((0.5 x')Q) x + x.c
__device__ __forceinline__ double w_quad_cost(int m, const double* Q, const double* c, const double* x, int lane) {
  double part = 0, lin = 0;
  if (lane < m) {
    double r = 0;
    for (int i = 0; i < m; i++) r += (0.5 * x[i]) * Q[i + LDM * lane];
    part = r * x[lane];
    lin = x[lane] * c[lane];
  }
  return wave_sum(part) + wave_sum(lin);
}

// src/boxqp.cpp:26-139 for one trajectory per wavefront.  Inputs in LDS: QuuF (Q), Qu (c), kprev
// (x0), lo, hi.  Outputs: L.x (solution), L.vfree, L.R (compact upper factor, ld LDM), nfR.
__device__ int w_box_qp(int m, WaveLds& L, int lane, int& nfR_out) {
  const double* Q = L.QuuF;
  const double* c = L.Qu;
  // :35 clamp
  if (lane < m) {
    const double a = (L.kprev[lane] < L.lo[lane]) ? L.lo[lane] : L.kprev[lane];
    L.x[lane] = (L.hi[lane] < a) ? L.hi[lane] : a;
    L.clamped[lane] = 0;
    L.oldcl[lane] = 0;
    L.vfree[lane] = 0;
  }
  lds_sync();
  // :36 val = x'Qx + x.c (no 1/2)
  double val;
  {
    double part = 0, lin = 0;
    if (lane < m) {
      double r = 0;
      for (int i = 0; i < m; i++) r += L.x[i] * Q[i + LDM * lane];
      part = r * L.x[lane];
      lin = L.x[lane] * c[lane];
    }
    val = wave_sum(part) + wave_sum(lin);
  }
  double oldvalue = 0;
  int result = 0, nfR = 0;
  for (int iter = 0; iter <= kQpMaxIter; iter++) {
    if (iter > 0 && (oldvalue - val) < kMinRelImprove * fabs(oldvalue)) {  // :54-57
      result = 4;
      break;
    }
    // :58 grad = Qx + c ; :62-71 clamped set
    int cl = 1;
    double dd = 0;
    if (lane < m) {
      double s = 0;
      for (int j = 0; j < m; j++) s += Q[lane + LDM * j] * L.x[j];
      const double g = s + c[lane];
      L.grad[lane] = g;
      L.oldcl[lane] = L.clamped[lane];
      const bool isc = (fabs(L.x[lane] - L.lo[lane]) < kClampTol && g > 0) || (fabs(L.x[lane] - L.hi[lane]) < kClampTol && g < 0);
      L.clamped[lane] = isc ? 1.0 : 0.0;
      L.vfree[lane] = isc ? 0 : 1;
      cl = isc ? 1 : 0;
      dd = L.oldcl[lane] - L.clamped[lane];
    }
    oldvalue = val;
    const unsigned long long free_mask = __ballot(lane < m && !cl);
    const int nf = __popcll(free_mask);
    const double dsum = wave_sum(dd);
    if (nf == 0) {  // :74-77
      result = 6;
      break;
    }
    // ascending list of free dims (order-preserving compaction, eigen_helpers.h:15-61)
    if (lane < m && !cl) L.idx[__popcll(free_mask & ((1ull << lane) - 1ull))] = lane;
    lds_sync();
    if (iter == 0 || dsum != 0) {  // :80
      // Qfree = Q[free, free]
      for (int e = lane; e < nf * nf; e += 64) {
        const int a = e % nf, b2 = e / nf;
        L.Qf[a + LDM * b2] = Q[L.idx[a] + LDM * L.idx[b2]];
      }
      lds_sync();
      // Eigen llt_inplace<Lower>::unblocked (Cholesky/LLT.h:302-325); stops at a non-positive pivot
      for (int k = 0; k < nf; k++) {
        double xk = L.Qf[k + LDM * k];
        if (k > 0) {
          double sq = 0;
          for (int j = 0; j < k; j++) sq += L.Qf[k + LDM * j] * L.Qf[k + LDM * j];
          xk -= sq;
        }
        if (xk <= 0.0) break;
        xk = sqrt(xk);
        lds_sync();
        if (lane == 0) L.Qf[k + LDM * k] = xk;
        const int i = k + 1 + lane;
        if (i < nf) {
          double v = L.Qf[i + LDM * k];
          if (k > 0) {
            double s = 0;
            for (int j = 0; j < k; j++) s += L.Qf[i + LDM * j] * L.Qf[k + LDM * j];
            v -= s;
          }
          L.Qf[i + LDM * k] = v / xk;
        }
        lds_sync();
      }
      // :86-88 R = L' (dense upper, zeros below)
      for (int e = lane; e < nf * nf; e += 64) {
        const int a = e % nf, b2 = e / nf;
        L.R[a + LDM * b2] = (a <= b2) ? L.Qf[b2 + LDM * a] : 0.0;
      }
      nfR = nf;
      lds_sync();
    }
    // :93-97
    {
      const double gn2 = wave_sum((lane < m && !cl) ? L.grad[lane] * L.grad[lane] : 0.0);
      if (sqrt(gn2) < kMinGrad) {
        result = 5;
        break;
      }
    }
    // :100 grad_clamped = Q (x .* clamped) + c
    if (lane < m) L.tmp[lane] = L.x[lane] * L.clamped[lane];
    lds_sync();
    if (lane < m) {
      double s = 0;
      for (int j = 0; j < m; j++) s += Q[lane + LDM * j] * L.tmp[j];
      L.gc[lane] = s + c[lane];
    }
    lds_sync();
    // Ri = R^-1 (upper triangular, column j on lane j), Minv = Ri Ri'
    if (lane < nfR) {
      const int j = lane;
      for (int i = 0; i < nfR; i++) L.Ri[i + LDM * j] = 0;
      L.Ri[j + LDM * j] = 1.0 / L.R[j + LDM * j];
      for (int i = j - 1; i >= 0; i--) {
        double s = 0;
        for (int l2 = i + 1; l2 <= j; l2++) s += L.R[i + LDM * l2] * L.Ri[l2 + LDM * j];
        L.Ri[i + LDM * j] = -s / L.R[i + LDM * i];
      }
    }
    lds_sync();
    for (int e = lane; e < nfR * nfR; e += 64) {
      const int a = e % nfR, b2 = e / nfR;
      double s = 0;
      for (int l2 = 0; l2 < nfR; l2++) s += L.Ri[a + LDM * l2] * L.Ri[b2 + LDM * l2];
      L.Minv[a + LDM * b2] = s;
    }
    if (lane < m) {
      L.search[lane] = 0;
      if (!cl) {
        const int r = __popcll(free_mask & ((1ull << lane) - 1ull));
        L.gfree[r] = L.gc[lane];
        L.xfree[r] = L.x[lane];
      }
    }
    lds_sync();
    // :103-119 search(free) = -(R^-1 R^-T) gc(free) - x(free)   (a stale factor of equal size is used as is)
    if (lane < nfR && lane < nf) {
      double s = 0;
      for (int l2 = 0; l2 < nfR && l2 < nf; l2++) s += -L.Minv[lane + LDM * l2] * L.gfree[l2];
      L.search[L.idx[lane]] = s - L.xfree[lane];
    }
    lds_sync();
    // :121 quadclamp_line_search (src/boxqp.cpp:143-178)
    bool failed = false;
    double v = 0;
    {
      double sl = 0;
      if (lane < m) {
        double s = 0;
        for (int j = 0; j < m; j++) s += Q[lane + LDM * j] * L.x[j];
        sl = L.search[lane] * (s + c[lane]);
      }
      const double slope = wave_sum(sl);
      if (slope >= 0) {
        failed = true;
      } else {
        double step = 1;
        if (lane < m) {
          const double xr = L.x[lane] + step * L.search[lane];
          const double a = (xr < L.lo[lane]) ? L.lo[lane] : xr;
          L.xc[lane] = (L.hi[lane] < a) ? L.hi[lane] : a;
        }
        lds_sync();
        v = w_quad_cost(m, Q, c, L.xc, lane);
        const double old_v = w_quad_cost(m, Q, c, L.x, lane);
        while ((v - old_v) / (step * slope) < kArmijo) {
          step *= kStepDec;
          lds_sync();
          if (lane < m) {
            const double xr = L.x[lane] + step * L.search[lane];
            const double a = (xr < L.lo[lane]) ? L.lo[lane] : xr;
            L.xc[lane] = (L.hi[lane] < a) ? L.hi[lane] : a;
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
    if (lane < m) L.x[lane] = L.xc[lane];  // :133-134
    val = v;
    lds_sync();
  }
  lds_sync();
  nfR_out = nfR;
  return result;
}

// One wavefront per trajectory.  mode as in the quad kernel (0: one pass, all trajectories;
// 1: STEP 2 with the lambda retry and the gradient-norm test for running trajectories).
__global__ __launch_bounds__(64) void k_backward_w(BatchView v, int n, int m, const double* __restrict__ u_min,
                                                   const double* __restrict__ u_max, SolverParams sp, int mode) {
  __shared__ WaveLds L;
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  if (b >= v.B) return;
  if (mode == 1 && v.status[b] != 0) return;
  const int T = v.T;
  const int REC = 2 * n * n + 2 * n * m + n + m + m * m;
  const int oFX = 0, oFU = oFX + n * n, oCX = oFU + n * m, oCXX = oCX + n, oCXU = oCXX + n * n, oCU = oCXU + n * m,
            oCUU = oCU + m;
  const double* __restrict__ Db = v.D + (size_t)b * (T + 1) * REC;
  const double* __restrict__ usb = v.us + (size_t)b * T * m;
  double* __restrict__ kb = v.kff + (size_t)b * T * m;
  double* __restrict__ Kb = v.Kfb + (size_t)b * T * m * n;
  double lambda = v.lambda[b], dlambda = v.dlambda[b];

  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0;
  while (true) {
    // :353-354
    {
      const double* r = Db + (size_t)T * REC;
      for (int e = lane; e < n; e += 64) L.Vx[e] = r[oCX + e];
      for (int e = lane; e < n * n; e += 64) L.Vxx[(e % n) + LDN * (e / n)] = r[oCXX + e];
      if (lane < m) L.kprev[lane] = kb[(size_t)(T - 1) * m + lane];
    }
    dV0 = dV1 = 0;
    diverge = 0;
    lds_sync();
    for (int i = T - 1; i >= 0; i--) {
      const double* r = Db + (size_t)i * REC;
      for (int e = lane; e < n * n; e += 64) L.fx[(e % n) + LDN * (e / n)] = r[oFX + e];
      for (int e = lane; e < n * m; e += 64) L.fu[(e % n) + LDN * (e / n)] = r[oFU + e];
      if (lane < m) {
        const double us = usb[(size_t)i * m + lane];
        L.lo[lane] = u_min[lane] - us;  // :369
        L.hi[lane] = u_max[lane] - us;
      }
      lds_sync();
      // :359-360
      for (int a = lane; a < n; a += 64) {
        double acc = 0;
        for (int q = 0; q < n; q++) acc += L.fx[q + LDN * a] * L.Vx[q];
        L.Qx[a] = r[oCX + a] + acc;
      }
      if (lane < m) {
        double acc = 0;
        for (int q = 0; q < n; q++) acc += L.fu[q + LDN * lane] * L.Vx[q];
        L.Qu[lane] = r[oCU + lane] + acc;
      }
      // A1 = fx' Vxx ; A2 = fu' Vxx
      for (int e = lane; e < n * n; e += 64) {
        const int a = e % n, c = e / n;
        double acc = 0;
        for (int q = 0; q < n; q++) acc += L.fx[q + LDN * a] * L.Vxx[q + LDN * c];
        L.A1[a + LDN * c] = acc;
      }
      for (int e = lane; e < m * n; e += 64) {
        const int a = e % m, c = e / m;
        double acc = 0;
        for (int q = 0; q < n; q++) acc += L.fu[q + LDN * a] * L.Vxx[q + LDN * c];
        L.A2[a + LDM * c] = acc;
      }
      lds_sync();
      // :361 Qxx ; :362 Qux ; :363/:367 Quu, QuuF
      for (int e = lane; e < n * n; e += 64) {
        const int a = e % n, c = e / n;
        double acc = 0;
        for (int q = 0; q < n; q++) acc += L.A1[a + LDN * q] * L.fx[q + LDN * c];
        L.Qxx[a + LDN * c] = r[oCXX + e] + acc;
      }
      for (int e = lane; e < m * n; e += 64) {
        const int a = e % m, c = e / m;
        double acc = 0;
        for (int q = 0; q < n; q++) acc += L.A2[a + LDM * q] * L.fx[q + LDN * c];
        L.Qux[a + LDM * c] = r[oCXU + c + n * a] + acc;
      }
      for (int e = lane; e < m * m; e += 64) {
        const int a = e % m, c = e / m;
        double acc = 0;
        for (int q = 0; q < n; q++) acc += L.A2[a + LDM * q] * L.fu[q + LDN * c];
        const double cuu = r[oCUU + e];
        L.Quu[a + LDM * c] = cuu + acc;
        L.QuuF[a + LDM * c] = (cuu + ((a == c) ? lambda : 0.0)) + acc;
      }
      lds_sync();
      int nfR = 0;
      const int result = w_box_qp(m, L, lane, nfR);
      if (result < 1) {  // :371
        diverge = i;
        break;
      }
      // :373-385  K rows of free dims
      for (int e = lane; e < m * n; e += 64) L.K[(e % m) + LDM * (e / m)] = 0;
      const unsigned long long free_mask = __ballot(lane < m && L.vfree[lane]);
      const int nf = __popcll(free_mask);
      if (lane < m && L.vfree[lane]) L.idx[__popcll(free_mask & ((1ull << lane) - 1ull))] = lane;
      lds_sync();
      if (nf > 0) {
        if (lane < nfR) {
          const int j = lane;
          for (int i2 = 0; i2 < nfR; i2++) L.Ri[i2 + LDM * j] = 0;
          L.Ri[j + LDM * j] = 1.0 / L.R[j + LDM * j];
          for (int i2 = j - 1; i2 >= 0; i2--) {
            double s = 0;
            for (int l2 = i2 + 1; l2 <= j; l2++) s += L.R[i2 + LDM * l2] * L.Ri[l2 + LDM * j];
            L.Ri[i2 + LDM * j] = -s / L.R[i2 + LDM * i2];
          }
        }
        lds_sync();
        for (int e = lane; e < nfR * nfR; e += 64) {
          const int a = e % nfR, b2 = e / nfR;
          double s = 0;
          for (int l2 = 0; l2 < nfR; l2++) s += L.Ri[a + LDM * l2] * L.Ri[b2 + LDM * l2];
          L.Minv[a + LDM * b2] = s;
        }
        lds_sync();
        const int nuse = (nf < nfR) ? nf : nfR;
        for (int e = lane; e < nuse * n; e += 64) {
          const int rr = e % nuse, c = e / nuse;
          double acc = 0;
          for (int l2 = 0; l2 < nuse; l2++) acc += -L.Minv[rr + LDM * l2] * L.Qux[L.idx[l2] + LDM * c];
          L.K[L.idx[rr] + LDM * c] = acc;
        }
      }
      lds_sync();
      // :388-389
      {
        const double d0 = wave_sum(lane < m ? L.x[lane] * L.Qu[lane] : 0.0);
        double part = 0;
        if (lane < m) {
          double rr = 0;
          for (int a = 0; a < m; a++) rr += (0.5 * L.x[a]) * L.Quu[a + LDM * lane];
          part = rr * L.x[lane];
        }
        dV0 += d0;
        dV1 += wave_sum(part);
      }
      // T1 = K' Quu (n x m)
      for (int e = lane; e < n * m; e += 64) {
        const int a = e % n, c = e / n;
        double acc = 0;
        for (int q = 0; q < m; q++) acc += L.K[q + LDM * a] * L.Quu[q + LDM * c];
        L.T1[a + LDN * c] = acc;
      }
      lds_sync();
      // :391 Vx ; :392 Vn into A1 ; :393 symmetrise into Vxx
      for (int a = lane; a < n; a += 64) {
        double t1 = 0, t2 = 0, t3 = 0;
        for (int c = 0; c < m; c++) t1 += L.T1[a + LDN * c] * L.x[c];
        for (int c = 0; c < m; c++) t2 += L.K[c + LDM * a] * L.Qu[c];
        for (int c = 0; c < m; c++) t3 += L.Qux[c + LDM * a] * L.x[c];
        L.Vxn[a] = ((L.Qx[a] + t1) + t2) + t3;
      }
      for (int e = lane; e < n * n; e += 64) {
        const int a = e % n, c = e / n;
        double t1 = 0, t2 = 0, t3 = 0;
        for (int q = 0; q < m; q++) t1 += L.T1[a + LDN * q] * L.K[q + LDM * c];
        for (int q = 0; q < m; q++) t2 += L.K[q + LDM * a] * L.Qux[q + LDM * c];
        for (int q = 0; q < m; q++) t3 += L.Qux[q + LDM * a] * L.K[q + LDM * c];
        L.A1[a + LDN * c] = ((L.Qxx[a + LDN * c] + t1) + t2) + t3;
      }
      lds_sync();
      for (int e = lane; e < n * n; e += 64) {
        const int a = e % n, c = e / n;
        L.Vxx[a + LDN * c] = 0.5 * (L.A1[a + LDN * c] + L.A1[c + LDN * a]);
      }
      for (int a = lane; a < n; a += 64) L.Vx[a] = L.Vxn[a];
      // :396-397
      if (lane < m) {
        kb[(size_t)i * m + lane] = L.x[lane];
        L.kprev[lane] = L.x[lane];
      }
      for (int e = lane; e < m * n; e += 64) Kb[(size_t)i * m * n + e] = L.K[(e % m) + LDM * (e / m)];
      lds_sync();
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

// canonical [B][S][len] block  <->  AoS record slot [b][s][off .. off+len)
__global__ void k_rec_aos(double* __restrict__ D, double* __restrict__ host_layout, int B, int S, int REC, int off, int len,
                          int to_record) {
  const size_t nel = (size_t)B * S * len;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nel; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % len);
    const size_t bs = i / len;
    if (to_record)
      D[bs * REC + off + e] = host_layout[i];
    else
      host_layout[i] = D[bs * REC + off + e];
  }
}

}  // namespace ilqr
