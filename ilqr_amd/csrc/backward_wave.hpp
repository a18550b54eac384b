// backward_wave.hpp -- what the generic backward kernels (ONE WAVEFRONT PER TRAJECTORY, runtime nx <= 32, nu <= 16: backward_wave2.hpp,
// backward_wave3.hpp) share: the box-QP's LDS block, matrix-core tile helpers, wave reductions and the literal box-QP w_box_qp.
// (Round 1's kernel k_backward_w, which held every matrix of the step in LDS, was retired in ABI 5; the notes below on its layout
// still describe WaveLds, which the register kernels use in part.)
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
// Work split: the O(n^3) products run on the matrix cores, v_mfma_f64_16x16x4_f64 on 16x16 output
// tiles with operands read from LDS (one double per lane and k-step; fp64 MFMA has the VALU's
// flop rate on gfx950, what it buys is 1 LDS read per 32 flops instead of per flop and a k-ordered
// FMA chain = the order of the reference's loops).  LDS matrices are zero-padded to whole tiles
// and use odd leading dimensions (33 / 17) so that lanes walking different columns hit different
// banks.  One trajectory per wavefront makes
// all control flow of the box-QP (src/boxqp.cpp:26-178) wave-uniform: projected-Newton
// iterations, the factor-on-count-change rule, the Armijo loop run exactly as written, with
// per-dimension work on lanes 0..m-1 and wave reductions for the scalars.
#pragma once
#include <stddef.h>

#include "boxqp.hpp"
#include "common.hpp"

namespace ilqr {

constexpr int WN = 32, WM = 16;   // maximum dimensions of this kernel
constexpr int LDN = WN + 1;       // leading dimension of LDS matrices with up to 32 rows
constexpr int LDM = WM + 1;       // ... with up to 16 rows

// Matrices share storage with ones that are dead by the time they are written, which brings a
// wavefront's LDS from 68 KB to 39.6 KB, i.e. from two to FOUR wavefronts per CU (160 KB; one per
// SIMD, which is also what the 512 registers of a wavefront allow):
//   Qxx  in Vxx : Vxx' is last read for A1 = fx'Vxx' and A2 = fu'Vxx'; Qxx is written after both and
//                 read once, when Vn is assembled into A1; the new Vxx then overwrites it
//   Quu, QuuF, Minv in fx : fx is last read for Qxx and Qux; the three m x m matrices are written
//                 after that (Quu/QuuF in the same phase, Minv by the box-QP) and are dead at the
//                 end of the step, when the next record's fx is copied in
//   Ri / Qf in fu : fu is last read for Quu.  Ri is written and read (for Minv) inside a
//                 factorisation; Qf is the scatter buffer of the K product after the box-QP
//   T1   in fu  : T1 = K'Quu is written after K (i.e. after Qf's last read); the next step refills
//                 fu completely
//   K    in A2  : A2 is last read for Qux / Quu; K is written after the box-QP (A2's padding
//                 columns are exact zeros, as K's must be)
struct WaveLds {
  double Vxx[LDN * WN], fx[LDN * WN], A1[LDN * WN];
  double fu[LDN * WM];
  double A2[LDM * WN], Qux[LDM * WN];
  __device__ __forceinline__ double* Qxx() { return Vxx; }
  __device__ __forceinline__ double* T1() { return fu; }
  __device__ __forceinline__ double* K() { return A2; }
  __device__ __forceinline__ double* Quu() { return fx; }
  __device__ __forceinline__ double* QuuF() { return fx + LDM * WM; }
  __device__ __forceinline__ double* Minv() { return fx + 2 * LDM * WM; }
  __device__ __forceinline__ double* Qf() { return fu; }
  __device__ __forceinline__ double* Ri() { return fu; }
  double Vx[WN], Qx[WN], Vxn[WN];
  double Qu[WM], x[WM], grad[WM], gc[WM], search[WM], lo[WM], hi[WM], clamped[WM], xc[WM], tmp[WM],
      kprev[WM], gfree[WM], xfree[WM];
  int vfree[WM], idx[WM];
};

typedef double double4_t __attribute__((ext_vector_type(4)));

// Experiment builds (-DILQR_W2_TIMING, scripts/w2_sections.sh): shader cycles per section of the generic backward step and of its
// box-QP, summed by the first wavefront of the grid and printed by ilqr_destroy.  Product builds compile the marks to nothing.
#ifdef ILQR_W2_TIMING
__device__ long long g_w2_cycles[8];   // step sections 0..7
__device__ long long g_q_cycles[8];    // box-QP sections 0..5
__device__ long long g_q_counts[4];    // QPs, projected-Newton iterations, factorisations, Armijo trips beyond the first
struct W2Clock {
  long long t, step[8], qp[8], cnt[4];
  __device__ void start() {
    for (int i = 0; i < 8; i++) step[i] = qp[i] = 0;
    for (int i = 0; i < 4; i++) cnt[i] = 0;
    t = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void mark(long long* acc, int k) {
    __builtin_amdgcn_sched_barrier(0);
    const long long n = __builtin_amdgcn_s_memtime();
    acc[k] += n - t;
    t = n;
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ void flush() {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      for (int i = 0; i < 8; i++) {
        atomicAdd((unsigned long long*)&g_w2_cycles[i], (unsigned long long)step[i]);
        atomicAdd((unsigned long long*)&g_q_cycles[i], (unsigned long long)qp[i]);
      }
      for (int i = 0; i < 4; i++) atomicAdd((unsigned long long*)&g_q_counts[i], (unsigned long long)cnt[i]);
    }
  }
};
#define ILQR_W2CLOCK_ARG , W2Clock& clk
#define ILQR_W2CLOCK_PASS , clk
#define ILQR_W2MARK(k) clk.mark(clk.step, k);
#define ILQR_QMARK(k) clk.mark(clk.qp, k);
#define ILQR_QCOUNT(k) clk.cnt[k] += 1;
#else
#define ILQR_W2CLOCK_ARG
#define ILQR_W2CLOCK_PASS
#define ILQR_W2MARK(k)
#define ILQR_QMARK(k)
#define ILQR_QCOUNT(k)
#endif

// One 16x16 output tile: acc += sum_k A(i,k) B(k,j) over `ksteps` k-steps of 4.  Lane l feeds
// A(i = l&15, k = 4 ks + (l>>4)) and B(k, j = l&15); it ends with D(row = (l>>4) + 4 r, col = l&15),
// r = 0..3 (the f64 C/D map of v_mfma_f64_16x16x4_f64).
template <int KSTEPS, class FA, class FB>
__device__ __forceinline__ double4_t mfma_tile(FA a_at, FB b_at, int lane) {
  // compile-time trip count (operands are zero-padded to whole tiles): all 2*KSTEPS LDS reads are
  // issued before the first MFMA, so the chain of dependent MFMAs runs back to back
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int ij = lane & 15, kq = lane >> 4;
  double av[KSTEPS], bv[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ks++) {
    av[ks] = a_at(ij, ks * 4 + kq);
    bv[ks] = b_at(ks * 4 + kq, ij);
  }
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ks++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], bv[ks], acc, 0, 0, 0);
  return acc;
}

// this lane's share of one 16-wide operand block, f(i or j = l&15, k = 4 ks + (l>>4)), ks = 0..KSTEPS-1
template <int KSTEPS, class F>
__device__ __forceinline__ void ld_operand(F f, int lane, double (&o)[KSTEPS]) {
  const int ij = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ks++) o[ks] = f(ij, ks * 4 + kq);
}

// sum over the 64 lanes (result in all lanes).  DPP inside each row of 16 lanes (xor 1, xor 2, half
// mirror, mirror: every lane of a row ends with the row total), then the four row totals through
// v_readlane.  The box-QP calls this a few times per Armijo trip; the ds_bpermute butterfly it
// replaces (6 dependent LDS round trips) was most of the kernel's box-QP time.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  auto row = [&](int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
  };
  return (row(0) + row(16)) + (row(32) + row(48));
}
// wave_sum of a value that is zero outside lanes 0..15 (everything indexed by a control dimension,
// m <= WM = 16): rows 1..3 total +0.0, so only row 0 is fetched; same value as wave_sum, bit for bit
__device__ __forceinline__ double wave_sum_row0(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  const double r0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0), __builtin_amdgcn_readlane(__double2loint(v), 0));
  return (r0 + 0.0) + 0.0;
}
__device__ __forceinline__ void lds_sync() {
  // One wavefront per block: the LDS pipeline executes a wavefront's DS instructions in issue
  // order, so a ds_read after a ds_write sees it whichever lane wrote.  Only the COMPILER has to be
  // kept from reordering LDS accesses across this point: a wavefront-scope fence (no hardware
  // wait).  A workgroup-scope release fence here would drain vmcnt -- i.e. every outstanding
  // global load/store -- at each of the ~40 sync points of a step.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// sum_{j in [lo, hi)} a(j) b(j), j ascending, hi <= WM.  Written as WM unrolled, masked terms: a loop
// with a runtime trip count reads LDS, waits, multiplies, one term at a time (~120 cycles each with
// a single wavefront per SIMD); unrolled, the 2 x 16 reads go out back to back.  A masked-out term
// contributes fma(0, 0, s) = s, so the value is that of the plain loop.
template <int N = WM, class FA, class FB>
__device__ __forceinline__ double dot_masked(int lo, int hi, FA a, FB b) {
  double av[N], bv[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    av[j] = a(j);
    bv[j] = b(j);
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    const bool in = (j >= lo) & (j < hi);
    s = __builtin_fma(in ? av[j] : 0.0, in ? bv[j] : 0.0, s);
  }
  return s;
}

// The same sum over all N terms without the masks, for operands that are ZERO-PADDED beyond the range (both of
// them: 0 * 0 adds nothing to the chain, so the value is dot_masked's bit for bit).  The box-QP's vectors (lanes
// >= m are never written after the kernel's clear) and its m x m matrices (every 16 x 16 entry written each step,
// zeros outside m x m) are; the masks were 4 of the 5 instructions of a term, a quarter of the kernel's VALU work.
template <int N = WM, class FA, class FB>
__device__ __forceinline__ double dot_padded(FA a, FB b) {
  double av[N], bv[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    av[j] = a(j);
    bv[j] = b(j);
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < N; j++) s = __builtin_fma(av[j], bv[j], s);
  return s;
}

// 0.5 x'Qx + x.c with Q m x m (ld LDM), include/boxqp.h:53-55, evaluated ((0.5 x')Q) x + x.c
__device__ __forceinline__ double w_quad_cost(int m, const double* Q, const double* c, const double* x, int lane) {
  double part = 0, lin = 0;
  if (lane < m) {
    const double r = dot_padded([&](int i) { return 0.5 * x[i]; }, [&](int i) { return Q[i + LDM * lane]; });
    part = r * x[lane];
    lin = x[lane] * c[lane];
  }
  return wave_sum_row0(part) + wave_sum_row0(lin);
}

// src/boxqp.cpp:26-139 for one trajectory per wavefront.  Inputs in LDS: QuuF (Q), Qu (c), kprev
// (x0), lo, hi.  Outputs: L.x (solution), L.vfree, L.Minv (R^-1 R^-T of the last factor, ld LDM), nfR.
template <class LDS>
__device__ int w_box_qp(int m, LDS& L, int lane, int& nfR_out ILQR_W2CLOCK_ARG, int* nfact_out = nullptr, int fixes = 0) {
  ILQR_QCOUNT(0)
  const double* Q = L.QuuF();
  const double* c = L.Qu;
  // :35 clamp
  if (lane < m) {
    const double a = (L.kprev[lane] < L.lo[lane]) ? L.lo[lane] : L.kprev[lane];
    L.x[lane] = (L.hi[lane] < a) ? L.hi[lane] : a;
    L.clamped[lane] = 0;
    L.vfree[lane] = 0;
  }
  lds_sync();
  // :36 val = x'Qx + x.c (no 1/2)
  double val;
  {
    double part = 0, lin = 0;
    if (lane < m) {
      const double r = dot_padded([&](int i) { return L.x[i]; }, [&](int i) { return Q[i + LDM * lane]; });
      part = r * L.x[lane];
      lin = L.x[lane] * c[lane];
    }
    val = wave_sum_row0(part) + wave_sum_row0(lin);
  }
  double oldvalue = 0;
  int result = 0, nfR = 0;
  int nfact_last = 0;  // pivots the last factorisation completed (== nfR: the whole block was positive definite)
  for (int iter = 0; iter <= kQpMaxIter; iter++) {
    ILQR_QCOUNT(1)
    if (iter > 0 && (oldvalue - val) < kMinRelImprove * fabs(oldvalue)) {  // :54-57
      result = 4;
      break;
    }
    // :58 grad = Qx + c ; :62-71 clamped set
    int cl = 1;
    double dd = 0;
    if (lane < m) {
      const double s = dot_padded([&](int j) { return Q[lane + LDM * j]; }, [&](int j) { return L.x[j]; });
      const double g = s + c[lane];
      L.grad[lane] = g;
      const double oldcl = L.clamped[lane];  // (lane-local: first pass reads the 0 written above)
      const bool isc = (fabs(L.x[lane] - L.lo[lane]) < kClampTol && g > 0) || (fabs(L.x[lane] - L.hi[lane]) < kClampTol && g < 0);
      L.clamped[lane] = isc ? 1.0 : 0.0;
      L.vfree[lane] = isc ? 0 : 1;
      cl = isc ? 1 : 0;
      dd = oldcl - L.clamped[lane];
    }
    oldvalue = val;
    const unsigned long long free_mask = __ballot(lane < m && !cl);
    const int nf = __popcll(free_mask);
    const double dsum = wave_sum_row0(dd);
    if (nf == 0) {  // :74-77
      result = 6;
      break;
    }
    // ascending list of free dims (order-preserving compaction, eigen_helpers.h:15-61)
    if (lane < m && !cl) L.idx[__popcll(free_mask & ((1ull << lane) - 1ull))] = lane;
    lds_sync();
    ILQR_QMARK(0)
    if (iter == 0 || dsum != 0) {  // :80
      ILQR_QCOUNT(2)
      // Qfree = Q[free, free], row i on lane i, in registers: the factorisation and the inversion
      // below are chains of short dot products with a square root / division between them; through
      // LDS every link of the chain paid a write -> read round trip (~35 K cycles per QP at nf = 16),
      // here a row-k broadcast is a v_readlane and the loops have compile-time bounds.
      auto bcast = [&](double v, int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
      };
      double row[WM];
      {
        const int gi = L.idx[(lane < nf) ? lane : 0];
#pragma unroll
        for (int j = 0; j < WM; j++) row[j] = (lane < nf && j < nf) ? Q[gi + LDM * L.idx[(j < nf) ? j : 0]] : 0.0;
      }
      // Eigen llt_inplace<Lower>::unblocked (Cholesky/LLT.h:302-325); stops at a non-positive pivot
      // and leaves the rest of the lower triangle as it was.
      // No IEEE square root / division in the chain: a pivot's root comes with its reciprocal (rsqrt_and_sqrt, boxqp.hpp:
      // v_rsq_f64 + Newton, <= 1 ulp each), the column is scaled by that reciprocal, and the triangular inverse below reuses
      // it for its diagonal -- 16 roots + 48 divisions of ~12-20 instructions each were a third of the step's VALU work.
      double my_inv = 0;   // lane k: 1 / L(k, k) of a factored pivot
      int n_fact = 0;      // pivots factored before the stop (wave-uniform)
      {
        bool stopped = false;
#pragma unroll
        for (int k = 0; k < WM; k++) {
          if (k < nf && !stopped) {
            double rk[WM];
            double sq = 0;
#pragma unroll
            for (int j = 0; j < k; j++) {
              rk[j] = bcast(row[j], k);
              sq = __builtin_fma(rk[j], rk[j], sq);
            }
            double xk = bcast(row[k], k);
            if (k > 0) xk -= sq;
            if (xk <= 0.0) {
              stopped = true;
            } else {
              double rs;
              rsqrt_and_sqrt(xk, rs, xk);
              double s = 0;
#pragma unroll
              for (int j = 0; j < k; j++) s = __builtin_fma(row[j], rk[j], s);
              double v = row[k];
              if (k > 0) v -= s;
              v = v * rs;
              row[k] = (lane == k) ? xk : ((lane > k && lane < nf) ? v : row[k]);
              my_inv = (lane == k) ? rs : my_inv;
              n_fact = k + 1;
            }
          }
        }
      }
      nfR = nf;
      nfact_last = n_fact;
      if ((fixes & 2) && n_fact < nf) {  // opt-in (ILQR_FLAG_REFERENCE_FIXES): not positive definite on the free subspace ends the QP (boxqp.cpp:85-88 ignores info())
        result = -1;
        break;
      }
      ILQR_QMARK(1)
      // :86-88 R = L' (upper); Ri = R^-1 (upper triangular, column j on lane j), Minv = Ri Ri'
      // (:105-112).  The reference inverts R in every iteration; R only changes here, so the product
      // is computed here and kept (same values) -- for the iterations that reuse a stale factor and
      // for the caller's K.  R(i, l2) = L(l2, i) = lane l2's row[i].
      {
        double ri[WM];
#pragma unroll
        for (int i = 0; i < WM; i++) ri[i] = 0.0;
#pragma unroll
        for (int i = WM - 1; i >= 0; i--) {
          if (i < nfR) {
            // 1 / R(i, i): kept from the factorisation; a pivot the factorisation stopped before (Eigen's partial factor: its
            // "root" is the unmodified entry) gets a reciprocal of its own
            const double inv_rii = (i < n_fact) ? bcast(my_inv, i) : recip(bcast(row[i], i));
            double s = 0;
#pragma unroll
            for (int l2 = i + 1; l2 < WM; l2++) s = __builtin_fma(bcast(row[i], l2), ri[l2], s);
            const double off = -s * inv_rii;
            ri[i] = (lane >= nfR) ? 0.0 : ((lane == i) ? inv_rii : ((lane > i) ? off : 0.0));
          }
        }
        if (lane < WM) {
#pragma unroll
          for (int i = 0; i < WM; i++) L.Ri()[i + LDM * lane] = ri[i];
        }
      }
      lds_sync();
      {
        const double4_t acc = mfma_tile<WM / 4>([&](int i, int k) { return L.Ri()[i + LDM * k]; },
                                                [&](int k, int j) { return L.Ri()[j + LDM * k]; }, lane);
        const int col = lane & 15, r0 = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; r++) L.Minv()[(r0 + 4 * r) + LDM * col] = acc[r];
      }
      lds_sync();
      ILQR_QMARK(2)
    }
    // :93-97
    {
      const double gn2 = wave_sum_row0((lane < m && !cl) ? L.grad[lane] * L.grad[lane] : 0.0);
      if (grad_norm_below_min(gn2)) {  // sqrt(gn2) < minGrad, boxqp.hpp
        result = 5;
        break;
      }
    }
    // :100 grad_clamped = Q (x .* clamped) + c
    if (lane < m) L.tmp[lane] = L.x[lane] * L.clamped[lane];
    lds_sync();
    if (lane < m) {
      const double s = dot_padded([&](int j) { return Q[lane + LDM * j]; }, [&](int j) { return L.tmp[j]; });
      L.gc[lane] = s + c[lane];
    }
    lds_sync();
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
      const double s = dot_masked(0, (nfR < nf) ? nfR : nf, [&](int l2) { return -L.Minv()[lane + LDM * l2]; }, [&](int l2) { return L.gfree[l2]; });
      L.search[L.idx[lane]] = s - L.xfree[lane];
    }
    lds_sync();
    ILQR_QMARK(3)
    // :121 quadclamp_line_search (src/boxqp.cpp:143-178)
    bool failed = false;
    double v = 0;
    {
      double sl = 0;
      if (lane < m) {
        const double s = dot_padded([&](int j) { return Q[lane + LDM * j]; }, [&](int j) { return L.x[j]; });
        sl = L.search[lane] * (s + c[lane]);
      }
      const double slope = wave_sum_row0(sl);
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
        while ((v - old_v) > kArmijo * (step * slope)) {  // (the reference's quotient test without the division: step * slope < 0 here; boxqp.hpp)
          step *= kStepDec;
          ILQR_QCOUNT(3)
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
    ILQR_QMARK(4)
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
  ILQR_QMARK(5)
  nfR_out = nfR;
  if (nfact_out) *nfact_out = nfact_last;
  return result;
}

// (Round 1's kernel k_backward_w -- every matrix of a step in LDS, 39.6 KB per wavefront -- lived here until ABI 5.  k_backward_w2
//  (backward_wave2.hpp) reproduces its bits with the matrices in registers and is the literal-order kernel the product keeps behind
//  ILQR_ROUTE_BACKWARD_W2; what remains in this file is what the register kernels share: the LDS block of the box-QP, the matrix-core
//  tile helpers, reductions, and w_box_qp -- boxqp.cpp:26-139 as written.)

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
