// backward_wave2.hpp -- the generic backward pass of backward_wave.hpp with the step's matrices in REGISTERS, so that
// two wavefronts (nx <= 32) or more (nx <= 16) share a SIMD.
//
// k_backward_w keeps every matrix of a step in LDS (39.6 KB per wavefront: four wavefronts per CU, one per SIMD) and
// was measured latency-bound: the box-QP's dependent chains and the write -> read round trips between product
// phases leave the matrix cores idle four fifths of the time.  A second wavefront per SIMD needs <= 20 KB of LDS
// and <= 256 registers.  What makes that possible is the operand maps of v_mfma_f64_16x16x4_f64:
//     C/D:  lane l holds  X(row = 4 r + (l >> 4), col = l & 15),  r = 0..3         ("natural" registers of X)
//     B:    lane l feeds  B(k = 4 ks + (l >> 4), j = l & 15)   = the natural registers of B, r = ks
//     A:    lane l feeds  A(i = l & 15, k = 4 ks + (l >> 4))   = the natural registers of A', r = ks
// A product's result is therefore the next product's B operand as it stands, and the A operand of a product with
// its TRANSPOSE.  The recursion is rearranged so that every A operand is available transposed:
//     A1' = Vxx' fx,  A2' = Vxx' fu        instead of A1 = fx'Vxx, A2 = fu'Vxx      (:361-363)
//     Qxx = cxx + (A1')' fx,  Qux = cxu' + (A2')' fx,  Quu = cuu + (A2')' fu
//     T1' = Quu' K                          instead of T1 = K'Quu                    (:392)
//     Vn  = ((Qxx + (T1')' K) + K'Qux) + Qux'K
// Each element is the same k-ordered FMA chain over the same factor pairs as before (a b = b a exactly), so the
// results equal k_backward_w's bit for bit (tests/test_gpu_generic_backward.py compares the two kernels with
// array_equal).  The matrix-vector products (Qx, Qu, Vx) run on the matrix cores as well, against a B operand
// whose 16 columns all hold the vector.  Records are read from HBM straight into the natural registers (32-byte
// segments: four consecutive rows of a column per lane group).
//
// Left in LDS (19.5 KB): the box-QP's m x m matrices and vectors (w_box_qp is shared with k_backward_w), one
// 32 x 32 scratch matrix for the transpose in Vxx <- (Vn + Vn')/2, K for the rare stale-factor path.
#pragma once
#include "backward_wave.hpp"

namespace ilqr {

template <int NT>  // NT = 16-row tiles covering n: 1 (n <= 16) or 2 (n <= 32)
struct Wave2Lds {
  static constexpr int N = 16 * NT, LD = N + 1;
  static constexpr int S_LEN = (LD * N > 3 * LDM * WM) ? LD * N : 3 * LDM * WM;
  double S[S_LEN];        // Quu | QuuF | Minv (m x m, ld LDM) until T1'; then Vn (ld LD) for the symmetrisation
  double Kbuf[LDM * N];   // K (m x n, ld LDM): written only when the box-QP returns a stale factor or nothing free
  double Tbuf[LDM * N];   // Ri / the scattered Minv (m x m, ld LDM); Qux for the stale-factor path
  __device__ __forceinline__ double* K() { return Kbuf; }
  __device__ __forceinline__ double* Quu() { return S; }
  __device__ __forceinline__ double* QuuF() { return S + LDM * WM; }
  __device__ __forceinline__ double* Minv() { return S + 2 * LDM * WM; }
  __device__ __forceinline__ double* Qf() { return Tbuf; }
  __device__ __forceinline__ double* Ri() { return Tbuf; }
  double Vx[N], cx[N], Qx[N];
  double Qu[WM], x[WM], grad[WM], gc[WM], search[WM], lo[WM], hi[WM], clamped[WM], xc[WM], tmp[WM], kprev[WM],
      gfree[WM], xfree[WM];
  int vfree[WM], idx[WM];
};
static_assert(sizeof(Wave2Lds<2>) <= 20 * 1024, "two wavefronts per SIMD: 8 x LDS <= 160 KB");
static_assert(sizeof(Wave2Lds<1>) <= 160 * 1024 / 12, "three wavefronts per SIMD: 12 x LDS <= 160 KB");

// n <= 16 NT (NT = 1: three wavefronts per SIMD, NT = 2: two), m <= 16.  Arguments as k_backward_w.
template <int NT>
__global__ __launch_bounds__(64, NT == 2 ? 2 : 3) void k_backward_w2(BatchView v, int n, int m, const double* __restrict__ u_min,
                                                       const double* __restrict__ u_max, SolverParams sp, int mode,
                                                       const double* __restrict__ const_rec) {
  __shared__ Wave2Lds<NT> L;
  constexpr int LDX = Wave2Lds<NT>::LD;  // leading dimension of the n x n scratch
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
  const int orow = lane >> 4, ocol = lane & 15;
  {
    double* z = reinterpret_cast<double*>(&L);
    const int nz = (int)(sizeof(Wave2Lds<NT>) / sizeof(double));
    for (int e = lane; e < nz; e += 64) z[e] = 0.0;
  }
  lds_sync();

  auto mfma = [](double a, double b2, double4_t c) __attribute__((always_inline)) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, c, 0, 0, 0);
  };
  const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  // natural registers: X[ti][tj][r] = X(16 ti + 4 r + orow, 16 tj + ocol); k-step ks of a 32-long sum is [ks >> 2][.][ks & 3]
  struct RecA {
    double fx[NT][NT][4], fu[NT][4], cx, cu, us;
  };
  struct RecB {
    double cxx[NT][4], cxu[4], cuu[4];  // one 16-column block
  };
  // Addressing: every element of a record block is (lane part) + (wave-uniform part).  The lane parts are kept in
  // four registers and made opaque once per step (`asm volatile`): left alone, hipcc hoists all 52 guarded 64-bit
  // addresses of a step out of the loop and spills them (114 scratch stores in the prologue, reloads in every step).
  unsigned lb_nn = (unsigned)(orow + n * ocol);   // X(4 r + orow, ocol) of an n-row block
  unsigned lb_tn = (unsigned)(ocol + n * orow);   // X'(.., ..): cxu is stored n x m, Qux is m x n
  unsigned lb_mm = (unsigned)(orow + m * ocol);
  auto ldm = [](const double* r, bool in, unsigned off) __attribute__((always_inline)) {
    const double val = r[in ? off : 0u];
    return in ? val : 0.0;
  };
  auto load_rec_a = [&](int i, RecA& q) __attribute__((always_inline)) {
    const double* r = Db + (size_t)i * REC;
    const double* rm = const_rec ? const_rec : r;
    asm volatile("" : "+v"(lb_nn));
#pragma unroll
    for (int ti = 0; ti < NT; ti++)
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int a0 = 16 * ti + 4 * rr;
        const bool ain = a0 + orow < n;
#pragma unroll
        for (int tj = 0; tj < NT; tj++) q.fx[ti][tj][rr] = ldm(rm, ain && 16 * tj + ocol < n, lb_nn + (unsigned)(oFX + a0 + n * 16 * tj));
        q.fu[ti][rr] = ldm(rm, ain && ocol < m, lb_nn + (unsigned)(oFU + a0));
      }
    q.cx = ldm(r, lane < n, (unsigned)(oCX + lane));
    q.cu = ldm(r, lane < m, (unsigned)(oCU + lane));
    q.us = (lane < m) ? usb[(size_t)i * m + lane] : 0.0;
  };
  // cxx, cxu (and cuu with the first half) in the natural registers where they are added, one 16-column block of
  // the outputs at a time: the whole of it would be 56 registers held across the products that need the most
  auto load_rec_b = [&](int i, int tj, RecB& q) __attribute__((always_inline)) {
    const double* r = const_rec ? const_rec : Db + (size_t)i * REC;
    asm volatile("" : "+v"(lb_nn), "+v"(lb_tn), "+v"(lb_mm));
#pragma unroll
    for (int ti = 0; ti < NT; ti++)
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int a0 = 16 * ti + 4 * rr;
        q.cxx[ti][rr] = ldm(r, a0 + orow < n && 16 * tj + ocol < n, lb_nn + (unsigned)(oCXX + a0 + n * 16 * tj));
      }
#pragma unroll
    for (int rr = 0; rr < 4; rr++)  // Qux(a, c) starts from cxu(c, a): offset c + n a
      q.cxu[rr] = ldm(r, 4 * rr + orow < m && 16 * tj + ocol < n, lb_tn + (unsigned)(oCXU + 16 * tj + n * 4 * rr));
    if (tj == 0) {
#pragma unroll
      for (int rr = 0; rr < 4; rr++) q.cuu[rr] = ldm(r, 4 * rr + orow < m && ocol < m, lb_mm + (unsigned)(oCUU + 4 * rr));
    }
  };

  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0;
#ifdef ILQR_W2_TIMING
  W2Clock clk;
  clk.start();
#endif
  while (true) {
    double Vxx[NT][NT][4];
    {  // :353-354
      const double* r = Db + (size_t)T * REC;
      for (int e = lane; e < n; e += 64) L.Vx[e] = r[oCX + e];
#pragma unroll
      for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int tj = 0; tj < NT; tj++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int a0 = 16 * ti + 4 * rr;
            Vxx[ti][tj][rr] = ldm(r, a0 + orow < n && 16 * tj + ocol < n, lb_nn + (unsigned)(oCXX + a0 + n * 16 * tj));
          }
      if (lane < m) L.kprev[lane] = kb[(size_t)(T - 1) * m + lane];
    }
    dV0 = dV1 = 0;
    diverge = 0;
    lds_sync();
    for (int i = T - 1; i >= 0; i--) {
      ILQR_W2MARK(7)
      // No prefetch of the next record, unlike k_backward_w: its 54 registers, held through the box-QP (61 ms) or
      // from the end of the previous step (29 ms), make hipcc spill to scratch; loaded here, where they are consumed
      // (27 ms), the wavefront sharing the SIMD covers the wait.
      RecA cur;
      load_rec_a(i, cur);
      if (lane < m) {
        L.lo[lane] = u_min[lane] - cur.us;  // :369
        L.hi[lane] = u_max[lane] - cur.us;
        L.tmp[lane] = cur.cu;
      }
      if (lane < n) L.cx[lane] = cur.cx;
      double fx[NT][NT][4], fu[NT][4];
#pragma unroll
      for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          fu[ti][rr] = cur.fu[ti][rr];
#pragma unroll
          for (int tj = 0; tj < NT; tj++) fx[ti][tj][rr] = cur.fx[ti][tj][rr];
        }
      lds_sync();
      // :359-360 Qx = cx + fx'Vx, Qu = cu + fu'Vx -- B operand: Vx in every column
      {
        double4_t qx[NT];
        double bv[4 * NT];
#pragma unroll
        for (int ks = 0; ks < 4 * NT; ks++) bv[ks] = L.Vx[4 * ks + orow];
        double4_t qu = zero4;
#pragma unroll
        for (int ti = 0; ti < NT; ti++) qx[ti] = zero4;
#pragma unroll
        for (int ks = 0; ks < 4 * NT; ks++) {
#pragma unroll
          for (int ti = 0; ti < NT; ti++) qx[ti] = mfma(fx[ks >> 2][ti][ks & 3], bv[ks], qx[ti]);
          qu = mfma(fu[ks >> 2][ks & 3], bv[ks], qu);
        }
        if (ocol == 0) {
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            if (4 * rr + orow < m) L.Qu[4 * rr + orow] = L.tmp[4 * rr + orow] + qu[rr];
#pragma unroll
            for (int ti = 0; ti < NT; ti++) {
              const int a = 16 * ti + 4 * rr + orow;
              if (a < n) L.Qx[a] = L.cx[a] + qx[ti][rr];
            }
          }
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
      double Qxx[NT][NT][4], Qux[NT][4];
#pragma unroll
      for (int tj = 0; tj < NT; tj++) {  // one 16-column block of the outputs at a time (registers)
        // (scheduling fences: left free, hipcc hoists both blocks' loads above the first products, runs out of
        //  registers there and turns every load into load -> vmcnt(0) -> scratch spill)
        __builtin_amdgcn_sched_barrier(0);
        RecB rec;
        load_rec_b(i, tj, rec);
        __builtin_amdgcn_sched_barrier(0);
        double4_t qxx[NT], qux = zero4, quu = zero4;
#pragma unroll
        for (int ti = 0; ti < NT; ti++) qxx[ti] = zero4;
#pragma unroll
        for (int ks = 0; ks < 4 * NT; ks++) {
#pragma unroll
          for (int ti = 0; ti < NT; ti++) qxx[ti] = mfma(a1t[ks >> 2][ti][ks & 3], fx[ks >> 2][tj][ks & 3], qxx[ti]);
          qux = mfma(a2t[ks >> 2][ks & 3], fx[ks >> 2][tj][ks & 3], qux);
          if (tj == 0) quu = mfma(a2t[ks >> 2][ks & 3], fu[ks >> 2][ks & 3], quu);
        }
#pragma unroll
        for (int ti = 0; ti < NT; ti++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int a = 16 * ti + 4 * rr + orow, c = 16 * tj + ocol;
            Qxx[ti][tj][rr] = (a < n && c < n) ? rec.cxx[ti][rr] + qxx[ti][rr] : 0.0;
            asm volatile("" : "+v"(Qxx[ti][tj][rr]));  // (computed HERE: sunk below the box-QP, both addends stay live across it)
          }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int a = 4 * rr + orow, c = 16 * tj + ocol;
          Qux[tj][rr] = (a < m && c < n) ? rec.cxu[rr] + qux[rr] : 0.0;
          asm volatile("" : "+v"(Qux[tj][rr]));
        }
        if (tj == 0) {
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int a = 4 * rr + orow, c = ocol;
            const bool in = (a < m && c < m);
            const double cuu = in ? rec.cuu[rr] : 0.0;
            L.Quu()[a + LDM * c] = in ? cuu + quu[rr] : 0.0;
            L.QuuF()[a + LDM * c] = in ? (cuu + ((a == c) ? lambda : 0.0)) + quu[rr] : 0.0;
          }
        }
      }
      lds_sync();
      ILQR_W2MARK(2)
      int nfR = 0;
      const int result = w_box_qp(m, L, lane, nfR ILQR_W2CLOCK_PASS, nullptr, sp.fixes);
      ILQR_W2MARK(3)
      if (result < 1) {  // :371
        diverge = i;
        break;
      }
      // :373-385  K rows of free dims, natural registers K[tj][r] = K(4 r + orow, 16 tj + ocol)
      double K[NT][4];
      const unsigned long long free_mask = __ballot(lane < m && L.vfree[lane]);
      const int nf = __popcll(free_mask);
      if (nf > 0 && nf == nfR) {
        // K = -(Minv scattered to the free rows / columns of an m x m matrix) Qux: see k_backward_w
        double* MF = L.Qf();
        if (nf == m) {
          MF = L.Minv();
        } else {
          if (lane < m && L.vfree[lane]) L.idx[__popcll(free_mask & ((1ull << lane) - 1ull))] = lane;
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
          for (int ks = 0; ks < WM / 4; ks++) acc = mfma(aM[ks], Qux[tj][ks], acc);
#pragma unroll
          for (int rr = 0; rr < 4; rr++) K[tj][rr] = -acc[rr];
        }
      } else {  // nothing free, or a stale factor of another size (:80): through LDS, as k_backward_w does
        if (lane < m && L.vfree[lane]) L.idx[__popcll(free_mask & ((1ull << lane) - 1ull))] = lane;
        for (int c = lane >> 4; c < n; c += 4) L.K()[(lane & 15) + LDM * c] = 0;
#pragma unroll
        for (int tj = 0; tj < NT; tj++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) L.Tbuf[(4 * rr + orow) + LDM * (16 * tj + ocol)] = Qux[tj][rr];
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
          for (int rr = 0; rr < 4; rr++) K[tj][rr] = L.K()[(4 * rr + orow) + LDM * (16 * tj + ocol)];
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
      // T1' = Quu' K (m x n)
      double4_t t1t[NT];
      {
        double aQ[WM / 4];
        ld_operand<WM / 4>([&](int i2, int k) { return L.Quu()[k + LDM * i2]; }, lane, aQ);
#pragma unroll
        for (int tj = 0; tj < NT; tj++) {
          t1t[tj] = zero4;
#pragma unroll
          for (int ks = 0; ks < WM / 4; ks++) t1t[tj] = mfma(aQ[ks], K[tj][ks], t1t[tj]);
        }
      }
      // :391 Vx = ((Qx + T1 k) + K'Qu) + Qux'k -- three matrix-vector products against replicated columns
      {
        double bx[WM / 4], bq[WM / 4];
#pragma unroll
        for (int ks = 0; ks < WM / 4; ks++) {
          bx[ks] = L.x[4 * ks + orow];
          bq[ks] = L.Qu[4 * ks + orow];
        }
        double qxv[NT][4];
#pragma unroll
        for (int ti = 0; ti < NT; ti++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++) qxv[ti][rr] = L.Qx[16 * ti + 4 * rr + orow];
#pragma unroll
        for (int ti = 0; ti < NT; ti++) {
          double4_t s1 = zero4, s2 = zero4, s3 = zero4;
#pragma unroll
          for (int ks = 0; ks < WM / 4; ks++) {
            s1 = mfma(t1t[ti][ks], bx[ks], s1);
            s2 = mfma(K[ti][ks], bq[ks], s2);
            s3 = mfma(Qux[ti][ks], bx[ks], s3);
          }
          if (ocol == 0) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
              const int a = 16 * ti + 4 * rr + orow;
              if (a < n) L.Vx[a] = ((qxv[ti][rr] + s1[rr]) + s2[rr]) + s3[rr];
            }
          }
        }
      }
      lds_sync();  // (Quu, x, Qu have been read: S may take Vn)
      ILQR_W2MARK(5)
      // :392 Vn = ((Qxx + T1 K) + K'Qux) + Qux'K ; :393 Vxx = (Vn + Vn')/2 through S
#pragma unroll
      for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int tj = 0; tj < NT; tj++) {
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
            L.S[(16 * ti + 4 * rr + orow) + LDX * (16 * tj + ocol)] = vn;
          }
        }
      lds_sync();
#pragma unroll
      for (int ti = 0; ti < NT; ti++)
#pragma unroll
        for (int tj = 0; tj < NT; tj++)
#pragma unroll
          for (int rr = 0; rr < 4; rr++)
            Vxx[ti][tj][rr] = 0.5 * (Vxx[ti][tj][rr] + L.S[(16 * tj + ocol) + LDX * (16 * ti + 4 * rr + orow)]);
      // :396-397
      if (lane < m) {
        kb[(size_t)i * m + lane] = L.x[lane];
        L.kprev[lane] = L.x[lane];
      }
#pragma unroll
      for (int tj = 0; tj < NT; tj++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          const int a = 4 * rr + orow, c = 16 * tj + ocol;
          if (a < m && c < n) Kb[(size_t)i * m * n + a + m * c] = K[tj][rr];
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
