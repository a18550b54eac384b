// kernels_wide2.hpp -- the wide tiles of kernels_wide.hpp for TWO controls (nx = 4, nu = 2: the reference's default example,
// include/double_integrator.h).  Same structure: a 64-trajectory tile, one THREAD per trajectory in the chain, producers -> LDS
// ring, the pending commit by a wavefront of its own, then the rollouts.  What differs from m = 1:
//   * a record is 58 scalars (+ 2 controls): a ring slot is 30 KB in fp64, four slots fill the LDS -- one tile per CU, two producers;
//   * the chain holds Vxx, fx, fu, Qxx, Qux, K, K'Quu and the 2 x 2 box-QP (box_qp2, boxqp.hpp) per thread: ~120 doubles live.
//     The block is FOUR wavefronts, one per SIMD, so that a wavefront may use 512 registers (the m = 1 kernels run two per SIMD);
//   * the four wavefronts roll out one 16-trajectory tile each, all eleven alphas per lane (rollout.hpp, NG = 3).
// Every element is computed by the expression, in the order, that backward_quad's m = 2 instantiation uses for it (that kernel
// is a column split of the same arithmetic): the route leaves the bits of k_solve_tile (tests/test_gpu_fused_sweep.py).
#pragma once
#include "kernels_wide.hpp"

namespace ilqr {

template <class M, class Gate, class RS>
__device__ __forceinline__ void backward_wide2(const BatchViewT<typename M::real>& v, const M& model, const SolverParams& sp, int mode, int wtile,
                                               int lane, Gate& gate, const typename M::real* __restrict__ ring) {
  using real = typename M::real;   // what is stored per knot
  using creal = double;            // what the recursion computes in (backward_quad.hpp: the mixed mode of fp32 handles)
  static_assert(M::NX == 4 && M::NU == 2, "wide tiles, two controls");
  constexpr int NU = 2;
  using R = Rec<4, 2>;
  const int tile = wtile * (WT / TW) + (lane >> 4), l = lane & (TW - 1);
  const int b = tile * TW + l;
  if (b >= v.B) return;
  if (mode == 1 && v.status[b] != 0) return;
  const int T = v.T;
  double lambda = v.lambda[b], dlambda = v.dlambda[b];
  const real* __restrict__ ust = v.us + tidx(tile, 0, 0, l, T, NU);
  real* __restrict__ kt = v.kff + tidx(tile, 0, 0, l, T, NU);
  real* __restrict__ Kt = v.Kfb + tidx(tile, 0, 0, l, T, NU * 4);
  typedef const __attribute__((address_space(3))) real lds_cd;

  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0, gacc = 0;
  auto one_pass = [&]() __attribute__((always_inline)) {
    gate.begin_pass();
    creal Vx[4], Vxx[16], kprev[NU];
    const creal lam_r = (creal)lambda;
    {
      gate.wait(T);
      lds_cd* r = (lds_cd*)(ring + gate.slot(T) * RS::ELEMS + lane * 2);
#pragma unroll
      for (int i = 0; i < 4; i++) Vx[i] = r[((R::CX + i) >> 1) * RS::ROW + ((R::CX + i) & 1)];  // :353
#pragma unroll
      for (int e = 0; e < 16; e++) Vxx[e] = r[((R::CXX + e) >> 1) * RS::ROW + ((R::CXX + e) & 1)];  // :354
    }
#pragma unroll
    for (int a = 0; a < NU; a++) kprev[a] = kt[((size_t)(T - 1) * NU + a) * TW];
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see backward_quad.hpp
    dV0 = dV1 = 0;
    diverge = 0;
    gacc = 0;
    for (int i = T - 1; i >= 0; i--) {
      gate.wait(i);
      lds_cd* r = (lds_cd*)(ring + gate.slot(i) * RS::ELEMS + lane * 2);
      auto el = [&](int e) { return (creal)r[(e >> 1) * RS::ROW + (e & 1)]; };
      creal fx[16], fu[8], us[NU];
#pragma unroll
      for (int e = 0; e < 16; e++) fx[e] = el(R::FX + e);
#pragma unroll
      for (int e = 0; e < 8; e++) fu[e] = el(R::FU + e);
#pragma unroll
      for (int a = 0; a < NU; a++) us[a] = el(RS::US + a);
      // replicated in the quad kernel: Qu, wv = Vxx' fu, Quu, QuuF     :360, :363, :367
      creal Qu[NU], Quu[NU * NU], QuuF[NU * NU];
#pragma unroll
      for (int a = 0; a < NU; a++) {
        creal acc = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) acc += fu[q + 4 * a] * Vx[q];
        Qu[a] = el(R::CU + a) + acc;
      }
#pragma unroll
      for (int c = 0; c < NU; c++) {
        creal wv[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += Vxx[rr + 4 * q] * fu[q + 4 * c];
          wv[rr] = acc;
        }
#pragma unroll
        for (int a = 0; a < NU; a++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += fu[q + 4 * a] * wv[q];
          const creal cuu = el(R::CUU + a + NU * c);
          Quu[a + NU * c] = cuu + acc;
          QuuF[a + NU * c] = (cuu + ((a == c) ? lam_r : creal(0))) + acc;
        }
      }
      // the quad kernel's lane s, for s = 0..3: W = Vxx' fx[:, s];  Qxx[:, s], Qx[s], Qux[:, s]     :359, :361, :362
      creal Qxx[16], Qx[4], Qux[NU][4];
#pragma unroll
      for (int s = 0; s < 4; s++) {
        creal W[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += Vxx[rr + 4 * q] * fx[q + 4 * s];
          W[rr] = acc;
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += fx[q + 4 * rr] * W[q];
          Qxx[rr + 4 * s] = el(R::CXX + rr + 4 * s) + acc;
        }
        {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += fx[q + 4 * s] * Vx[q];
          Qx[s] = el(R::CX + s) + acc;
        }
#pragma unroll
        for (int a = 0; a < NU; a++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) acc += fu[q + 4 * a] * W[q];
          Qux[a][s] = el(R::CXU + s + 4 * a) + acc;
        }
      }
      // :369  box-QP
      creal lo[NU], hi[NU];
#pragma unroll
      for (int a = 0; a < NU; a++) {
        lo[a] = model.u_min[a] - us[a];
        hi[a] = model.u_max[a] - us[a];
      }
      BoxQP2Result<creal> qr;
      box_qp2(QuuF, Qu, kprev, lo, hi, qr, false);
      const bool ok = qr.result >= 1;
      if (!ok) diverge = i;
      const creal x[NU] = {qr.x[0], qr.x[1]};
      // :373-385  K[:, s] = -(R^-1 R^-T) Qux[free, s] scattered to the free rows
      creal K[NU][4];
      {
        const bool both = qr.free0 & qr.free1;
#pragma unroll
        for (int s = 0; s < 4; s++) {
          const creal q0 = qr.free0 ? Qux[0][s] : Qux[1][s];  // rows_w_ind(Qux, v_free)(:, s), by rank
          const creal kA = (qr.nfR == 2) ? (-qr.m00 * q0 + -qr.m01 * Qux[1][s]) : -qr.m00 * q0;  // rank 0 (the second term only if both are free)
          const creal kB = -qr.m01 * Qux[0][s] + -qr.m11 * Qux[1][s];                              // rank 1
          K[0][s] = qr.free0 ? kA : creal(0);
          K[1][s] = qr.free1 ? (both ? kB : kA) : creal(0);
        }
      }
      // :388-389
      {
        creal d0 = 0;
#pragma unroll
        for (int a = 0; a < NU; a++) d0 += x[a] * Qu[a];
        if (ok) dV0 += (double)d0;
        creal d1 = 0;
#pragma unroll
        for (int c = 0; c < NU; c++) {
          creal rq = 0;
#pragma unroll
          for (int a = 0; a < NU; a++) rq += (creal(0.5) * x[a]) * Quu[a + NU * c];
          d1 += rq * x[c];
        }
        if (ok) dV1 += (double)d1;
      }
      // :391-393
      creal T1[NU][4];  // (K' Quu)[r, c]
#pragma unroll
      for (int c = 0; c < NU; c++)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          creal acc = 0;
#pragma unroll
          for (int q = 0; q < NU; q++) acc += K[q][rr] * Quu[q + NU * c];
          T1[c][rr] = acc;
        }
      creal Vxn[4], Vn[16];
#pragma unroll
      for (int s = 0; s < 4; s++) {
        {
          creal t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
          for (int c = 0; c < NU; c++) {
            t1 += T1[c][s] * x[c];
            t2 += K[c][s] * Qu[c];
            t3 += Qux[c][s] * x[c];
          }
          Vxn[s] = ((Qx[s] + t1) + t2) + t3;
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
          creal t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
          for (int q = 0; q < NU; q++) {
            t1 += T1[q][rr] * K[q][s];
            t2 += K[q][rr] * Qux[q][s];
            t3 += Qux[q][rr] * K[q][s];
          }
          Vn[rr + 4 * s] = ((Qxx[rr + 4 * s] + t1) + t2) + t3;
        }
      }
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        Vxx[rr + 4 * rr] = Vn[rr + 4 * rr];
#pragma unroll
        for (int c = rr + 1; c < 4; c++) {
          const creal sym = creal(0.5) * (Vn[rr + 4 * c] + Vn[c + 4 * rr]);
          Vxx[rr + 4 * c] = sym;
          Vxx[c + 4 * rr] = sym;
        }
        Vx[rr] = Vxn[rr];
      }
      // :405-412 term of the gradient norm
      {
        creal mx = 0;
#pragma unroll
        for (int a = 0; a < NU; a++) {
          const creal val = abs_of(x[a]) * (creal)recip((real)abs_of(us[a]) + real(1));
          mx = (a == 0 || val > mx) ? val : mx;
        }
        if (ok) gacc += (double)mx;
      }
      // :396-397
      if (ok) {
#pragma unroll
        for (int a = 0; a < NU; a++) {
          kprev[a] = (creal)(real)x[a];  // the stored gain, as the reference reads k[i + 1] back (:369)
          kt[(unsigned)((i * NU + a) * TW)] = (real)x[a];
#pragma unroll
          for (int s = 0; s < 4; s++) Kt[(unsigned)((i * NU * 4 + a + NU * s) * TW)] = (real)K[a][s];
        }
      }
      if (!ok) break;
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
    for (int t = 0; t < T; t++) {
      real mx = 0;
#pragma unroll
      for (int a = 0; a < NU; a++) {
        const real val = abs_of(kt[((size_t)t * NU + a) * TW]) / (abs_of(ust[((size_t)t * NU + a) * TW]) + 1);
        mx = (a == 0 || val > mx) ? val : mx;
      }
      acc += (double)mx;
    }
  }
  const double gnorm = acc / T;
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

// Whole iterations for ONE two-control wide tile (see k_solve_wide).  grid = ntiles / 4, block = 256 = four wavefronts:
// 0 the chain, 1 and 2 producers, 3 the pending commit; all four roll out (one 16-trajectory tile each, eleven alphas per lane).
template <class M, class MFD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_solve_wide2(BatchViewT<typename M::real> v, M model, MFD fdm, AlphaSet alphas,
                                                                                                 SolverParams sp, int n_iters, int force,
                                                                                                 int* __restrict__ commit_idx, int commit_pending,
                                                                                                 long long* __restrict__ phase_ticks) {
  using real = typename M::real;
  using Cfg = WideCfg<3>;
  __shared__ WideShared<real, M::NX, M::NU, Cfg::kProd, Cfg::kRingKb> sh;
  __shared__ int tile_running;
  const int wtile = blockIdx.x, wave = threadIdx.x >> 6;
  constexpr int kShareReals = 4 * ((2 * M::NU + M::NU * M::NX + M::NX + 3) / 4) * TW;
  static_assert(Cfg::kWaves * kShareReals <= (int)(sizeof(sh.ring) / sizeof(real)), "the ring holds every wavefront's rollout rows");
  real* const roll_share = sh.ring + wave * kShareReals;
  long long t_sweep = 0, t_roll = 0, t0 = 0;
  const bool timing = (phase_ticks != nullptr) & (threadIdx.x == 0);
  const long long c_begin = timing ? clock64() : 0, w_begin = timing ? wall_clock64() : 0;
  int it = 0;
  for (; it < n_iters; it++) {
    if (timing) t0 = wall_clock64();
    sweep_backward_wide<M, Cfg::kProd, MFD>(v, model, fdm, sp, 1, force, (it > 0 || commit_pending) ? commit_idx : nullptr, wtile, sh, wave);
    phase_barrier();
    if (timing) {
      const long long t1 = wall_clock64();
      t_sweep += t1 - t0;
      t0 = t1;
    }
    rollout_tile<M, true, true, Cfg::kPrefetch, false, true, true, 3>(v, model, alphas, NALPHA, v.cost_c, 1, sp, nullptr, wtile * (WT / TW) + wave, nullptr, false, 0, roll_share);
    phase_barrier();  // the candidates' costs are in memory
    if (threadIdx.x < WT)
      accept_one(v, sp, wtile * WT + (int)threadIdx.x, [&](int a) { return v.cost_c[(size_t)a * v.Bp + wtile * WT + threadIdx.x]; }, commit_idx,
                 /*count_running=*/it == n_iters - 1);
    if (threadIdx.x == 0) tile_running = 0;
    phase_barrier();  // status, lambda, commit indices are in memory for the next sweep
    if (timing) t_roll += wall_clock64() - t0;
    if (!sp.fixed_work) {
      const int b = wtile * WT + (int)threadIdx.x;
      if (threadIdx.x < WT && b < v.B && v.status[b] == 0) tile_running = 1;
      __syncthreads();
      if (!tile_running) {
        it++;
        break;
      }
    }
  }
  if (timing) {
    const long long cyc = clock64() - c_begin, wall = wall_clock64() - w_begin;
    for (int q = 0; q < WT / TW; q++) {  // (the host averages over 16-trajectory tiles)
      long long* p = phase_ticks + 5 * (wtile * (WT / TW) + q);
      p[0] += t_sweep;
      p[1] += t_roll;
      p[2] += it;
      p[3] += cyc;
      p[4] += wall;
    }
  }
}

}  // namespace ilqr
