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

template <class M>
__global__ __launch_bounds__(64) void k_backward_h(BatchViewT<typename M::real> v, M model, SolverParams sp, int mode) {
  using real = typename M::real;
  static_assert(M::NX == 4 && M::NU == 1, "16 lanes per trajectory: 4 x 4 state matrices, scalar control");
  using R = Rec<4, 1>;
  typedef real real2_t __attribute__((ext_vector_type(2)));
  __shared__ real lds_steps[104];
  load_step_table(lds_steps);
  const int lane = threadIdx.x;
  const int tile = (int)blockIdx.x >> 2, sub = (int)blockIdx.x & 3;
  const int j = lane >> 4, c = (lane >> 2) & 3, r = lane & 3, s = r;
  const int l = 4 * sub + j;  // trajectory inside the tile
  const int b = tile * TW + l;
  if (b >= v.B) return;                       // row-uniform
  if (mode == 1 && v.status[b] != 0) return;  // row-uniform
  const int T = v.T;
  double lambda = v.lambda[b], dlambda = v.dlambda[b];
  // Addressing: a wave-uniform base per tile (SGPR pair) + 32-bit BYTE offsets per lane and element, so that a load is
  // one v_add_u32 (step offset + element offset) and a global_load with an SGPR base -- per-lane 64-bit pointers cost
  // 27 address instructions per step.
  const char* __restrict__ Dtile = reinterpret_cast<const char*>(v.D + didx(tile, 0, 0, 0, T + 1, R::SIZE));
  const real* __restrict__ ust = v.us + tidx(tile, 0, 0, l, T, 1);
  real* __restrict__ kt = v.kff + tidx(tile, 0, 0, l, T, 1);
  real* __restrict__ Kt = v.Kfb + tidx(tile, 0, 0, l, T, 4);
  constexpr unsigned kStepBytes = (R::SIZE / 2) * 2 * TW * sizeof(real);
  auto off = [&](int e) { return (unsigned)(((e >> 1) * (2 * TW) + (e & 1) + 2 * l) * sizeof(real)); };
  // per-lane element offsets inside a record (constant over the pass)
  const unsigned o_fxc0 = off(R::FX + 4 * c), o_fxc1 = off(R::FX + 4 * c + 2);
  const unsigned o_fxr0 = off(R::FX + 4 * r), o_fxr1 = off(R::FX + 4 * r + 2);
  unsigned o_fxs[4];
#pragma unroll
  for (int k = 0; k < 4; k++) o_fxs[k] = off(R::FX + ((c - k) & 3) + 4 * c);
  const unsigned o_fur = off(R::FU + r), o_cxc = off(R::CX + c), o_cxx = off(R::CXX + r + 4 * c),
                 o_cxuc = off(R::CXU + c), o_cxur = off(R::CXU + r);
  const unsigned o_fu0 = off(R::FU), o_fu1 = off(R::FU + 2), o_tail = off(R::CU);
  const int tlane = (lane & ~15) + 4 * r + c;  // the lane holding element [c, r]

  auto load = [&](int t, HexStep<real>& d) __attribute__((always_inline)) {
    const unsigned tb = (unsigned)t * kStepBytes;
    auto pair = [&](unsigned o) { return *reinterpret_cast<const real2_t*>(Dtile + (tb + o)); };
    auto one = [&](unsigned o) { return *reinterpret_cast<const real*>(Dtile + (tb + o)); };
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
    d.us = ust[(unsigned)(t * TW)];
  };

  constexpr int kWaitAll = (7 << 4) | (15 << 8);  // s_waitcnt vmcnt(0) only
  int diverge = 0;
  bool done = false;
  double dV0 = 0, dV1 = 0, gacc = 0;
  auto one_pass = [&]() __attribute__((always_inline)) {
    real Vxx, Vx[4], kprev;
    const real lam_r = (real)lambda;
    {
      const unsigned tb = (unsigned)T * kStepBytes;
      Vxx = *reinterpret_cast<const real*>(Dtile + (tb + o_cxx));  // :354
#pragma unroll
      for (int i = 0; i < 4; i++) Vx[i] = *reinterpret_cast<const real*>(Dtile + (tb + off(R::CX + i)));  // :353
    }
    kprev = kt[(size_t)(T - 1) * TW];
    dV0 = dV1 = 0;
    diverge = 0;
    gacc = 0;
    auto step = [&](int i, const HexStep<real>& raw) -> bool {
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
      __builtin_amdgcn_s_waitcnt(kWaitAll);
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

}  // namespace ilqr
