// derivatives.hpp -- the finite-difference sweep (src/derivatives.cpp + include/finite_diff.h) of one knot, into HBM records or
// into a slot of the LDS ring, with the pending commit of an accepted candidate performed on the way.
#pragma once
#include "rollout.hpp"

namespace ilqr {

// ------------------------------------------------------------------------------------------
// finite-difference derivatives
// ------------------------------------------------------------------------------------------
// include/finite_diff.h:67-86 applied to a scalar functor of an N-vector.
template <int N, class real, class F>
__device__ __forceinline__ void fd_hessian(const real* x, F f, real* out /* N x N col-major */) {
#pragma unroll
  for (int i = 0; i < N; i++)
#pragma unroll
    for (int j = i; j < N; j++) {
      real pp[N], pm[N], mp[N], mm[N];
#pragma unroll
      for (int l = 0; l < N; l++) pp[l] = pm[l] = mp[l] = mm[l] = x[l];
      pp[i] += real(kEps);
      pp[j] += real(kEps);
      pm[i] += real(kEps);
      pm[j] -= real(kEps);
      mp[i] -= real(kEps);
      mp[j] += real(kEps);
      mm[i] -= real(kEps);
      mm[j] -= real(kEps);
      const real v = (f(pp) - f(mp) - f(pm) + f(mm)) * real(1.0 / (4 * kEps * kEps));  // (x (1 / c), not / c: see derivatives_of_knot)
      out[i + N * j] = v;
      out[j + N * i] = v;
    }
}
// include/finite_diff.h:22-33
template <int N, class real, class F>
__device__ __forceinline__ void fd_gradient(const real* x, F f, real* out) {
#pragma unroll
  for (int i = 0; i < N; i++) {
    real p[N], m[N];
#pragma unroll
    for (int l = 0; l < N; l++) p[l] = m[l] = x[l];
    p[i] += real(kEps);
    m[i] -= real(kEps);
    out[i] = (f(p) - f(m)) * real(1.0 / (2 * kEps));
  }
}

// One thread per knot point (b, t), t = 0..T.  block = 256 = 16 trajectories x 16 time steps,
// grid = (ceil((T+1)/16), ntiles).  force != 0: every trajectory (stage call / bench mode),
// otherwise only running trajectories whose flgChange is set (ilqr_core.cpp:115).
// commit_idx (may be null): a line search accepted candidate commit_idx[b] for trajectory b and
// its copy into the nominal trajectory is still pending -- this kernel reads the knot from the
// candidate and performs the copy on the way (the separate k_commit pass is only used to flush).
// RING (k_sweep_backward): the record and the knot's nominal control are ALSO written to the LDS
// slot `rs` (this lane's pair column of the slot: element e at rs[(e>>1)*2*TW + (e&1)], the
// control behind the record), where the backward wavefront of the same block reads them.
#ifndef ILQR_RING_KB
#define ILQR_RING_KB 150  // one block per CU; the two-blocks-per-CU variant of k_sweep_backward uses 60
#endif
// PAD: extra `real`s per pair row.  0 = the HBM tile layout (row = 16 trajectories x 2 elements = a whole number of LDS
// bank cycles: rows of the same trajectory share their banks, which is what the 4-lane backward wavefront wants -- its
// lanes read one row for 16 trajectories).  A 16-lane-per-trajectory chain (tried in rounds 2 and 4, profiles/r04_hex_experiment.txt) reads up to 8 ROWS for
// one trajectory with one instruction: PAD = 2 (16 bytes) spreads the rows over the banks.
template <int NX, int NU, class real, int RING_KB = ILQR_RING_KB, int PAD = 0>
struct RingSlot {
  static constexpr int US = Rec<NX, NU>::SIZE;               // controls follow the record
  static constexpr int PAIRS = (Rec<NX, NU>::SIZE + NU + 1) / 2;
  static constexpr int ROW = 2 * TW + PAD;                   // `real`s from one element pair to the next
  static constexpr int ELEMS = PAIRS * ROW;                  // per slot, in units of `real`
  static constexpr int SLOTS = ((RING_KB * 1024 / (int)sizeof(real)) / ELEMS) / 4 * 4;  // ring size: what fits in RING_KB of LDS
};

// MFD / fdm: the model in the arithmetic the finite differences are TAKEN in.  For an fp64 handle that is
// the model itself.  For an fp32 handle it is its double-precision twin: eps = 1e-3 second differences of
// a cost of O(1e3) are rounding noise in float (1e3 x 6e-8 / 4e-6 = 15 against Hessian entries of 800), so
// the knot (float) is widened, the sweep below runs in double exactly as for an fp64 handle, and the record
// is rounded to float when stored.  Rollouts, the commit and the analytic route stay in the handle's own
// arithmetic (`model`).
// PERM: the pair rows of the slot are stored in the order hex_pair_pos gives them (the matrix-core chain's ring, backward_hex.hpp).
__host__ __device__ constexpr int hex_pair_pos(int p) {
  // fx (pairs 0..7) and cxx (pairs 12..19) are read by the 16-lanes-per-trajectory chain as element (r, c) = pair 2 c + (r >> 1)
  // -- and cxx also as (c, r) = pair 2 r + (c >> 1) -- by 32 lanes at a time (ds_read_b64: r in {0, 1} or {2, 3}): four pair rows per
  // access, 64 bytes each, and the LDS has four 64-byte bank groups.  In record order the rows 0, 2, 4, 6 fall on two groups (a
  // 2-way bank conflict on two of the step's nine reads); with q = 4 a + 2 b + c stored at 4 a + 2 b + (a ^ c) the four rows of
  // every one of those accesses ({c fixed}, {a fixed}) land on the four groups.
  const int q = (p < 8) ? p : p - 12;
  if (p >= 8 && (p < 12 || p >= 20)) return p;
  const int a = (q >> 2) & 1, b = (q >> 1) & 1, c = q & 1;
  return (p - q) + 4 * a + 2 * b + (a ^ c);
}
template <class M, bool RING = false, class MFD = M, int RING_PAD = 0, bool PERM = false>
__device__ __forceinline__ void derivatives_of_knot(const BatchViewT<typename M::real>& v, const M& model, const MFD& fdm, int force,
                                                    const int* __restrict__ commit_idx, int tile, int t, int l,
                                                    typename M::real* rs = nullptr, bool records = true) {
  using real = typename M::real;
  using fdr = typename MFD::real;
  using RSl = RingSlot<M::NX, M::NU, real>;
  constexpr int NX = M::NX, NU = M::NU;
  using R = Rec<NX, NU>;
  typedef real real2_t __attribute__((ext_vector_type(2)));
  const int b = tile * TW + l;
  const int T = v.T;
  if (t > T || b >= v.B) return;
  const int ci = commit_idx ? commit_idx[b] : -1;
  // Stand-alone sweep (k_derivatives): the records live in HBM, and a trajectory whose last line search failed
  // keeps them (flgChange = 0, ilqr_core.cpp:115).  Fused sweep (RING): the records exist ONLY in the LDS ring,
  // for as long as the backward wavefront needs them -- every running trajectory's are recomputed each time
  // (they are a function of the nominal trajectory: same values), nothing is written to HBM but the commit.
  // (records == false: the caller only wants the pending commit performed -- a fused sweep whose backward pass has moved on)
  const bool want = records && (force || (v.status[b] == 0 && (RING || v.flg_change[b])));
  if (ci < 0 && !want) return;
  const real dt = (real)v.dt;

  real xk[NX], uk[NU];  // the knot as stored
  {
    if (ci >= 0) {  // knot t of the accepted candidate
      candidate_knot(v, model, ci, tile, t, l, xk, uk);
    } else {
#pragma unroll
      for (int i = 0; i < NX; i++) xk[i] = v.xs[tidx(tile, t, i, l, T + 1, NX)];
#pragma unroll
      for (int j = 0; j < NU; j++) uk[j] = (t < T) ? v.us[tidx(tile, t, j, l, T, NU)] : real(0);  // derivatives.cpp:35-38
    }
    if (ci >= 0) {  // the pending commit of ilqr_core.cpp:210-213 ("accept": xs, us keep the new rollout)
#pragma unroll
      for (int i = 0; i < NX; i++) v.xs[tidx(tile, t, i, l, T + 1, NX)] = xk[i];
      if (t < T) {
#pragma unroll
        for (int j = 0; j < NU; j++) v.us[tidx(tile, t, j, l, T, NU)] = uk[j];
      }
    }
  }
  if (!want) return;  // (finished trajectory whose last candidate was committed above)

  real* D = RING ? nullptr : v.D + didx(tile, t, 0, l, T + 1, R::SIZE);
  auto put = [&](int e, fdr val_) {
    const real val = (real)val_;
    if (RING)
      rs[(PERM ? hex_pair_pos(e >> 1) : (e >> 1)) * (2 * TW + RING_PAD) + (e & 1)] = val;
    else
      D[(size_t)(e >> 1) * (2 * TW) + (e & 1)] = val;
  };
  auto put2 = [&](int e, fdr v0, fdr v1) {  // e even: one store of a pair
    real2_t w;
    w.x = (real)v0;
    w.y = (real)v1;
    if (RING)
      *reinterpret_cast<real2_t*>(rs + (PERM ? hex_pair_pos(e >> 1) : (e >> 1)) * (2 * TW + RING_PAD)) = w;
    else
      *reinterpret_cast<real2_t*>(D + (size_t)(e >> 1) * (2 * TW)) = w;
  };
  if (RING) {
#pragma unroll
    for (int j = 0; j < NU; j++) rs[((RSl::US + j) >> 1) * (2 * TW + RING_PAD) + ((RSl::US + j) & 1)] = uk[j];
    // m = 1: the slot's last pair has a free half next to u -- the weight 1 / (|u| + 1) of this knot's gradient-norm
    // term (:405-412) goes there, computed here (the same recip() of the same value) instead of on the backward
    // wavefront's chain
    if constexpr (NU == 1) rs[((RSl::US + 1) >> 1) * (2 * TW + RING_PAD) + ((RSl::US + 1) & 1)] = recip(abs_of(uk[0]) + real(1));
  }

  if constexpr (has_analytic_record<M>::value) {
    if (v.analytic) {  // opt-in: the model's exact derivatives (wave-uniform branch)
      real rec[R::SIZE];
      model.analytic_record(xk, uk, dt, t == T, rec);
#pragma unroll
      for (int e = 0; e < R::SIZE; e += 2) put2(e, (fdr)rec[e], (fdr)rec[e + 1]);
      return;
    }
  }
  // the knot in the finite differences' arithmetic (a no-op unless the handle is fp32)
  fdr x[NX], u[NU];
#pragma unroll
  for (int i = 0; i < NX; i++) x[i] = (fdr)xk[i];
#pragma unroll
  for (int j = 0; j < NU; j++) u[j] = (fdr)uk[j];
  const fdr dtf = (fdr)dt;
  // The reference DIVIDES its differences by 2 eps and 4 eps^2 (finite_diff.h:30,44,82).  Here they are multiplied by the
  // reciprocals (500 and 250000 to the last bit): a rounding-level deviation (<= 1 ulp of the entry, far inside the 1e-6 the
  // records are compared at) that removes twenty IEEE division sequences -- 230 of the sweep's 810 instructions per round.
  const fdr inv2eps = fdr(1.0 / (2 * kEps));
  if (t < T) {
    // fx, fu: central differences of the Euler map (derivatives.cpp:19-25, finite_diff.h:35-47)
#pragma unroll
    for (int i = 0; i < NX; i++) {
      fdr p[NX], m[NX], fp[NX], fm[NX];
#pragma unroll
      for (int q = 0; q < NX; q++) p[q] = m[q] = x[q];
      p[i] += fdr(kEps);
      m[i] -= fdr(kEps);
      integrate_dynamics(fdm, p, u, dtf, fp);
      integrate_dynamics(fdm, m, u, dtf, fm);
#pragma unroll
      for (int r = 0; r < NX; r += 2)
        put2(R::FX + r + NX * i, (fp[r] - fm[r]) * inv2eps, (fp[r + 1] - fm[r + 1]) * inv2eps);
    }
#pragma unroll
    for (int i = 0; i < NU; i++) {
      fdr p[NU], m[NU], fp[NX], fm[NX];
#pragma unroll
      for (int q = 0; q < NU; q++) p[q] = m[q] = u[q];
      p[i] += fdr(kEps);
      m[i] -= fdr(kEps);
      integrate_dynamics(fdm, x, p, dtf, fp);
      integrate_dynamics(fdm, x, m, dtf, fm);
#pragma unroll
      for (int r = 0; r < NX; r += 2)
        put2(R::FU + r + NX * i, (fp[r] - fm[r]) * inv2eps, (fp[r + 1] - fm[r + 1]) * inv2eps);
    }
    // cx, cu (derivatives.cpp:44-47)
    fdr g[NX > NU ? NX : NU];
    fd_gradient<NX>(x, [&](const fdr* xx) { return fdm.cost(xx, u); }, g);
#pragma unroll
    for (int i = 0; i < NX; i += 2) put2(R::CX + i, g[i], g[i + 1]);
    fd_gradient<NU>(u, [&](const fdr* uu) { return fdm.cost(x, uu); }, g);
#pragma unroll
    for (int i = 0; i < NU; i++) put(R::CU + i, g[i]);
    // cxx (derivatives.cpp:76-96)
    fdr H[NX * NX];
    fd_hessian<NX>(x, [&](const fdr* xx) { return fdm.cost(xx, u); }, H);
#pragma unroll
    for (int e = 0; e < NX * NX; e += 2) put2(R::CXX + e, H[e], H[e + 1]);
  } else {
#pragma unroll
    for (int e = 0; e < NX * NX + NX * NU; e += 2) put2(R::FX + e, fdr(0), fdr(0));  // fx[T], fu[T] stay zero
    fdr g[NX];
    fd_gradient<NX>(x, [&](const fdr* xx) { return fdm.final_cost(xx); }, g);  // :49
#pragma unroll
    for (int i = 0; i < NX; i += 2) put2(R::CX + i, g[i], g[i + 1]);
#pragma unroll
    for (int i = 0; i < NU; i++) put(R::CU + i, fdr(0));  // :50-51
    fdr H[NX * NX];
    fd_hessian<NX>(x, [&](const fdr* xx) { return fdm.final_cost(xx); }, H);  // :92
#pragma unroll
    for (int e = 0; e < NX * NX; e += 2) put2(R::CXX + e, H[e], H[e + 1]);
  }
  // cuu at every t, with u = 0 at t = T (derivatives.cpp:98-112)
  {
    fdr H[NU * NU];
    fd_hessian<NU>(u, [&](const fdr* uu) { return fdm.cost(x, uu); }, H);
#pragma unroll
    for (int e = 0; e < NU * NU; e++) put(R::CUU + e, H[e]);
  }
  // cxu (derivatives.cpp:114-144)
#pragma unroll
  for (int i = 0; i < NX; i++)
#pragma unroll
    for (int j = 0; j < NU; j++) {
      fdr px[NX], mx[NX], pu[NU], mu[NU];
#pragma unroll
      for (int q = 0; q < NX; q++) px[q] = mx[q] = x[q];
#pragma unroll
      for (int q = 0; q < NU; q++) pu[q] = mu[q] = u[q];
      px[i] += fdr(kEps);
      mx[i] -= fdr(kEps);
      pu[j] += fdr(kEps);
      mu[j] -= fdr(kEps);
      fdr val;
      if (t < T)
        val = (fdm.cost(px, pu) - fdm.cost(mx, pu) - fdm.cost(px, mu) + fdm.cost(mx, mu)) * fdr(1.0 / (4 * (kEps * kEps)));
      else  // :140 (the reference's own "TODO this is wrong"; value is never consumed)
        val = (fdm.final_cost(px) - fdm.final_cost(mx) - fdm.final_cost(px) + fdm.final_cost(mx)) * fdr(1.0 / (4 * (kEps * kEps)));
      put(R::CXU + i + NX * j, val);
    }
}

// grid = (ceil((T+1)/16), ntiles), block = 256 = 16 time steps x 16 trajectories
template <class M, class MFD = M>
__global__ __launch_bounds__(256) void k_derivatives(BatchViewT<typename M::real> v, M model, MFD fdm, int force, const int* __restrict__ commit_idx) {
  const int l = threadIdx.x & (TW - 1);
  const int t = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *v.n_running = 0;  // k_accept of this iteration recounts
  derivatives_of_knot<M, false, MFD>(v, model, fdm, force, commit_idx, (int)blockIdx.y, t, l);
}

}  // namespace ilqr
