// generic.hpp -- the device-model path for state / control dimensions up to 32 / 16.
//
//   LqModel           device twin of the synthetic LQ model (BASELINE.json configs[4])
//   k_rollout_g<M>    forward_pass (src/ilqr_core.cpp:305-337): one THREAD per rollout, the state
//                     and control vectors lane-private in registers
//   k_derivatives_g<M> the finite-difference sweep (src/derivatives.cpp, include/finite_diff.h):
//                     one WAVEFRONT per knot, one LANE per evaluation point
//
// Everything here works on the trajectory-contiguous ("AoS") layout of the generic handles
//     xs [b][T+1][nx]   us,kff [b][T][nu]   Kfb [b][T][nu*nx] (column-major nu x nx)
//     D  [b][T+1][REC]  record order FX,FU,CX,CXX,CXU,CU,CUU (common.hpp)
// that k_backward_w (backward_wave.hpp) consumes.  A device model has compile-time MAXIMUM
// dimensions NX, NU (vectors are register arrays, every loop is unrolled, padding entries are
// zero and stay zero) and runtime dimensions nx <= NX, nu <= NU which bound what is loaded,
// perturbed and stored.
#pragma once
#include "kernels.hpp"  // AlphaSet, quad_bcast, integrate_dynamics

namespace ilqr {

constexpr int GN = 32, GM = 16;
#ifndef ILQR_FD_POINTS_X
#define ILQR_FD_POINTS_X 2  // evaluation points per lane in the cxx sweep (measured at B=1024: 2 -> 51.5 ms, 3 -> 50.9, 4 -> 69.6)
#endif

// Reads through the constant address space: the matrices of a model are the same for every lane
// and are not written while a kernel runs, so their loads become scalar loads (s_load) and the
// products take them as SGPR operands.
typedef const __attribute__((address_space(4))) double cmem_d;

// Synthetic LQ model: xdot = A x + B u, cost 0.5 (x'Qx + u'Ru), final cost 0.5 x'Qf x.  The
// matrices are row-major and zero-padded to the maximum dimensions.  Sums run left to right
// over the column index (the padding adds exact zeros at the end of every sum).
struct LqModel {
  using real = double;  // the generic path is fp64 only
  static constexpr int NX = GN, NU = GM;
  int nx, nu;
  const double *A, *Bm, *Q, *R, *Qf;  // device: [GN][GN], [GN][GM], [GN][GN], [GM][GM], [GN][GN]
  const double *umin = nullptr, *umax = nullptr;  // device [nu]: Model::u_min / u_max (only the opt-in clamped rollout reads them)
  __device__ __forceinline__ double limit_lo(int j) const { return umin[j]; }
  __device__ __forceinline__ double limit_hi(int j) const { return umax[j]; }

  __device__ __forceinline__ void dynamics(const double* x, const double* u, double* dx) const {
    cmem_d* a = (cmem_d*)A;
    cmem_d* bm = (cmem_d*)Bm;
#pragma unroll
    for (int i = 0; i < GN; i += 4) {  // four independent row sums in flight (see quad)
      double acc[4] = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < GN; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] += a[(i + q) * GN + j] * x[j];
#pragma unroll
      for (int j = 0; j < GM; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] += bm[(i + q) * GM + j] * u[j];
#pragma unroll
      for (int q = 0; q < 4; q++) dx[i + q] = acc[q];
    }
  }
  template <int N>
  static __device__ __forceinline__ double quad(cmem_d* Mx, const double* vv) {
    // four rows at a time: the same sums in the same order, but four independent accumulator
    // chains in flight (one wavefront per SIMD has nothing else to hide the FMA latency)
    static_assert(N % 4 == 0, "row blocking");
    double s = 0;
#pragma unroll
    for (int i = 0; i < N; i += 4) {
      double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
#pragma unroll
      for (int j = 0; j < N; j++) {
        r0 += Mx[(i + 0) * N + j] * vv[j];
        r1 += Mx[(i + 1) * N + j] * vv[j];
        r2 += Mx[(i + 2) * N + j] * vv[j];
        r3 += Mx[(i + 3) * N + j] * vv[j];
      }
      s += vv[i] * r0;
      s += vv[i + 1] * r1;
      s += vv[i + 2] * r2;
      s += vv[i + 3] * r3;
    }
    return s;
  }
  // the same quadratic form at TWO points: every matrix element is fetched once and used twice
  // (the finite-difference sweep is bound by how fast the scalar loads deliver the matrices)
  template <int N>
  static __device__ __forceinline__ void quad2(cmem_d* Mx, const double* va, const double* vb, double& sa, double& sb) {
    static_assert(N % 2 == 0, "row blocking");
    sa = sb = 0;
#pragma unroll
    for (int i = 0; i < N; i += 2) {
      double ra0 = 0, ra1 = 0, rb0 = 0, rb1 = 0;
#pragma unroll
      for (int j = 0; j < N; j++) {
        const double m0 = Mx[i * N + j], m1 = Mx[(i + 1) * N + j];
        ra0 += m0 * va[j];
        rb0 += m0 * vb[j];
        ra1 += m1 * va[j];
        rb1 += m1 * vb[j];
      }
      sa += va[i] * ra0;
      sa += va[i + 1] * ra1;
      sb += vb[i] * rb0;
      sb += vb[i + 1] * rb1;
    }
  }
  // The running cost is a function of x plus a function of u.  A model that says so (kSeparableCost)
  // lets the finite-difference sweep evaluate each part once per DISTINCT perturbed argument and
  // assemble cost(x, u) = cost_from_parts(cost_x(x), cost_u(u)) -- the same arithmetic on the same
  // values, so the same bits, from 2.6x fewer multiply-adds at n = 32, m = 16.
#ifndef ILQR_LQ_SEPARABLE
#define ILQR_LQ_SEPARABLE 1  // (0: experiment build that evaluates every point through cost(x, u))
#endif
  static constexpr bool kSeparableCost = ILQR_LQ_SEPARABLE != 0;
  static constexpr bool kHasAnalyticRecord = true;
  __device__ __forceinline__ double cost_x(const double* x) const { return quad<GN>((cmem_d*)Q, x); }
  __device__ __forceinline__ double cost_u(const double* u) const { return quad<GM>((cmem_d*)R, u); }
  // P points at once: every matrix element is fetched once and used P times
  template <int N, int P>
  static __device__ __forceinline__ void quadP(cmem_d* Mx, const double (*vv)[N], double* sums) {
    static_assert(N % 2 == 0, "row blocking");
#pragma unroll
    for (int h = 0; h < P; h++) sums[h] = 0;
#pragma unroll
    for (int i = 0; i < N; i += 2) {
      double r0[P], r1[P];
#pragma unroll
      for (int h = 0; h < P; h++) r0[h] = r1[h] = 0;
#pragma unroll
      for (int j = 0; j < N; j++) {
        const double m0 = Mx[i * N + j], m1 = Mx[(i + 1) * N + j];
#pragma unroll
        for (int h = 0; h < P; h++) {
          r0[h] += m0 * vv[h][j];
          r1[h] += m1 * vv[h][j];
        }
      }
#pragma unroll
      for (int h = 0; h < P; h++) {
        sums[h] += vv[h][i] * r0[h];
        sums[h] += vv[h][i + 1] * r1[h];
      }
    }
  }
  template <int P>
  __device__ __forceinline__ void cost_xP(const double (*xs)[GN], double* q) const { quadP<GN, P>((cmem_d*)Q, xs, q); }
  template <int P>
  __device__ __forceinline__ void cost_uP(const double (*us)[GM], double* q) const { quadP<GM, P>((cmem_d*)R, us, q); }
  __device__ __forceinline__ void cost_x2(const double* xa, const double* xb, double& qa, double& qb) const {
    quad2<GN>((cmem_d*)Q, xa, xb, qa, qb);
  }
  __device__ __forceinline__ void cost_u2(const double* ua, const double* ub, double& qa, double& qb) const {
    quad2<GM>((cmem_d*)R, ua, ub, qa, qb);
  }
  static __device__ __forceinline__ double cost_from_parts(double qx, double qu) { return 0.5 * (qx + qu); }
  // cost_x is a quadratic form x'Mx with this (zero-padded, row-major) matrix: the 2n(n+1) points of
  // the cxx sweep can then go through the matrix cores as one [32 x 32] x [32 x points] product
#ifndef ILQR_LQ_MFMA_CXX
#define ILQR_LQ_MFMA_CXX 1
#endif
  static constexpr bool kQuadraticCostX = ILQR_LQ_MFMA_CXX != 0;
  __device__ __forceinline__ cmem_d* cost_x_matrix() const { return (cmem_d*)Q; }
  // xdot = A x + B u with these (zero-padded, row-major) matrices, summed over x then over u: the
  // 2 (nx + nu) points of the Jacobian sweep then go through the matrix cores as well
#ifndef ILQR_LQ_MFMA_FX
#define ILQR_LQ_MFMA_FX 1
#endif
  static constexpr bool kLinearDynamics = ILQR_LQ_MFMA_FX != 0;
  __device__ __forceinline__ const double* dynamics_x_matrix() const { return A; }
  __device__ __forceinline__ const double* dynamics_u_matrix() const { return Bm; }
  // cost_u is the quadratic form u'Mu with this matrix (the cuu sweep on the matrix cores)
  __device__ __forceinline__ const double* cost_u_matrix() const { return R; }
  __device__ __forceinline__ double cost(const double* x, const double* u) const {
    return cost_from_parts(cost_x(x), cost_u(u));
  }
  __device__ __forceinline__ void cost2(const double* xa, const double* ua, const double* xb, const double* ub, double& fa,
                                        double& fb) const {
    double qa, qb, ra, rb;
    cost_x2(xa, xb, qa, qb);
    cost_u2(ua, ub, ra, rb);
    fa = cost_from_parts(qa, ra);
    fb = cost_from_parts(qb, rb);
  }
  __device__ __forceinline__ double final_cost(const double* x) const { return 0.5 * quad<GN>((cmem_d*)Qf, x); }
  // exact derivatives (opt-in), written cooperatively by the 64 lanes of the knot's wavefront into
  // the record D (runtime offsets for nx, nu; conventions of derivatives.cpp at t = T); x, u point
  // to the knot in memory (u may be null at t = T)
  __device__ __forceinline__ void analytic_record(const double* __restrict__ x, const double* __restrict__ u, double dt, bool last,
                                                  double* __restrict__ D, int lane) const {
    const int oFX = 0, oFU = oFX + nx * nx, oCX = oFU + nx * nu, oCXX = oCX + nx, oCXU = oCXX + nx * nx, oCU = oCXU + nx * nu,
              oCUU = oCU + nu;
    const double* Wx = last ? Qf : Q;
    for (int e = lane; e < nx * nx; e += 64) {
      const int r = e % nx, c = e / nx;
      D[oFX + e] = last ? 0.0 : ((r == c) ? 1.0 : 0.0) + dt * A[r * GN + c];
      D[oCXX + e] = 0.5 * (Wx[r * GN + c] + Wx[c * GN + r]);
    }
    for (int e = lane; e < nx * nu; e += 64) {
      const int r = e % nx, c = e / nx;
      D[oFU + e] = last ? 0.0 : dt * Bm[r * GM + c];
      D[oCXU + e] = 0.0;
    }
    for (int e = lane; e < nu * nu; e += 64) {
      const int r = e % nu, c = e / nu;
      D[oCUU + e] = 0.5 * (R[r * GM + c] + R[c * GM + r]);
    }
    for (int i = lane; i < nx; i += 64) {  // d/dx 0.5 x'Wx = 0.5 (W + W') x
      double acc = 0;
      for (int j = 0; j < nx; j++) acc += 0.5 * (Wx[i * GN + j] + Wx[j * GN + i]) * x[j];
      D[oCX + i] = acc;
    }
    for (int i = lane; i < nu; i += 64) {
      double acc = 0;
      if (!last)
        for (int j = 0; j < nu; j++) acc += 0.5 * (R[i * GM + j] + R[j * GM + i]) * u[j];
      D[oCU + i] = acc;
    }
  }
};

// ------------------------------------------------------------------------------------------
// forward rollout
// ------------------------------------------------------------------------------------------
//   RG_INIT    u_t = us[t]                                (init_traj: K empty, ilqr_core.cpp:316)
//              one lane per trajectory; writes xs, us and the cost in place
//   RG_SEARCH  u_t = us[t] + alpha k[t] + K[t](x_t - xs[t])  for the 11 alphas (:185-220)
//              a wavefront = 5 trajectories x 11 alphas, so the rollouts of one trajectory
//              fetch its nominal knots and gains together; only the cost leaves the kernel
//   RG_COMMIT  the same rollout for the ONE alpha k_accept chose (commit_idx), written over the
//              nominal xs/us in place (knot t is read before it is overwritten); one lane per
//              trajectory.  write_cost: warm start (:65-76), where this rollout also defines cost.
// No candidate trajectories are stored on this path: with nx + nu = 48 doubles per knot they
// would be 11 x 77 KB per trajectory and iteration; re-running the accepted rollout costs 1/11 of
// the search.
// A user's device twin (models.hpp: UserModelT<double>, any NX <= 32, NU <= 16 that the tiled nx = 4 kernels do not take) as the
// generic kernels see a model: compile-time dimensions = runtime dimensions, no structure promised (every perturbed point of
// the finite-difference sweep is evaluated through dynamics() / cost() / final_cost() as src/derivatives.cpp does).
template <class U>
struct GenericModelOf : U {
  int nx = U::NX, nu = U::NU;
  __device__ __forceinline__ double limit_lo(int j) const { return this->u_min[j]; }
  __device__ __forceinline__ double limit_hi(int j) const { return this->u_max[j]; }
  static constexpr bool kSeparableCost = false, kQuadraticCostX = false, kLinearDynamics = false, kHasAnalyticRecord = false;
  __device__ __forceinline__ void cost2(const double* xa, const double* ua, const double* xb, const double* ub, double& fa, double& fb) const {
    fa = this->cost(xa, ua);
    fb = this->cost(xb, ub);
  }
};

enum { RG_INIT = 0, RG_SEARCH = 1, RG_COMMIT = 2 };
constexpr int kSearchTraj = 64 / NALPHA;  // trajectories per wavefront in RG_SEARCH (5)

template <class M, int MODE>
__global__ __launch_bounds__(64) void k_rollout_g(BatchView v, M model, AlphaSet alphas, double* __restrict__ cost_out,
                                                  const int* __restrict__ commit_idx, int mode, int write_cost, int fixes) {
  constexpr int NX = M::NX, NU = M::NU;
  const int nx = model.nx, nu = model.nu, T = v.T;
  const int lane = threadIdx.x;
  int b, a = 0;
  if (MODE == RG_SEARCH) {
    const int tl = lane / NALPHA;
    a = lane - tl * NALPHA;
    b = blockIdx.x * kSearchTraj + tl;
    if (tl >= kSearchTraj) return;
  } else {
    b = blockIdx.x * 64 + lane;
  }
  if (b >= v.B) return;
  if (MODE == RG_SEARCH && mode == 1 && !(v.status[b] == 0 && v.backpass_done[b])) return;
  if (MODE == RG_COMMIT) {
    a = commit_idx[b];
    if (a < 0) return;
  }
  double alpha = 0;
#pragma unroll
  for (int q = 0; q < NALPHA; q++)
    if (a == q) alpha = alphas.a[q];
  const double dt = v.dt;

  double x[NX];
#pragma unroll
  for (int i = 0; i < NX; i++) x[i] = (i < nx) ? v.x0[(size_t)b * nx + i] : 0.0;
  double total = 0;
  double* xsb = v.xs + (size_t)b * (T + 1) * nx;
  double* usb = v.us + (size_t)b * T * nu;
  const double* kb = v.kff + (size_t)b * T * nu;
  const double* Kb = v.Kfb + (size_t)b * T * nu * nx;
  for (int t = 0; t < T; t++) {
    double u[NU];
#pragma unroll
    for (int j = 0; j < NU; j++) u[j] = (j < nu) ? usb[(size_t)t * nu + j] : 0.0;
    if (MODE != RG_INIT) {
      double d[NX];
#pragma unroll
      for (int i = 0; i < NX; i++) d[i] = (i < nx) ? x[i] - xsb[(size_t)t * nx + i] : 0.0;
      const double* Kt = Kb + (size_t)t * nu * nx;
#pragma unroll
      for (int j = 0; j < NU; j++) {
        if (j < nu) {
          u[j] += kb[(size_t)t * nu + j] * alpha;  // :190
          double acc = 0;
#pragma unroll
          for (int i = 0; i < NX; i++)
            if (i < nx) acc += Kt[j + nu * i] * d[i];
          u[j] += acc;  // :316
        }
      }
    }
    if (fixes & 1) {  // opt-in (ILQR_FLAG_REFERENCE_FIXES): "the right way" of ilqr_core.cpp:327-329 -- the clamped control is stored and integrated
#pragma unroll
      for (int j = 0; j < NU; j++)
        if (j < nu) u[j] = fmin(fmax(u[j], model.limit_lo(j)), model.limit_hi(j));
    }
    if (MODE != RG_SEARCH) {  // :323 (no clamping)
#pragma unroll
      for (int i = 0; i < NX; i++)
        if (i < nx) xsb[(size_t)t * nx + i] = x[i];
#pragma unroll
      for (int j = 0; j < NU; j++)
        if (j < nu) usb[(size_t)t * nu + j] = u[j];
    }
    total += model.cost(x, u);  // :324
    double x1[NX];
    integrate_dynamics(model, x, u, dt, x1);  // :325
#pragma unroll
    for (int i = 0; i < NX; i++) x[i] = x1[i];
  }
  if (MODE != RG_SEARCH) {
#pragma unroll
    for (int i = 0; i < NX; i++)
      if (i < nx) xsb[(size_t)T * nx + i] = x[i];
  }
  total += model.final_cost(x);  // :335
  if (MODE == RG_SEARCH)
    cost_out[(size_t)a * v.Bp + b] = total;
  else if (MODE == RG_INIT || write_cost)
    cost_out[b] = total;
}

// ------------------------------------------------------------------------------------------
// exact derivatives of the LQ model (ILQR_FLAG_ANALYTIC_DERIVATIVES)
// ------------------------------------------------------------------------------------------
// The record of a knot t < T is constant (fx = I + dt A, fu = dt B, cxx, cuu, cxu = 0) except for
// cx = sym(Q) x_t and cu = sym(R) u_t; k_backward_w still wants one 27 KB record per knot, so this
// kernel is a store stream: 44 GB per sweep at B = 8192, T = 200.  One wavefront per chunk of
// kAnalyticChunk knots of one trajectory holds the constant part in registers in its store mapping
// (rows along 32 lanes, two columns per instruction: 256-byte runs) and streams it out per knot;
// values are the expressions of LqModel::analytic_record, which still writes knot T.
constexpr int kAnalyticChunk = 8;
// what: 0 = whole records; 1 = only what differs from knot to knot (cx, cu; the whole record of knot T)
// plus ONE copy of the constant matrices in const_rec, which k_backward_w then reads for every knot
// t < T -- 77 KB + one record per trajectory instead of 5.4 MB; 2 = the constant matrices of every
// knot t < T, unconditionally (fills in what 1 skipped, for the derivative getter).
__global__ __launch_bounds__(64) void k_analytic_lq(BatchView v, LqModel model, int force, int what, double* __restrict__ const_rec,
                                                    int chunk) {
  static_assert(GN == 32 && GM == 16, "store mapping below is written for a 32 x 16 model");
  const int nx = model.nx, nu = model.nu, T = v.T;
  const int lane = threadIdx.x;
  const int nchunk = (T + 1 + chunk - 1) / chunk;  // knots per wavefront: kAnalyticChunk, or more when only cx, cu are written
  const int b = blockIdx.x / nchunk, t0 = (blockIdx.x - b * nchunk) * chunk;
  if (what != 2 && what != 3 && blockIdx.x == 0 && lane == 0) *v.n_running = 0;  // k_accept of this iteration recounts
  const bool fill_const = ((what == 1 || what == 3) && blockIdx.x == 0);  // what = 3: ONLY the two constant records (k_backward_w3's fused route)
  if (what == 3 && blockIdx.x != 0) return;
  if (what != 2 && !fill_const && !(force || (v.status[b] == 0 && v.flg_change[b]))) return;
  const bool skip_knots = (what == 1) && !(force || (v.status[b] == 0 && v.flg_change[b]));  // (block 0 came for const_rec only)
  const int oFX = 0, oFU = oFX + nx * nx, oCX = oFU + nx * nu, oCXX = oCX + nx, oCXU = oCXX + nx * nx, oCU = oCXU + nx * nu,
            oCUU = oCU + nu, REC = oCUU + nu * nu;
  const double dt = v.dt;
  const int r32 = lane & 31, chalf = lane >> 5, r16 = lane & 15, cq = lane >> 4;
  double fx[16], cxx[16], fu[8], cuu[4], wx[GN], wu[GM];
  const bool need_matrices = (what != 1) || fill_const;  // (wave-uniform: a vectors-only wavefront skips these loads)
  if (need_matrices) {
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int c = 2 * j + chalf;
    fx[j] = ((r32 == c) ? 1.0 : 0.0) + dt * model.A[r32 * GN + c];
    cxx[j] = 0.5 * (model.Q[r32 * GN + c] + model.Q[c * GN + r32]);
  }
#pragma unroll
  for (int j = 0; j < 8; j++) fu[j] = dt * model.Bm[r32 * GM + 2 * j + chalf];
#pragma unroll
  for (int j = 0; j < 4; j++) cuu[j] = 0.5 * (model.R[r16 * GM + 4 * j + cq] + model.R[(4 * j + cq) * GM + r16]);
  }
  // even nx: the same values as row PAIRS (rows 2 rp, 2 rp + 1 of column 4 j + cq), stored 16 bytes per
  // lane -- 1 KB per store instruction instead of 512 B (every offset of the record is even then)
  typedef double double2v __attribute__((ext_vector_type(2)));
  const bool pairs = (nx & 1) == 0;
  const int rp = lane & 15;
  double2v fx2[8], cxx2[8], fu2[4];
  if (need_matrices) {
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int c = 4 * j + cq;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int r = 2 * rp + h;
      fx2[j][h] = ((r == c) ? 1.0 : 0.0) + dt * model.A[r * GN + c];
      cxx2[j][h] = 0.5 * (model.Q[r * GN + c] + model.Q[c * GN + r]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int h = 0; h < 2; h++) fu2[j][h] = dt * model.Bm[(2 * rp + h) * GM + 4 * j + cq];
  }
  // rows of sym(Q) on lanes 0..31, rows of sym(R) on lanes 32..47
#pragma unroll
  for (int j = 0; j < GN; j++) wx[j] = 0.5 * (model.Q[r32 * GN + j] + model.Q[j * GN + r32]);
#pragma unroll
  for (int j = 0; j < GM; j++) wu[j] = 0.5 * (model.R[r16 * GM + j] + model.R[j * GM + r16]);

  // knot -1 stands for const_rec (matrices only)
  for (int t = fill_const ? -1 : t0; t < t0 + chunk && t <= T; t++) {
    if (t >= 0 && skip_knots) break;
    double* __restrict__ D = (t < 0) ? const_rec : v.D + ((size_t)b * (T + 1) + t) * REC;
    const bool matrices = (t < 0) || what != 1, vectors = (t >= 0) && what != 2;
    if (t == T) {
      if (what != 2) model.analytic_record(v.xs + ((size_t)b * (T + 1) + t) * nx, nullptr, dt, true, D, lane);
      break;
    }
    double accx = 0, accu = 0;
    if (vectors) {
      // the knot arrives with ONE coalesced load per vector (lane j holds x_j / u_j) and is handed round with
      // v_readlane; 48 separate broadcast loads per knot made this the whole cost of the sweep once the
      // matrices were no longer stored per knot.  Padding terms are fma(0, 0, acc): the sums are unchanged.
      const double* __restrict__ x = v.xs + ((size_t)b * (T + 1) + t) * nx;
      const double* __restrict__ u = v.us + ((size_t)b * T + t) * nu;
      const double xv = (lane < nx) ? x[lane] : 0.0, uv = (lane < nu) ? u[lane] : 0.0;
      auto lane_value = [](double val, int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(val), l), __builtin_amdgcn_readlane(__double2loint(val), l));
      };
#pragma unroll
      for (int j = 0; j < GN; j++) accx = __builtin_fma(wx[j], lane_value(xv, j), accx);
#pragma unroll
      for (int j = 0; j < GM; j++) accu = __builtin_fma(wu[j], lane_value(uv, j), accu);
      if (lane < nx) D[oCX + lane] = accx;
      if (lane >= GN && lane - GN < nu) D[oCU + lane - GN] = accu;
    }
    if (!matrices) continue;
    if (pairs) {
      if (2 * rp < nx) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int c = 4 * j + cq;
          if (c < nx) {
            *reinterpret_cast<double2v*>(D + oFX + 2 * rp + nx * c) = fx2[j];
            *reinterpret_cast<double2v*>(D + oCXX + 2 * rp + nx * c) = cxx2[j];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int c = 4 * j + cq;
          if (c < nu) {
            *reinterpret_cast<double2v*>(D + oFU + 2 * rp + nx * c) = fu2[j];
            *reinterpret_cast<double2v*>(D + oCXU + 2 * rp + nx * c) = double2v{0.0, 0.0};
          }
        }
      }
    } else if (r32 < nx) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int c = 2 * j + chalf;
        if (c < nx) {
          D[oFX + r32 + nx * c] = fx[j];
          D[oCXX + r32 + nx * c] = cxx[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int c = 2 * j + chalf;
        if (c < nu) {
          D[oFU + r32 + nx * c] = fu[j];
          D[oCXU + r32 + nx * c] = 0.0;
        }
      }
    }
    if (r16 < nu) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (4 * j + cq < nu) D[oCUU + r16 + nu * (4 * j + cq)] = cuu[j];
    }
    if (what == 3) {  // ... and knot T's record behind it (its cx belongs to nobody's x_T: the fused backward pass forms cx[T] = cxx[T] x_T itself)
      model.analytic_record(v.xs, nullptr, dt, true, const_rec + REC, lane);
      return;
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward rollout of the LQ model on the matrix cores
// ------------------------------------------------------------------------------------------
// One WAVEFRONT per trajectory; the 11 candidate rollouts of the line search are the columns of
// a [32 x 16] state block X (columns 11..15 run alpha = 0 and are dropped), and every product of
// a rollout step is a chain of v_mfma_f64_16x16x4_f64 over that block:
//     U  = us + alpha k + K (X - xs)          K D   : 1 tile  x 8 k-steps
//     qx = diag(X' (Q X)),  qu = diag(U' (R U))   : 2 x 8 + 8,  4 + 4
//     X1 = X + dt (A X + B U)                      : 2 tiles x (8 + 4)
// 64 MFMAs per knot for 16 columns, against 3.3 K FMAs per knot and rollout of the thread-per-
// rollout kernel above (k_rollout_g, whose fully unrolled scalar-operand products spill SGPRs: 38 ms
// per search at B = 8192, T = 200).  What makes this layout work:
//   * The C/D map of the instruction (lane l holds rows (l>>4) + 4r, column l&15) is also its B
//     map with k-step r (k = 4r + (l>>4), j = l&15) and, transposed, its A map.  So a product's
//     output registers ARE the next product's operand registers: the state block, the control
//     block and Q X / R U never leave registers or change lanes for the whole rollout.
//   * x'(Qx) for 16 columns is the diagonal of X'(QX): one more MFMA chain, whose k order is the
//     order of the scalar sum (the 15/16 off-diagonal outputs are the price).
//   * A, B, Q, R in operand layout are 44 doubles per lane, loaded once per kernel.
// Each chain runs k ascending from a zero accumulator, as the sums of LqModel::dynamics / quad do,
// so the results are those of k_rollout_g bit for bit (tests/test_gpu_lq_end_to_end.py compares them).
// Modes as above.  RG_COMMIT / RG_INIT run the same block with every column on the same rollout and
// store column 0.  RG_SEARCH with candidate buffers (v.cand_x / v.cand_u, [b][alpha][t][row]): every column stores its
// states and controls on the way, so that the commit of the accepted one is a copy (k_commit_lq) instead of a twelfth
// rollout as long as the eleven (the 3.5 KB per step and trajectory leave under the MFMA chains).
// ACCEPT (RG_SEARCH with candidate buffers only): the wavefront also performs STEP 3/4 for its trajectory (accept_one: selection,
// lambda schedule, termination -- k_accept's work); the copy of the accepted candidate stays k_commit_lq's.  An iteration of the LQ
// path with exact derivatives is then three launches (backward pass, search + accept, commit) instead of five.  commit_idx is written, not read.
#ifndef ILQR_ROLLOUT_LQ_WAVES
#define ILQR_ROLLOUT_LQ_WAVES 1   // one wavefront per SIMD (312 registers incl. accumulators).  Two (256 registers, 160 bytes of scratch) was measured
                                  // SLOWER at configs[4]: 4.5 -> 5.5 ms per search (round 6)
#endif
template <int MODE, bool ACCEPT = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ILQR_ROLLOUT_LQ_WAVES, ILQR_ROLLOUT_LQ_WAVES))) void k_rollout_lq(BatchView v, LqModel model, AlphaSet alphas, double* __restrict__ cost_out,
                                                   int* __restrict__ commit_idx, int mode, int write_cost, SolverParams sp) {
  static_assert(GN == 32 && GM == 16, "operand blocks below are written for a 32 x 16 model");
  static_assert(!ACCEPT || MODE == RG_SEARCH, "the accept epilogue belongs to the search");
  typedef double double4_t __attribute__((ext_vector_type(4)));
  const int nx = model.nx, nu = model.nu, T = v.T;
  const int lane = threadIdx.x, g = lane >> 4, p = lane & 15;
  const int b = blockIdx.x;
  if (b >= v.B) return;
  if (MODE == RG_SEARCH && mode == 1 && !(v.status[b] == 0 && v.backpass_done[b])) {
    if (ACCEPT && lane == 0) accept_one(v, sp, b, [](int) { return 0.0; }, commit_idx);  // (no search: the "no step" branch, :264-282; a finished trajectory: commit -1)
    return;
  }
  int a = p;
  if (MODE == RG_COMMIT) {
    a = commit_idx[b];
    if (a < 0) return;
  }
  double alpha = 0;
#pragma unroll
  for (int q = 0; q < NALPHA; q++)
    if (a == q) alpha = alphas.a[q];
  const double dt = v.dt;
  auto mfma = [](double x, double y, double4_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); };
  const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};

  // model matrices as A operands: row (16 ti + p), k = 4 ks + g
  double opA[2][8], opQ[2][8], opB[2][4], opR[4];
#pragma unroll
  for (int ti = 0; ti < 2; ti++) {
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      opA[ti][ks] = model.A[(16 * ti + p) * GN + 4 * ks + g];
      opQ[ti][ks] = model.Q[(16 * ti + p) * GN + 4 * ks + g];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) opB[ti][ks] = model.Bm[(16 * ti + p) * GM + 4 * ks + g];
  }
#pragma unroll
  for (int ks = 0; ks < 4; ks++) opR[ks] = model.R[p * GM + 4 * ks + g];

  double* xsb = v.xs + (size_t)b * (T + 1) * nx;
  double* usb = v.us + (size_t)b * T * nu;
  const double* kb = v.kff + (size_t)b * T * nu;
  const double* Kb = v.Kfb + (size_t)b * T * nu * nx;
  const bool clamp_u = (sp.fixes & 1) != 0;  // opt-in (ILQR_FLAG_REFERENCE_FIXES): the clamped control is stored and integrated, ilqr_core.cpp:327-329
  double ulo[4], uhi[4];                     // limits of this lane's control rows g + 4 r
#pragma unroll
  for (int r = 0; r < 4; r++) {
    ulo[r] = (clamp_u && g + 4 * r < nu) ? model.umin[g + 4 * r] : 0.0;
    uhi[r] = (clamp_u && g + 4 * r < nu) ? model.umax[g + 4 * r] : 0.0;
  }

  // this lane's share of knot t: xs rows 4ks+g, K(p, 4ks+g), us / k rows g+4r
  struct Knot {
    double xs[8], K[8], us[4], k[4];
  };
  auto load_knot = [&](int t, Knot& q) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      const int i = 4 * ks + g;
      q.xs[ks] = (MODE != RG_INIT && i < nx) ? xsb[(size_t)t * nx + i] : 0.0;
      q.K[ks] = (MODE != RG_INIT && i < nx && p < nu) ? Kb[(size_t)t * nu * nx + p + nu * i] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int j = g + 4 * r;
      q.us[r] = (j < nu) ? usb[(size_t)t * nu + j] : 0.0;
      q.k[r] = (MODE != RG_INIT && j < nu) ? kb[(size_t)t * nu + j] : 0.0;
    }
  };

  // the candidate of column p (RG_SEARCH with buffers)
  const bool keep = MODE == RG_SEARCH && v.cand_x != nullptr && p < NALPHA;
  double* cxb = keep ? v.cand_x + ((size_t)b * NALPHA + p) * (T + 1) * nx : nullptr;
  double* cub = keep ? v.cand_u + ((size_t)b * NALPHA + p) * T * nu : nullptr;

  double x[8];  // X rows 4ks+g, column p
#pragma unroll
  for (int ks = 0; ks < 8; ks++) x[ks] = (4 * ks + g < nx) ? v.x0[(size_t)b * nx + 4 * ks + g] : 0.0;
  double4_t total = zero4;  // running cost of column p sits on the diagonal: lane g == (p & 3), element p >> 2
  Knot cur;
  if (T > 0) load_knot(0, cur);
  for (int t = 0; t < T; t++) {
    const Knot kn = cur;
    if (t + 1 < T) load_knot(t + 1, cur);  // next knot in flight during this one
    double u[4];
    double4_t ax[2] = {zero4, zero4}, qx[2] = {zero4, zero4};
    if (MODE != RG_INIT) {
      double d[8];
#pragma unroll
      for (int ks = 0; ks < 8; ks++) d[ks] = x[ks] - kn.xs[ks];
      double4_t kd = zero4;
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {  // three independent chains share the issue slots
        kd = mfma(kn.K[ks], d[ks], kd);
#pragma unroll
        for (int ti = 0; ti < 2; ti++) {
          qx[ti] = mfma(opQ[ti][ks], x[ks], qx[ti]);
          ax[ti] = mfma(opA[ti][ks], x[ks], ax[ti]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        double uj = kn.us[r];
        uj += kn.k[r] * alpha;  // :190
        uj += kd[r];            // :316
        u[r] = uj;
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ks++)
#pragma unroll
        for (int ti = 0; ti < 2; ti++) {
          qx[ti] = mfma(opQ[ti][ks], x[ks], qx[ti]);
          ax[ti] = mfma(opA[ti][ks], x[ks], ax[ti]);
        }
#pragma unroll
      for (int r = 0; r < 4; r++) u[r] = kn.us[r];
    }
    if (clamp_u) {
#pragma unroll
      for (int r = 0; r < 4; r++) u[r] = fmin(fmax(u[r], ulo[r]), uhi[r]);
    }
    if (MODE != RG_SEARCH && p == 0) {  // :323 (no clamping)
#pragma unroll
      for (int ks = 0; ks < 8; ks++)
        if (4 * ks + g < nx) xsb[(size_t)t * nx + 4 * ks + g] = x[ks];
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (g + 4 * r < nu) usb[(size_t)t * nu + g + 4 * r] = u[r];
    }
    if (keep) {
#pragma unroll
      for (int ks = 0; ks < 8; ks++)
        if (4 * ks + g < nx) cxb[(size_t)t * nx + 4 * ks + g] = x[ks];
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (g + 4 * r < nu) cub[(size_t)t * nu + g + 4 * r] = u[r];
    }
    // :324 cost = 0.5 (x'Qx + u'Ru)
    double4_t ru = zero4, cx = zero4, cu = zero4;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      ru = mfma(opR[ks], u[ks], ru);
#pragma unroll
      for (int ti = 0; ti < 2; ti++) ax[ti] = mfma(opB[ti][ks], u[ks], ax[ti]);
      cx = mfma(x[ks], qx[0][ks], cx);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      cx = mfma(x[4 + ks], qx[1][ks], cx);
      cu = mfma(u[ks], ru[ks], cu);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) total[r] += LqModel::cost_from_parts(cx[r], cu[r]);
    // :325 x1 = x + dx dt
#pragma unroll
    for (int ti = 0; ti < 2; ti++)
#pragma unroll
      for (int r = 0; r < 4; r++) x[4 * ti + r] = x[4 * ti + r] + ax[ti][r] * dt;
  }
  if (MODE != RG_SEARCH && p == 0) {
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
      if (4 * ks + g < nx) xsb[(size_t)T * nx + 4 * ks + g] = x[ks];
  }
  if (keep) {
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
      if (4 * ks + g < nx) cxb[(size_t)T * nx + 4 * ks + g] = x[ks];
  }
  {  // :335 final cost 0.5 x'Qf x
    double4_t qf[2] = {zero4, zero4}, cf = zero4;
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
      for (int ti = 0; ti < 2; ti++) qf[ti] = mfma(model.Qf[(16 * ti + p) * GN + 4 * ks + g], x[ks], qf[ti]);
#pragma unroll
    for (int ks = 0; ks < 8; ks++) cf = mfma(x[ks], qf[ks >> 2][ks & 3], cf);
#pragma unroll
    for (int r = 0; r < 4; r++) total[r] += 0.5 * cf[r];
  }
  // the diagonal element of column p: lane g == (p & 3), element p >> 2
  double mine = total[0];
#pragma unroll
  for (int r = 1; r < 4; r++)
    if ((p >> 2) == r) mine = total[r];
  if (g == (p & 3)) {
    if (MODE == RG_SEARCH) {
      if (p < NALPHA) cost_out[(size_t)p * v.Bp + b] = mine;
    } else if (p == 0 && (MODE == RG_INIT || write_cost)) {
      cost_out[b] = mine;
    }
  }
  if constexpr (ACCEPT) {
    __shared__ double cst[NALPHA];
    if (g == (p & 3) && p < NALPHA) cst[p] = mine;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // (one wavefront: orders its LDS accesses)
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) accept_one(v, sp, b, [&](int a2) { return cst[a2]; }, commit_idx);
    // (the copy of the accepted candidate over the nominal trajectory stays a launch of its own, k_commit_lq, where every CU streams:
    //  one wavefront per trajectory copying its 77 KB here was measured 0.25 ms per iteration slower)
  }
}

// The commit of the accepted candidates (ilqr_core.cpp:210-213) when k_rollout_lq<RG_SEARCH> kept them: one block per
// trajectory copies states and controls of column commit_idx[b] over the nominal ones.  The bits of the rollout that was scored.
__global__ __launch_bounds__(256) void k_commit_lq(BatchView v, int nx, int nu, const int* __restrict__ commit_idx) {
  const int b = blockIdx.x;
  const int a = commit_idx[b];
  if (a < 0) return;
  const int T = v.T;
  const size_t nxs = (size_t)(T + 1) * nx, nus = (size_t)T * nu;
  const double* __restrict__ cx = v.cand_x + ((size_t)b * NALPHA + a) * nxs;
  const double* __restrict__ cu = v.cand_u + ((size_t)b * NALPHA + a) * nus;
  double* __restrict__ xs = v.xs + (size_t)b * nxs;
  double* __restrict__ us = v.us + (size_t)b * nus;
  for (size_t i = threadIdx.x; i < nxs; i += blockDim.x) xs[i] = cx[i];
  for (size_t i = threadIdx.x; i < nus; i += blockDim.x) us[i] = cu[i];
}

// ------------------------------------------------------------------------------------------
// finite-difference derivatives
// ------------------------------------------------------------------------------------------
// One wavefront per knot (b, t); every lane evaluates the model at ONE perturbed point, then
// neighbouring lanes combine their values (DPP inside a quad) exactly as the reference's
// expressions do:
//   fx, fu   column i = (F(+eps e_i) - F(-eps e_i)) / 2eps,  F = Euler map       finite_diff.h:35-47
//   cx, cu   (f(+) - f(-)) / 2eps                                                  finite_diff.h:22-33
//   cxx, cuu (f(pp) - f(mp) - f(pm) + f(mm)) / 4eps^2, j >= i, mirrored; the perturbations are
//            applied one after the other (the diagonal is x + eps + eps, x + eps - eps, ...)   :67-86
//   cxu      (c(px,pu) - c(mx,pu) - c(px,mu) + c(mx,mu)) / 4eps^2               derivatives.cpp:114-144
// including the t = T special cases (fx[T] = fu[T] = 0, cx/cxx from final_cost, cu[T] = 0, cuu[T]
// from cost(x_T, 0), the cxu[T] formula the reference itself marks wrong).
template <class M>
// t_only >= 0: one block per trajectory, knot t_only alone (the last knot behind k_derivatives_lq, which sweeps the knots t < T).
#ifndef ILQR_FD_G_WAVES_SMALL
// Wavefronts per SIMD of k_derivatives_g: two at NX > 16 (186 registers for the LQ model's 32-vectors); THREE for a model of NX <= 16 (168
// registers, 40 bytes of scratch): every lane of this kernel is a latency-bound chain through the user's cost function, and the third
// wavefront hides more of it than the few spilled values cost -- the pendulum chain's sweep 17.2 -> 14.0 ms (four: 128 registers, 216
// bytes of scratch, 18.7 ms).  Same instructions per lane: the same bits.
#define ILQR_FD_G_WAVES_SMALL 3
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((M::NX <= 16 ? ILQR_FD_G_WAVES_SMALL : 2), (M::NX <= 16 ? ILQR_FD_G_WAVES_SMALL : 2)))) void k_derivatives_g(BatchView v, M model, int force, int t_only) {
  constexpr int NX = M::NX, NU = M::NU;
  const int nx = model.nx, nu = model.nu, T = v.T;
  const int lane = threadIdx.x;
  const int b = (t_only >= 0) ? blockIdx.x : blockIdx.x / (T + 1), t = (t_only >= 0) ? t_only : blockIdx.x - b * (T + 1);
  if (t_only < 0 && blockIdx.x == 0 && lane == 0) *v.n_running = 0;  // k_accept of this iteration recounts
  if (!(force || (v.status[b] == 0 && v.flg_change[b]))) return;
  const int oFX = 0, oFU = oFX + nx * nx, oCX = oFU + nx * nu, oCXX = oCX + nx, oCXU = oCXX + nx * nx, oCU = oCXU + nx * nu,
            oCUU = oCU + nu, REC = oCUU + nu * nu;
  double* D = v.D + ((size_t)b * (T + 1) + t) * REC;
  const bool last = (t == T);

#define ILQR_DMARK(k)
  double x[NX], u[NU];  // the knot (same in every lane): one coalesced load per vector, handed round with
                        // v_readlane (NX + NU same-address loads cost ~10 us per knot, see k_analytic_lq)
  {
    static_assert(NX <= 64 && NU <= 64, "one lane per component");
    const double xv = (lane < nx) ? v.xs[((size_t)b * (T + 1) + t) * nx + lane] : 0.0;
    const double uv = (lane < nu && !last) ? v.us[((size_t)b * T + (last ? 0 : t)) * nu + lane] : 0.0;  // derivatives.cpp:35-38
    auto lane_value = [](double val, int l) {
      return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(val), l), __builtin_amdgcn_readlane(__double2loint(val), l));
    };
#pragma unroll
    for (int i = 0; i < NX; i++) x[i] = lane_value(xv, i);
#pragma unroll
    for (int j = 0; j < NU; j++) u[j] = lane_value(uv, j);
  }

  if constexpr (M::kHasAnalyticRecord) {
    if (v.analytic) {  // opt-in: the model's exact derivatives (reads the knot from memory: runtime indices)
      model.analytic_record(v.xs + ((size_t)b * (T + 1) + t) * nx, last ? nullptr : v.us + ((size_t)b * T + t) * nu, v.dt, last, D, lane);
      return;
    }
  } else if constexpr (has_analytic_record<M>::value) {
    // a user twin's own analytic_record (the contract of models.hpp: one thread writes the whole record, Rec<> order, column-major
    // blocks, derivatives.cpp's conventions at t = T) -- the record's offsets here are that order at nx = NX, nu = NU
    if (v.analytic) {
      if (lane == 0) model.analytic_record(x, u, v.dt, last, D);
      return;
    }
  }

  // point = knot, then (target 1, index i1) += d1, then (target 2, index i2) += d2; index -1 = none
  auto perturbed = [&](bool x1, int i1, double d1, bool x2, int i2, double d2, double* px, double* pu)
                       __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NX; c++) {
      double val = x[c];
      val = (x1 && c == i1) ? val + d1 : val;
      val = (x2 && c == i2) ? val + d2 : val;
      px[c] = val;
    }
#pragma unroll
    for (int c = 0; c < NU; c++) {
      double val = u[c];
      val = (!x1 && c == i1) ? val + d1 : val;
      val = (!x2 && c == i2) ? val + d2 : val;
      pu[c] = val;
    }
  };

  // ---- fx, fu ----
  if (!last) {
    const int E = 2 * (nx + nu);
    if constexpr (M::kLinearDynamics && NX == 32 && NU == 16) {
      // F(P) = P + dt (A Px + B Pu) for 16 points at a time on the matrix cores: lane (g = l >> 4,
      // p = l & 15) supplies component 4 ks + g of point base + p and receives rows g + 4 r (+16) of its
      // image -- the components it supplied; each row's sum runs over x, then over u, k ascending, as
      // LqModel::dynamics does.  The thread-per-point form of this product (below) needs A and B as
      // scalar operands of 1.5 K unrolled FMAs per lane and spent most of its time spilling SGPRs:
      // it was half of the whole sweep.
      typedef double double4_t __attribute__((ext_vector_type(4)));
      const int g = lane >> 4, p16 = lane & 15;
      double opA[2][8], opB[2][4], xg[8], ug[4];
      const double* Am = model.dynamics_x_matrix();
      const double* Bmm = model.dynamics_u_matrix();
#pragma unroll
      for (int ti = 0; ti < 2; ti++) {
#pragma unroll
        for (int ks = 0; ks < 8; ks++) opA[ti][ks] = Am[(16 * ti + p16) * NX + 4 * ks + g];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) opB[ti][ks] = Bmm[(16 * ti + p16) * NU + 4 * ks + g];
      }
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {  // x[4 ks + g] without indexing the register array by g
        const double a0 = x[4 * ks], a1 = x[4 * ks + 1], a2 = x[4 * ks + 2], a3 = x[4 * ks + 3];
        xg[ks] = (g == 0) ? a0 : (g == 1) ? a1 : (g == 2) ? a2 : a3;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const double a0 = u[4 * ks], a1 = u[4 * ks + 1], a2 = u[4 * ks + 2], a3 = u[4 * ks + 3];
        ug[ks] = (g == 0) ? a0 : (g == 1) ? a1 : (g == 2) ? a2 : a3;
      }
      for (int base = 0; base < E; base += 16) {
        const int e = base + p16;
        const bool valid = e < E;
        const int var = e >> 1;
        const double d = (e & 1) ? -kEps : kEps;
        const int ix = (valid && var < nx) ? var : -1, iu = (valid && var >= nx) ? var - nx : -1;
        double bx[8], bu[4];
#pragma unroll
        for (int ks = 0; ks < 8; ks++) bx[ks] = (4 * ks + g == ix) ? xg[ks] + d : xg[ks];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) bu[ks] = (4 * ks + g == iu) ? ug[ks] + d : ug[ks];
        double4_t acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
        for (int ks = 0; ks < 8; ks++)
#pragma unroll
          for (int ti = 0; ti < 2; ti++) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[ti][ks], bx[ks], acc[ti], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
          for (int ti = 0; ti < 2; ti++) acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(opB[ti][ks], bu[ks], acc[ti], 0, 0, 0);
        double* col = (var < nx) ? D + oFX + nx * var : D + oFU + nx * (var - nx);
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const double F = bx[4 * ti + r] + acc[ti][r] * v.dt;  // include/model.h:12-15
            const double other = dpp_swap1(F);                    // the point next door: e ^ 1
            const double val = (F - other) / (2 * kEps);
            const int row = 16 * ti + g + 4 * r;
            if (valid && !(e & 1) && row < nx) col[row] = val;
          }
      }
    } else
    for (int base = 0; base < E; base += 64) {
      const int e = base + lane;
      const bool valid = e < E;
      const int var = e >> 1;
      const double d = (e & 1) ? -kEps : kEps;
      double px[NX], pu[NU], F[NX];
      perturbed(var < nx, valid ? (var < nx ? var : var - nx) : -1, d, true, -1, 0.0, px, pu);
      integrate_dynamics(model, px, pu, v.dt, F);
      double* col = (var < nx) ? D + oFX + nx * var : D + oFU + nx * (var - nx);
#pragma unroll
      for (int r = 0; r < NX; r++) {
        const double fp = quad_bcast<0>(F[r]), fm = quad_bcast<1>(F[r]);      // lanes 4q, 4q+1
        const double fp2 = quad_bcast<2>(F[r]), fm2 = quad_bcast<3>(F[r]);    // lanes 4q+2, 4q+3
        const double val = ((lane & 2) ? (fp2 - fm2) : (fp - fm)) / (2 * kEps);
        if (valid && !(e & 1) && r < nx) col[r] = val;
      }
    }
  } else {
    for (int e = lane; e < nx * nx + nx * nu; e += 64) D[oFX + e] = 0.0;  // fx[T], fu[T] stay zero
    for (int e = lane; e < nu; e += 64) D[oCU + e] = 0.0;                 // :50-51
  }

  ILQR_DMARK(0)
  // ---- cost derivatives of a separable running cost (t < T): every distinct argument once ----
  if constexpr (M::kSeparableCost) {
    if (!last) {
      __shared__ double sx[2 * NX], su[2 * NU], s0[2];  // cost_x(x +- eps e_i), cost_u(u +- eps e_j), (cost_x(x), cost_u(u))
      // Matrix-core evaluation of x'Qx for models that expose Q (kQuadraticCostX): Y = Q P for 16
      // points at a time (v_mfma_f64_16x16x4_f64: two 16-row blocks of Q x eight k-steps), then
      // f_j = p_j . y_j.  Lane l = (kq = l >> 4, j = l & 15) supplies A = Q[16 ib + j][4 ks + kq]
      // (resident in registers for the whole kernel) and B = component 4 ks + kq of point j, built
      // from the knot and the point's perturbations; it receives rows kq + 4 r (+16) of y_j, i.e.
      // exactly the components it supplied.  Two tiles = four accumulator chains are in flight.
      typedef double double4_t __attribute__((ext_vector_type(4)));
      const int j16 = lane & 15, kq = lane >> 4;
      double qa[2][8], xb[8];
      if constexpr (M::kQuadraticCostX) {
        cmem_d* Qm = model.cost_x_matrix();
#pragma unroll
        for (int ib = 0; ib < 2; ib++)
#pragma unroll
          for (int ks = 0; ks < 8; ks++) qa[ib][ks] = ((const double*)Qm)[(16 * ib + j16) * NX + 4 * ks + kq];
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {  // x[4 ks + kq] without indexing the register array by kq
          const double a0 = x[4 * ks], a1 = x[4 * ks + 1], a2 = x[4 * ks + 2], a3 = x[4 * ks + 3];
          xb[ks] = (kq == 0) ? a0 : (kq == 1) ? a1 : (kq == 2) ? a2 : a3;
        }
      }
      // points (h = 0, 1): knot with component i1 += d1, then component i2 += d2 (index -1: none);
      // returns p'Qp of each in all four lanes of the point's column
      auto forms2 = [&](const int* i1, const double* d1, const int* i2, const double* d2, double* out) __attribute__((always_inline)) {
        double bv[2][8];
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
          for (int ks = 0; ks < 8; ks++) {
            const int comp = 4 * ks + kq;
            double val = xb[ks];
            val = (comp == i1[h]) ? val + d1[h] : val;
            val = (comp == i2[h]) ? val + d2[h] : val;
            bv[h][ks] = val;
          }
        double4_t y[2][2] = {{{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}};
#pragma unroll
        for (int ks = 0; ks < 8; ks++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            y[h][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[0][ks], bv[h][ks], y[h][0], 0, 0, 0);
            y[h][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[1][ks], bv[h][ks], y[h][1], 0, 0, 0);
          }
#pragma unroll
        for (int h = 0; h < 2; h++) {
          // y[h][0][r] = (Q p)[kq + 4 r], y[h][1][r] = (Q p)[16 + kq + 4 r]; matching components bv[r], bv[4 + r]
          double part = 0;
#pragma unroll
          for (int r = 0; r < 4; r++) part += bv[h][r] * y[h][0][r];
#pragma unroll
          for (int r = 0; r < 4; r++) part += bv[h][4 + r] * y[h][1][r];
          part += __shfl_xor(part, 16, 64);
          part += __shfl_xor(part, 32, 64);
          out[h] = part;
        }
      };
      {
        double px[NX], pu[NU];
        if constexpr (M::kQuadraticCostX && NX == 32) {
          // x singles and the knot itself: points e = 0 .. 2 nx (e = 2 nx: no perturbation)
          for (int base = 0; base <= 2 * nx; base += 32) {
            int i1[2], i2[2] = {-1, -1};
            double d1[2], d2[2] = {0.0, 0.0}, f[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const int e = base + 16 * h + j16;
              i1[h] = (e < 2 * nx) ? (e >> 1) : -1;
              d1[h] = (e & 1) ? -kEps : kEps;
            }
            forms2(i1, d1, i2, d2, f);
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const int e = base + 16 * h + j16;
              if (kq == 0 && e < 2 * nx) sx[e] = f[h];
              if (kq == 0 && e == 2 * nx) s0[0] = f[h];
            }
          }
        } else {
          // x singles: lane e -> x + (-1)^e eps e_{e/2}
          perturbed(true, lane < 2 * nx ? (lane >> 1) : -1, (lane & 1) ? -kEps : kEps, true, -1, 0.0, px, pu);
          const double qx = model.cost_x(px);
          if (lane < 2 * nx) sx[lane] = qx;
          if (2 * nx < 64) {               // wave-uniform
            if (lane == 2 * nx) s0[0] = qx;  // (no perturbation applied on this lane: cost_x(x))
          } else {                         // nx = 32: no spare lane, one more (uniform) evaluation
            double bx[NX];
#pragma unroll
            for (int c = 0; c < NX; c++) bx[c] = x[c];
            const double q0 = model.cost_x(bx);
            if (lane == 0) s0[0] = q0;
          }
        }
        // u singles on lanes 0..2nu-1, cost_u(u) on the next lane
        perturbed(false, lane < 2 * nu ? (lane >> 1) : -1, (lane & 1) ? -kEps : kEps, true, -1, 0.0, px, pu);
        const double qu = model.cost_u(pu);
        if (lane < 2 * nu) su[lane] = qu;
        if (lane == 2 * nu) s0[1] = qu;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): one wavefront, LDS operations complete in order
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      ILQR_DMARK(1)
      const double qx0 = s0[0], qu0 = s0[1];
      // cx, cu (derivatives.cpp:44-47)
      for (int i = lane; i < nx; i += 64)
        D[oCX + i] = (M::cost_from_parts(sx[2 * i], qu0) - M::cost_from_parts(sx[2 * i + 1], qu0)) / (2 * kEps);
      for (int j = lane; j < nu; j += 64)
        D[oCU + j] = (M::cost_from_parts(qx0, su[2 * j]) - M::cost_from_parts(qx0, su[2 * j + 1])) / (2 * kEps);
      // cxu (derivatives.cpp:114-144): c(px,pu) - c(mx,pu) - c(px,mu) + c(mx,mu)
      for (int p = lane; p < nx * nu; p += 64) {
        const int i = p / nu, j = p - i * nu;
        const double v4 = M::cost_from_parts(sx[2 * i], su[2 * j]) - M::cost_from_parts(sx[2 * i + 1], su[2 * j]) -
                          M::cost_from_parts(sx[2 * i], su[2 * j + 1]) + M::cost_from_parts(sx[2 * i + 1], su[2 * j + 1]);
        D[oCXU + i + nx * j] = v4 / (4 * kEps * kEps);
      }
      ILQR_DMARK(2)
      // cxx, cuu: every point of the two Hessians is a distinct argument (finite_diff.h:67-86)
      auto hessian = [&](auto on_x, auto per_lane, int n, int oH, double other) __attribute__((always_inline)) {
        constexpr bool X = decltype(on_x)::value;
        constexpr int P = decltype(per_lane)::value;  // evaluation points per lane and trip
        constexpr int NV = X ? NX : NU;
        const int npts = 2 * n * (n + 1);
        for (int base = 0; base < npts; base += 64 * P) {
          double f[P];
          int ii[P], jj[P], ee[P];
          double pa[P][NV];
#pragma unroll
          for (int h = 0; h < P; h++) {
            const int e = base + 64 * h + lane;
            ee[h] = e;
            int p = e >> 2, i = 0;
            if (e < npts) {
              while (p >= n - i) {
                p -= n - i;
                i++;
              }
            } else {
              p = 0;
            }
            ii[h] = (e < npts) ? i : -1;
            jj[h] = (e < npts) ? i + p : -1;
            const double d1 = (e & 1) ? -kEps : kEps, d2 = (e & 2) ? -kEps : kEps;
            // the point: base vector with the two perturbations applied one after the other
#pragma unroll
            for (int c = 0; c < NV; c++) {
              double val = X ? x[c < NX ? c : 0] : u[c < NU ? c : 0];
              val = (c == ii[h]) ? val + d1 : val;
              val = (c == jj[h]) ? val + d2 : val;
              pa[h][c] = val;
            }
          }
          if constexpr (X)
            model.template cost_xP<P>(pa, f);
          else
            model.template cost_uP<P>(pa, f);
#pragma unroll
          for (int h = 0; h < P; h++) {
            const double fv = X ? M::cost_from_parts(f[h], other) : M::cost_from_parts(other, f[h]);
            const double f0 = quad_bcast<0>(fv), f1 = quad_bcast<1>(fv), f2 = quad_bcast<2>(fv), f3 = quad_bcast<3>(fv);
            if (ee[h] < npts && (ee[h] & 3) == 0) {
              const double val = (f0 - f1 - f2 + f3) / (4 * kEps * kEps);
              D[oH + ii[h] + n * jj[h]] = val;
              D[oH + jj[h] + n * ii[h]] = val;
            }
          }
        }
      };
      if constexpr (M::kQuadraticCostX && NX == 32) {
        // cxx on the matrix cores: the 2 nx (nx + 1) points of the upper triangle, 4 sign
        // combinations per pair.  This lane's pair (i, i + rem) is advanced incrementally (a tile of
        // 16 points is 4 pairs).  Same points, same Q as the VALU route; the sums inside a form run
        // in the MFMA's order.
        const int npts = 2 * nx * (nx + 1);
        int pi = 0, prem = j16 >> 2;
        auto normalise = [&](int& i, int& rem) __attribute__((always_inline)) {
          while (i < nx && rem >= nx - i) {
            rem -= nx - i;
            i++;
          }
        };
        normalise(pi, prem);
        for (int base = 0; base < npts; base += 32) {
          int ti[2], tj[2], te[2];
          double d1[2], d2[2], f[2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int e = base + 16 * h + j16;
            te[h] = e;
            ti[h] = (e < npts) ? pi : -1;
            tj[h] = (e < npts) ? pi + prem : -1;
            d1[h] = (e & 1) ? -kEps : kEps;
            d2[h] = (e & 2) ? -kEps : kEps;
            prem += 4;  // the next tile's pair
            normalise(pi, prem);
          }
          forms2(ti, d1, tj, d2, f);
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const double fv = M::cost_from_parts(f[h], qu0);
            const double f0 = quad_bcast<0>(fv), f1 = quad_bcast<1>(fv), f2 = quad_bcast<2>(fv), f3 = quad_bcast<3>(fv);
            if (te[h] < npts && (te[h] & 3) == 0 && kq == 0) {
              const double val = (f0 - f1 - f2 + f3) / (4 * kEps * kEps);
              D[oCXX + ti[h] + nx * tj[h]] = val;
              D[oCXX + tj[h] + nx * ti[h]] = val;
            }
          }
        }
      } else {
        hessian(std::true_type{}, std::integral_constant<int, ILQR_FD_POINTS_X>{}, nx, oCXX, qu0);
      }
      ILQR_DMARK(3)
      hessian(std::false_type{}, std::integral_constant<int, 2>{}, nu, oCUU, qx0);
      ILQR_DMARK(4)
      return;
    }
  }

  // ---- scalar-valued evaluations: one list, groups aligned to quads ----
  const int n_cx = 2 * nx, n_cu = last ? 0 : 2 * nu;
  const int n_cxx = 2 * nx * (nx + 1), n_cuu = 2 * nu * (nu + 1), n_cxu = 4 * nx * nu;
  const int g_cu = (n_cx + 3) & ~3, g_cxx = g_cu + ((n_cu + 3) & ~3), g_cuu = g_cxx + n_cxx, g_cxu = g_cuu + n_cuu,
            total = g_cxu + n_cxu;
  struct Point {
    int cat, i1, i2;
    bool t1x, t2x;
    double d1, d2;
  };
  auto decode = [&](int e) __attribute__((always_inline)) {
    Point q;
    q.cat = -1;
    q.i1 = q.i2 = -1;
    q.t1x = q.t2x = true;
    q.d1 = q.d2 = 0;
    if (e < n_cx) {
      q.cat = 0;
      q.i1 = e >> 1;
      q.d1 = (e & 1) ? -kEps : kEps;
    } else if (e >= g_cu && e < g_cu + n_cu) {
      q.cat = 1;
      q.t1x = false;
      q.i1 = (e - g_cu) >> 1;
      q.d1 = (e & 1) ? -kEps : kEps;
    } else if (e >= g_cxx && e < total) {
      int n, p;
      if (e < g_cuu) {
        q.cat = 2;
        n = nx;
        p = (e - g_cxx) >> 2;
      } else if (e < g_cxu) {
        q.cat = 3;
        n = nu;
        p = (e - g_cuu) >> 2;
        q.t1x = q.t2x = false;
      } else {
        q.cat = 4;
        n = 0;
        p = (e - g_cxu) >> 2;
        q.t2x = false;
      }
      if (q.cat == 4) {  // (i, j) row-major over nx x nu, as the loops of derivatives.cpp:117-118
        q.i1 = p / nu;
        q.i2 = p - q.i1 * nu;
      } else {  // upper triangle, row by row (finite_diff.h:70-71)
        int i = 0;
        while (p >= n - i) {
          p -= n - i;
          i++;
        }
        q.i1 = i;
        q.i2 = i + p;
      }
      const int combo = e & 3;  // 0: pp  1: mp  2: pm  3: mm   (first perturbation's sign changes fastest)
      q.d1 = (combo & 1) ? -kEps : kEps;
      q.d2 = (combo & 2) ? -kEps : kEps;
    }
    return q;
  };
  // lanes of a quad combine their values and the lane of the first point stores the entry
  auto combine = [&](int e, const Point& q, double f) __attribute__((always_inline)) {
    const double f0 = quad_bcast<0>(f), f1 = quad_bcast<1>(f), f2 = quad_bcast<2>(f), f3 = quad_bcast<3>(f);
    if (q.cat == 0 || q.cat == 1) {
      if (!(e & 1)) {
        const double g = (((lane & 2) ? f2 : f0) - ((lane & 2) ? f3 : f1)) / (2 * kEps);
        D[(q.cat == 0 ? oCX : oCU) + q.i1] = g;
      }
    } else if (q.cat >= 2 && (e & 3) == 0) {
      // (at t = T the cxu lanes hold final_cost(px), (mx), (px), (mx): the expression of :140)
      const double val = (f0 - f1 - f2 + f3) / (4 * kEps * kEps);
      if (q.cat == 2) {
        D[oCXX + q.i1 + nx * q.i2] = val;
        D[oCXX + q.i2 + nx * q.i1] = val;
      } else if (q.cat == 3) {
        D[oCUU + q.i1 + nu * q.i2] = val;
        D[oCUU + q.i2 + nu * q.i1] = val;
      } else {
        D[oCXU + q.i1 + nx * q.i2] = val;
      }
    }
  };
  // two points per lane and trip (model.cost2: every model constant fetched once serves both)
  for (int base = 0; base < total; base += 128) {
    const int ea = base + lane, eb = base + 64 + lane;
    const Point qa = decode(ea), qb = decode(eb);
    double pxa[NX], pua[NU], pxb[NX], pub[NU];
    perturbed(qa.t1x, qa.i1, qa.d1, qa.t2x, qa.i2, qa.d2, pxa, pua);
    perturbed(qb.t1x, qb.i1, qb.d1, qb.t2x, qb.i2, qb.d2, pxb, pub);
    double fa = 0, fb = 0;
    if (!last) {
      model.cost2(pxa, pua, pxb, pub, fa, fb);
    } else {
      // final_cost for the x-derivatives at t = T (:49, :92, :140); cuu[T] is cost(x_T, .)
      if (qa.cat >= 0) fa = (qa.cat != 3) ? model.final_cost(pxa) : model.cost(pxa, pua);
      if (qb.cat >= 0) fb = (qb.cat != 3) ? model.final_cost(pxb) : model.cost(pxb, pub);
    }
    combine(ea, qa, fa);
    combine(eb, qb, fb);
  }
}

// ------------------------------------------------------------------------------------------
// finite differences of the LQ model, every point evaluated by what moved
// ------------------------------------------------------------------------------------------
// k_derivatives_g evaluates every perturbed point of the cost Hessians DENSELY: Y = Q P on the matrix cores for 16 points at a time,
// 2 177 + 577 matrix-vector products per knot, 88 % of a finite-difference iteration of configs[4] (and fp64 matrix instructions run
// at the VALU's rate on gfx950: scripts/ubench/coissue.hip).  But a perturbed point differs from the knot in ONE or TWO components, and
// the model has said that its cost is a quadratic form (kQuadraticCostX, cost_u_matrix): with y = Q x and z = Q'x formed once per knot,
//     (Q p)_i = y_i + delta_i Q[i][i] + delta_j Q[i][j],      x . (Q p) = x . y + delta_i z_i + delta_j z_j,
//     p'Q p = x . (Q p) + delta_i (Q p)_i + delta_j (Q p)_j
// -- the same function value at the same point (delta = the perturbation as it was actually applied, fl(fl(x_i + d1) + d2) - x_i on the
// diagonal, finite_diff.h:67-86), from a dozen multiply-adds instead of n^2; the differences of these values are then taken exactly
// as before.  The values agree with the dense evaluation to rounding (1e-16 of |x'Qx|, which the second difference amplifies by
// 1 / 4 eps^2 = 2.5e5 like every other rounding of f), the records with the reference's to the tolerance they already had.
// One wavefront sweeps kLqKnotsPerWave knots of a trajectory (Q, R in LDS, padded so that rows and columns both read without bank
// conflicts: loaded once per wavefront instead of once per knot).  A point is ONE lane's work: the 2 n singles in one pass, the
// Hessians' upper triangles as 8 x 8 blocks of pairs (lane = (i & 7) + 8 (j & 7): both mirror images of an entry leave as 64-byte
// runs), the four sign combinations of a pair on its lane.  The Jacobian sweep: A x + B u once per knot on the matrix cores, a point's
// image from it and one column.  Three wavefronts per SIMD (12.5 KB of LDS, <= 168 registers).
// Knot T (final_cost, the reference's conventions there) is k_derivatives_g's (t_only).  ILQR_ROUTE_LQ_DENSE_FD keeps the dense sweep for every knot (cross-check: tests/test_gpu_lq_end_to_end.py).
constexpr int kLqKnotsPerWave = 8;
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_derivatives_lq(BatchView v, LqModel model, int force) {
  static_assert(GN == 32 && GM == 16, "operand blocks below are written for a 32 x 16 model");
  typedef double double4_t __attribute__((ext_vector_type(4)));
  const int nx = model.nx, nu = model.nu, T = v.T;
  const int lane = threadIdx.x, g = lane >> 4, p = lane & 15;
  const int nchunk = (T + kLqKnotsPerWave - 1) / kLqKnotsPerWave;  // knots 0 .. T-1
  const int b = blockIdx.x / nchunk, t0 = (blockIdx.x - b * nchunk) * kLqKnotsPerWave;
  if (blockIdx.x == 0 && lane == 0) *v.n_running = 0;  // k_accept of this iteration recounts
  if (!(force || (v.status[b] == 0 && v.flg_change[b]))) return;
  const int oFX = 0, oFU = oFX + nx * nx, oCX = oFU + nx * nu, oCXX = oCX + nx, oCXU = oCXX + nx * nx, oCU = oCXU + nx * nu,
            oCUU = oCU + nu, REC = oCUU + nu * nu;
  constexpr int LQ = GN + 1, LR = GM + 1;  // Qc[col * LQ + row] = Q[row][col]: a column across lanes is contiguous, a row has stride 33 doubles -- neither conflicts
  __shared__ double Qc[GN * LQ], Rc[GM * LR];
  __shared__ double xk[GN], uk[GM], y0s[GN], zxs[GN], yu0s[GM], zus[GM], sx[2 * GN], su[2 * GM];
  auto sync = []() __attribute__((always_inline)) {  // one wavefront: the LDS executes its operations in order; only the compiler must not reorder them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  for (int e = lane; e < GN * GN; e += 64) Qc[(e & 31) * LQ + (e >> 5)] = model.Q[e];   // e = row * GN + col
  for (int e = lane; e < GM * GM; e += 64) Rc[(e & 15) * LR + (e >> 4)] = model.R[e];
  // (A, B as matrix-core A operands are fetched where a knot's dense product uses them, not kept: the registers they would hold for
  //  the whole wavefront are what a third wavefront per SIMD needs)

  for (int t = t0; t < t0 + kLqKnotsPerWave && t < T; t++) {
    double* __restrict__ D = v.D + ((size_t)b * (T + 1) + t) * REC;
    sync();  // (the previous knot's readers of xk .. su are through)
    {
      const double xv = (lane < nx) ? v.xs[((size_t)b * (T + 1) + t) * nx + lane] : 0.0;
      const double uv = (lane < nu) ? v.us[((size_t)b * T + t) * nu + lane] : 0.0;
      if (lane < GN) xk[lane] = xv;
      if (lane < GM) uk[lane] = uv;
    }
    sync();
    double xb[8], ub[4];  // this lane's components 4 ks + g of the knot
#pragma unroll
    for (int ks = 0; ks < 8; ks++) xb[ks] = xk[4 * ks + g];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) ub[ks] = uk[4 * ks + g];

    // ---- fx, fu (finite_diff.h:35-47): F(p) = p + dt (A p_x + B p_u).  dx0 = A x + B u once per knot on the matrix cores; a point that
    // differs from the knot in one component by delta has A p_x + B p_u = dx0 + delta A[:, i] (or delta B[:, j]): rows g + 4 r (+ 16) of 16
    // points per pass, the column read from the model's matrix (L2-resident: 8 KB) ----
    {
      double4_t acc0[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
      {
        double opA[2][8], opB[2][4];
#pragma unroll
        for (int ti = 0; ti < 2; ti++) {
#pragma unroll
          for (int ks = 0; ks < 8; ks++) opA[ti][ks] = model.A[(16 * ti + p) * GN + 4 * ks + g];
#pragma unroll
          for (int ks = 0; ks < 4; ks++) opB[ti][ks] = model.Bm[(16 * ti + p) * GM + 4 * ks + g];
        }
#pragma unroll
        for (int ks = 0; ks < 8; ks++)
#pragma unroll
          for (int ti = 0; ti < 2; ti++) acc0[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(opA[ti][ks], xb[ks], acc0[ti], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
#pragma unroll
          for (int ti = 0; ti < 2; ti++) acc0[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(opB[ti][ks], ub[ks], acc0[ti], 0, 0, 0);
      }
      constexpr double inv2e = 1.0 / (2 * kEps);  // (the quotient of finite_diff.h:44 as a product, as derivatives.hpp does)
      const int E = 2 * (nx + nu);
      for (int base = 0; base < E; base += 16) {
        const int e = base + p;
        const bool valid = e < E;
        const int var = valid ? (e >> 1) : 0;
        const bool isx = var < nx;
        const int idx = isx ? var : var - nx;
        const double z = isx ? xk[idx] : uk[idx];
        const double pz = z + ((e & 1) ? -kEps : kEps);
        const double del = valid ? pz - z : 0.0;
        const double* __restrict__ colp = isx ? model.A + idx : model.Bm + idx;  // element (row, idx): + row * GN or row * GM
        const int ld = isx ? GN : GM;
        double cv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) cv[k] = colp[(16 * (k >> 2) + 4 * (k & 3) + g) * ld];
        double* col = isx ? D + oFX + nx * idx : D + oFU + nx * idx;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const int row = 16 * (k >> 2) + 4 * (k & 3) + g;
          const double dxr = __builtin_fma(del, cv[k], acc0[k >> 2][k & 3]);
          const double pr = (isx && row == idx) ? pz : xb[k];
          const double F = pr + dxr * v.dt;          // include/model.h:12-15
          const double other = dpp_swap1(F);          // the point next door: e ^ 1
          const double val = (F - other) * inv2e;
          if (valid && !(e & 1) && row < nx) col[row] = val;
        }
      }
    }

    // ---- y = Q x (lanes 0..31: one row each), z = Q'x (lanes 32..63: one column each); R u, R'u likewise on lanes 0..15, 16..31 ----
    {
      const int r = lane & 31;
      const bool tr = lane >= 32;
      const double* qrow = Qc + (tr ? r * LQ : r);
      const int step = tr ? 1 : LQ;
      double acc = 0;
#pragma unroll 8
      for (int c = 0; c < GN; c++) acc = __builtin_fma(qrow[c * step], xk[c], acc);
      (tr ? zxs : y0s)[r] = acc;
      const int ru = lane & 15;
      const bool tu = (lane & 16) != 0;
      const double* rrow = Rc + (tu ? ru * LR : ru);
      const int stepu = tu ? 1 : LR;
      double accu = 0;
#pragma unroll 8
      for (int c = 0; c < GM; c++) accu = __builtin_fma(rrow[c * stepu], uk[c], accu);
      if (lane < 32) (tu ? zus : yu0s)[ru] = accu;
    }
    sync();
    double qx0, qu0;  // cost_x(x), cost_u(u)
    {
      double part = (lane < 32) ? xk[lane & 31] * y0s[lane & 31] : ((lane < 48) ? uk[lane & 15] * yu0s[lane & 15] : 0.0);
#pragma unroll
      for (int sh = 1; sh < 32; sh <<= 1) part += __shfl_xor(part, sh, 64);
      qx0 = __shfl(part, 0, 64);
      qu0 = __shfl(part, 32, 64);
    }

    // ---- single perturbations: cost_x(x +- eps e_i) on lane 2 i (+ 1), cost_u(u +- eps e_j) ----
    {
      const int i = lane >> 1;
      const double d = (lane & 1) ? -kEps : kEps;
      if (i < nx) {
        const double x1 = xk[i];
        const double del = (x1 + d) - x1;
        const double Y = __builtin_fma(del, Qc[i * LQ + i], y0s[i]);              // (Q p)[i]
        sx[lane] = __builtin_fma(del, Y, __builtin_fma(del, zxs[i], qx0));
      }
      if (i < nu) {
        const double u1 = uk[i];
        const double del = (u1 + d) - u1;
        const double Y = __builtin_fma(del, Rc[i * LR + i], yu0s[i]);
        su[lane] = __builtin_fma(del, Y, __builtin_fma(del, zus[i], qu0));
      }
    }
    sync();
    // cx, cu (derivatives.cpp:44-47)
    for (int i = lane; i < nx; i += 64)
      D[oCX + i] = (LqModel::cost_from_parts(sx[2 * i], qu0) - LqModel::cost_from_parts(sx[2 * i + 1], qu0)) / (2 * kEps);
    for (int j = lane; j < nu; j += 64)
      D[oCU + j] = (LqModel::cost_from_parts(qx0, su[2 * j]) - LqModel::cost_from_parts(qx0, su[2 * j + 1])) / (2 * kEps);
    // cxu (derivatives.cpp:114-144): c(px,pu) - c(mx,pu) - c(px,mu) + c(mx,mu); consecutive lanes write consecutive rows of a column
    for (int q = lane; q < nx * nu; q += 64) {
      const int j = q / nx, i = q - j * nx;
      const double v4 = LqModel::cost_from_parts(sx[2 * i], su[2 * j]) - LqModel::cost_from_parts(sx[2 * i + 1], su[2 * j]) -
                        LqModel::cost_from_parts(sx[2 * i], su[2 * j + 1]) + LqModel::cost_from_parts(sx[2 * i + 1], su[2 * j + 1]);
      D[oCXU + i + nx * j] = v4 / (4 * kEps * kEps);
    }
    // ---- cxx, cuu: the upper triangle, four sign combinations per pair (finite_diff.h:67-86): pp, mp, pm, mm on the pair's lane ----
    auto hessian = [&](auto on_x, int n, int oH, double other, double q0) __attribute__((always_inline)) {
      constexpr bool X = decltype(on_x)::value;
      constexpr int LD = X ? LQ : LR;
      const double* Mc = X ? Qc : Rc;
      const double* zk = X ? xk : uk;
      const double* yk = X ? y0s : yu0s;
      const double* zz = X ? zxs : zus;
      constexpr double inv4e2 = 1.0 / (4 * kEps * kEps);  // (the quotient of finite_diff.h:84 as a product: <= 1 ulp of an entry, as derivatives.hpp does)
      const int nb = (n + 7) >> 3;
      for (int bi = 0; bi < nb; bi++)
        for (int bj = bi; bj < nb; bj++) {
          const int i = 8 * bi + (lane & 7), j = 8 * bj + (lane >> 3);
          const bool valid = (i <= j) & (j < n);
          const int ii = valid ? i : 0, jj = valid ? j : 0;
          const bool diag = ii == jj;
          const double x1 = zk[ii], x2 = zk[jj];
          // the perturbations as they are applied: +-eps on component i, then +-eps on component j (the same component on the diagonal)
          const double p1p = x1 + kEps, p1m = x1 - kEps;
          const double Mii = Mc[ii * LD + ii], Mij = Mc[jj * LD + ii], Mji = Mc[ii * LD + jj], Mjj = Mc[jj * LD + jj];
          const double yi = yk[ii], yj = yk[jj], zi = zz[ii], zj = zz[jj];
          double fv[4];
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const double first = (c & 1) ? p1m : p1p, d2 = (c & 2) ? -kEps : kEps;
            const double del1 = (diag ? first + d2 : first) - x1;
            const double del2 = diag ? 0.0 : (x2 + d2) - x2;
            const double Y1 = __builtin_fma(del2, Mij, __builtin_fma(del1, Mii, yi));   // (M p)[i]
            const double Y2 = __builtin_fma(del2, Mjj, __builtin_fma(del1, Mji, yj));   // (M p)[j]
            const double xy = __builtin_fma(del2, zj, __builtin_fma(del1, zi, q0));      // z . (M p)
            const double f = __builtin_fma(del2, Y2, __builtin_fma(del1, Y1, xy));
            fv[c] = X ? LqModel::cost_from_parts(f, other) : LqModel::cost_from_parts(other, f);
          }
          if (valid) {
            const double val = (fv[0] - fv[1] - fv[2] + fv[3]) * inv4e2;
            D[oH + ii + n * jj] = val;
            D[oH + jj + n * ii] = val;
          }
        }
    };
    hessian(std::true_type{}, nx, oCXX, qu0, qx0);
    hessian(std::false_type{}, nu, oCUU, qx0, qu0);
  }
}

}  // namespace ilqr
