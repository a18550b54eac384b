// common.hpp -- shared device/host definitions of the MI355X-native batched iLQR engine.
//
// Device storage ("tiled" layout, private to a handle): trajectories are grouped in tiles of
// TW = 16 consecutive batch indices; an array with S slots (time steps) of E doubles per
// trajectory is stored as  [tile][S][E][TW]  -- the innermost 16 doubles are one 128-byte
// line holding the same element of 16 neighbouring trajectories.  Every kernel maps
// consecutive lanes to consecutive trajectories of a tile, so each vector memory instruction
// touches whole 128-byte lines, and one tile's whole time series is one contiguous stream.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ilqr {

constexpr int TW = 16;       // trajectories per tile (16 x 8 B = one 128-B line)
constexpr int NALPHA = 11;   // include/ilqr.h:24
constexpr int MAXN = 32;
constexpr int MAXM = 16;

// include/ilqr.h:24 -- the rounded literals, not 10^linspace(0,-3,11)
__device__ __constant__ const double kAlpha[NALPHA] = {1.0000, 0.5012, 0.2512, 0.1259, 0.0631, 0.0316,
                                                       0.0158, 0.0079, 0.0040, 0.0020, 0.0010};
static const double kAlphaHost[NALPHA] = {1.0000, 0.5012, 0.2512, 0.1259, 0.0631, 0.0316,
                                          0.0158, 0.0079, 0.0040, 0.0020, 0.0010};

// include/finite_diff.h:9, src/derivatives.cpp:10
constexpr double kEps = 1e-3;

// include/boxqp.h:19-24,63
constexpr int kQpMaxIter = 100;
constexpr double kMinGrad = 1e-8;
constexpr double kMinRelImprove = 1e-8;
constexpr double kStepDec = 0.6;
constexpr double kMinStep = 1e-22;
constexpr double kArmijo = 0.1;
constexpr double kClampTol = 1e-4;

// Line-search candidates are stored as CHECKPOINTS: every control u_t (the rollout's own output)
// but the state only at every CT-th knot.  Whoever needs knot t of an accepted candidate (the
// derivative sweep, the commit copy, the getter) re-integrates at most CT-1 Euler steps from the
// checkpoint with the same device function the rollout used.  11 full candidate trajectories
// per iteration were 909 MB of stores and bound the rollout kernel; checkpoints are 273 MB.
//   cand_u  [NALPHA][tile][T][nu][TW]       (tidx with tile' = alpha*ntiles + tile)
//   cand_x  [NALPHA][tile][NCH][nx][TW]     NCH = T/CT + 1, entry c = state at knot c*CT
constexpr int CT = 8;

__host__ __device__ inline size_t tidx(int tile, int s, int e, int l, int S, int E) {
  return (((size_t)tile * S + s) * E + e) * TW + l;
}

// Element offsets inside one derivative record (all matrices column-major).  The order puts the
// two odd-sized blocks (cu, cuu; together nu(nu+1) doubles) last, so every block starts at an even
// offset and the record can be stored as PAIRS of doubles.
template <int NX, int NU>
struct Rec {
  static constexpr int FX = 0;
  static constexpr int FU = FX + NX * NX;
  static constexpr int CX = FU + NX * NU;
  static constexpr int CXX = CX + NX;
  static constexpr int CXU = CXX + NX * NX;
  static constexpr int CU = CXU + NX * NU;
  static constexpr int CUU = CU + NU;
  static constexpr int SIZE = CUU + NU * NU;
  static_assert(NX % 2 == 0, "pair layout assumes an even state dimension");
  static_assert(SIZE % 2 == 0, "record must be a whole number of pairs");
};
inline int rec_size(int nx, int nu) { return 2 * nx * nx + 2 * nx * nu + nx + nu + nu * nu; }
// runtime offsets in the ABI's order fx, fu, cx, cu, cxx, cxu, cuu
inline void rec_offsets(int nx, int nu, int off[7], int len[7]) {
  const int FX = 0, FU = FX + nx * nx, CX = FU + nx * nu, CXX = CX + nx, CXU = CXX + nx * nx, CU = CXU + nx * nu, CUU = CU + nu;
  const int o[7] = {FX, FU, CX, CU, CXX, CXU, CUU};
  const int n[7] = {nx * nx, nx * nu, nx, nu, nx * nx, nx * nu, nu * nu};
  for (int i = 0; i < 7; i++) {
    off[i] = o[i];
    len[i] = n[i];
  }
}
// f(std::integral_constant<int, 0>()), ..., f(std::integral_constant<int, N - 1>()): an unrolled loop whose index is a compile-time value
template <int N, int I = 0, class F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<N, I + 1>(f);
  }
}
// Derivative records are stored pair-interleaved: [tile][knot][REC/2][TW][2] -- element e of lane
// l sits next to element e^1 of the same trajectory, so one 16-byte access per lane moves two
// elements (the backward kernel needs 37 elements per lane and step: 18 loads instead of 37).
__host__ __device__ inline size_t didx(int tile, int t, int e, int l, int T1, int REC) {
  return ((((size_t)tile * T1 + t) * (REC / 2) + (e >> 1)) * TW + l) * 2 + (e & 1);
}

// Solver tunables (include/ilqr.h:14-24), passed by value to kernels.
struct SolverParams {
  int max_iter;
  double tol_fun, tol_grad, lambda_factor, lambda_max, lambda_min, z_min;
  int fixed_work;
  int fixes;  // bit 0 clamped rollout, bit 1 a failed Cholesky ends the box-QP (ILQR_FLAG_REFERENCE_FIXES); bit 2 lambda on Vxx (ILQR_FLAG_REGULARIZE_VXX)
};

// Raw views of a batch's device state, passed by value to kernels.
// `real` is the handle's arithmetic (ilqr_desc.dtype): double -- the reference's -- or float (BASELINE.json
// configs[3]).  Everything per knot (states, controls, gains, derivative records, line-search candidates)
// is stored and computed in `real`; the per-trajectory scalars that are accumulated over the horizon or
// carried across iterations (costs, dV, gradient norm, lambda / dlambda) stay double in both modes -- a
// float sum of 499 stage costs would not resolve the cost CHANGES the line search and tolFun test on.
template <class real>
struct BatchViewT {
  int B, Bp, ntiles, T;
  double dt;
  // trajectories
  real* x0;   // [tile][1][nx][TW]
  real* xs;   // [tile][T+1][nx][TW]
  real* us;   // [tile][T][nu][TW]
  real* kff;  // [tile][T][nu][TW]
  real* Kfb;  // [tile][T][nu*nx][TW]
  real* D;    // derivative records, pair-interleaved [tile][T+1][REC/2][TW][2]  (didx)
  real* cand_u; // line-search candidates: every control      [NALPHA][tile][T][nu][TW]
  real* cand_x; //                         checkpoint states  [NALPHA][tile][NCH][nx][TW]
  int nch;        // NCH = T/CT + 1
  double* cost_c; // [NALPHA][Bp]
  // per-trajectory scalars [Bp]
  double* cost;
  double* lambda;
  double* dlambda;
  double* dV;     // [2][Bp]
  double* gnorm;
  int* status;    // ilqr_traj_status
  int* iters;
  int* flg_change;
  int* alpha_idx;     // accepted alpha of the last line search, -1 = none
  int* diverge;       // return value of the last backward_pass()
  int* backpass_done; // ilqr_core.cpp:136
  int* n_running;     // [1] device counter
  int analytic;       // ILQR_FLAG_ANALYTIC_DERIVATIVES: the models' exact derivatives instead of finite differences
};
using BatchView = BatchViewT<double>;  // (the generic nx <= 32 path, generic.hpp / backward_wave.hpp, is fp64 only)

}  // namespace ilqr
