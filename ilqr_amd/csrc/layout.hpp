// layout.hpp -- canonical [B][S][E] <-> tiled [tile][S][E][16] conversion, slot permutation (compaction), per-trajectory
// state reset (init_traj, src/ilqr_core.cpp:11-56).  Lane mapping everywhere: consecutive lanes = consecutive trajectories of
// a tile, so each vector load / store touches whole 128-byte lines of the tiled layout (common.hpp).
#pragma once
#include <type_traits>

#include "boxqp.hpp"
#include "common.hpp"
#include "models.hpp"

namespace ilqr {

// ------------------------------------------------------------------------------------------
// layout conversion
// ------------------------------------------------------------------------------------------
// canonical src[b][s][e]  ->  tiled dst[tile][s][e][l]      (one thread per tiled element)
template <class real>
__global__ void k_pack(const double* __restrict__ src, real* __restrict__ dst, int B, int ntiles, int S, int E) {
  const size_t n = (size_t)ntiles * S * E * TW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i % TW);
    size_t r = i / TW;
    const int e = (int)(r % E);
    r /= E;
    const int s = (int)(r % S);
    const int tile = (int)(r / S);
    const int b = tile * TW + l;
    dst[i] = (b < B) ? (real)src[((size_t)b * S + s) * E + e] : real(0);
  }
}
// tiled src -> canonical dst   (one thread per canonical element; reads are line-strided but
// this path only serves getters)
template <class real>
__global__ void k_unpack(const real* __restrict__ src, double* __restrict__ dst, int B, int S, int E) {
  const size_t n = (size_t)B * S * E;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E);
    size_t r = i / E;
    const int s = (int)(r % S);
    const int b = (int)(r / S);
    dst[i] = (double)src[tidx(b / TW, s, e, b % TW, S, E)];
  }
}
// tiled record sub-range [off, off+E) of a record of size REC  <->  canonical [B][S][E]
template <class real>
__global__ void k_pack_rec(const double* __restrict__ src, real* __restrict__ dst, int B, int ntiles, int S,
                           int REC, int off, int E) {
  const size_t n = (size_t)ntiles * S * E * TW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i % TW);
    size_t r = i / TW;
    const int e = (int)(r % E);
    r /= E;
    const int s = (int)(r % S);
    const int tile = (int)(r / S);
    const int b = tile * TW + l;
    dst[didx(tile, s, off + e, l, S, REC)] = (b < B) ? (real)src[((size_t)b * S + s) * E + e] : real(0);
  }
}
template <class real>
__global__ void k_unpack_rec(const real* __restrict__ src, double* __restrict__ dst, int B, int S, int REC,
                             int off, int E) {
  const size_t n = (size_t)B * S * E;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E);
    size_t r = i / E;
    const int s = (int)(r % S);
    const int b = (int)(r / S);
    dst[i] = (double)src[didx(b / TW, s, off + e, b % TW, S, REC)];
  }
}

// dst slot j <- src slot perm[j], for every slot of the padded batch (compaction of running trajectories between chunks
// of a full solve, capi.hip): tiled arrays [tile][S][E][16] and per-trajectory scalars
template <class real>
__global__ void k_permute_tiled(const real* __restrict__ src, real* __restrict__ dst, const int* __restrict__ perm, int ntiles, int S, int E) {
  const size_t n = (size_t)ntiles * S * E * TW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(i % TW);
    size_t r = i / TW;
    const int e = (int)(r % E);
    r /= E;
    const int s = (int)(r % S);
    const int tile = (int)(r / S);
    const int p = perm[tile * TW + l];
    dst[i] = src[tidx(p / TW, s, e, p % TW, S, E)];
  }
}
template <class T>
__global__ void k_permute_scalar(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ perm, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) dst[j] = src[perm[j]];
}

// ------------------------------------------------------------------------------------------
// per-trajectory state reset (init_traj, ilqr_core.cpp:11-56; statics of ilqr.h:17-18)
// ------------------------------------------------------------------------------------------
template <class real>
__global__ void k_reset_state(BatchViewT<real> v, double lambda0, double dlambda0) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= v.Bp) return;
  v.lambda[b] = lambda0;
  v.dlambda[b] = dlambda0;
  v.dV[b] = 0;
  v.dV[v.Bp + b] = 0;
  v.gnorm[b] = 0;
  v.status[b] = (b < v.B) ? 0 : 4;  // padding lanes never run
  v.iters[b] = 0;
  v.flg_change[b] = 1;
  v.alpha_idx[b] = -1;
  v.diverge[b] = 0;
  v.backpass_done[b] = 0;
}

}  // namespace ilqr
