// kernels.hpp -- the HIP kernels of the nx = 4 tiled path (gfx950 / MI355X), one header per stage:
//
//   layout.hpp           k_pack / k_unpack / k_permute_* / k_reset_state
//   rollout.hpp          rollout_tile, k_rollout, accept_one, k_accept, candidate checkpoints
//   derivatives.hpp      derivatives_of_knot, k_derivatives
//   backward_thread.hpp  k_backward_t (one thread per trajectory; cross-check)
//   backward_quad.hpp    backward_quad, k_backward_q (four lanes per trajectory)
//   solve_tile.hpp       the LDS ring, k_sweep_backward, k_solve_tile (persistent tiles), k_commit
//   backward_hex.hpp     k_solve_hex: one tile per CU, its backward pass as four matrix-core chains (the metric batch)
//   (kernels_wide.hpp    64-trajectory wide tiles for saturating batches; generic.hpp, backward_wave*.hpp: nx <= 32)
#pragma once
#include "backward_quad.hpp"
#include "backward_thread.hpp"
#include "derivatives.hpp"
#include "layout.hpp"
#include "rollout.hpp"
#include "solve_tile.hpp"
#include "backward_hex.hpp"
