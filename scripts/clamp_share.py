"""How often is a control clamped in the bench workload?  Per iteration of the headline workload (acrobot T=499 B=4096 +-1.5, fixed work):
the share of knots whose feedback row K is zero (box-QP left through 'all clamped', boxqp.cpp:74-77), per trajectory and per group of four
consecutive trajectories (one matrix-core chain wavefront of k_solve_hex).    gpurun -- 'python scripts/clamp_share.py [iters] [limit]'"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import BatchILQR, capi  # noqa: E402
from tests.util import acrobot_x0  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 23
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
B, T = 4096, 499
g = BatchILQR("acrobot", B, T, 0.01, u_min=-lim, u_max=lim, flags=capi.FLAG_FIXED_WORK, params=dict(max_iter=iters + 2))
g.init_traj(acrobot_x0(B), np.zeros((B, T, 1)))
for it in range(iters):
    g.iterate(1)
    k, K = g.gains()
    z = np.all(K.reshape(B, T, -1) == 0, axis=2)          # [B][T] clamped knots
    z4 = z.reshape(B // 4, 4, T).all(axis=1)
    print("iteration %2d: clamped knots %.3f of all; all four trajectories of a chain wavefront clamped %.3f; lambda median %.3g" %
          (it, z.mean(), z4.mean(), float(np.median(g.lambdas()[0]))))
