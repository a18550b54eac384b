"""Which line-search alpha is accepted, per trajectory and per 16-trajectory tile, on the bench workload (iterations 1..24)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0
B, T = 4096, 499
g = BatchILQR("acrobot", B, T, 0.02, u_min=-1.5, u_max=1.5, flags=capi.FLAG_FIXED_WORK, params=dict(max_iter=200))
g.init_traj(acrobot_x0(B), np.zeros((B, T, 1)))
for it in range(1, 41):
    g.iterate(1)
    al = g.status()[2]
    a = np.where(al < 0, 11, al)  # no step: all 11 needed
    tile_max = a.reshape(-1, 16).max(axis=1)
    if it in (1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 32, 40):
        print("it %2d: mean alpha idx %.2f, P(traj<=3) %.2f P(traj<=7) %.2f | tiles: P(max<=3) %.2f P(max<=7) %.2f  none-accepted %.3f" % (
            it, a.mean(), (a <= 3).mean(), (a <= 7).mean(), (tile_max <= 3).mean(), (tile_max <= 7).mean(), (al < 0).mean()))
