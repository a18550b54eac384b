import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0
B, T = 4096, 499
for lim in (1.5, 5.0):
    g = BatchILQR("acrobot", B, T, 0.02, u_min=-lim, u_max=lim, flags=capi.FLAG_FIXED_WORK)
    g.init_traj(acrobot_x0(B), np.zeros((B, T, 1)))
    print("limit", lim)
    for it in range(14):
        g.iterate(1)
        st, iters, al = g.status()
        h = np.bincount(al + 1, minlength=12)
        print(it, "none:", h[0], "alpha idx hist:", h[1:].tolist(), "mean cost %.4g" % g.cost().mean())
