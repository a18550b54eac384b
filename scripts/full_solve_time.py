import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0
B, T = 4096, 499
for route, name in ((0, "hex"), (capi.ROUTE_QUAD_CHAIN, "quad"), (0, "hex"), (capi.ROUTE_QUAD_CHAIN, "quad")):
    for lim in (1.5, 5.0):
        g = BatchILQR("acrobot", B, T, 0.02, u_min=-lim, u_max=lim, route=route, params=dict(max_iter=100))
        x0 = acrobot_x0(B)
        g.init_traj(x0, np.zeros((B, T, 1)))
        t0 = time.perf_counter()
        g.generate_trajectory()
        c = g.cost()
        dt = time.perf_counter() - t0
        st, it, al = g.status()
        print(name, "lim", lim, "full solve %.1f ms, mean iterations %.1f, mean cost %.4f, converged %.2f" % (dt * 1e3, it.mean(), c.mean(), (st != 4).mean()))
        g.close()
