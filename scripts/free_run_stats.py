"""Free-running full solves, device against oracle, several seeds: status distribution and median log10 cost (chaotic dynamics:
the two sides follow different paths after a few iterations; what must agree is the statistics)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import BatchILQR
from oracle import oracle as O
from tests.util import acrobot_x0
B, T, DT = 1024, 499, 0.02
O.build()
om = O.Model("acrobot")
d = []
for seed in (77, 78, 79, 80):
    x0 = acrobot_x0(B, scale=1.0, seed=seed)
    u0 = np.zeros((B, T, 1))
    g = BatchILQR("acrobot", B, T, DT)
    g.init_traj(x0, u0)
    g.generate_trajectory()
    cost, (st, it, al) = g.cost(), g.status()
    g.close()
    ro = O.batch_solve(om, x0, u0, DT)
    md, mo = np.median(np.log10(cost)), np.median(np.log10(ro["cost"]))
    d.append(md - mo)
    print("seed %d: device median %.3f mean %.3f iters %.1f | oracle median %.3f mean %.3f iters %.1f | status dev %s orc %s" % (
        seed, md, np.mean(np.log10(cost)), it.mean(), mo, np.mean(np.log10(ro["cost"])), ro["iters"].mean(),
        [int((st == s).sum()) for s in (1, 2, 3, 4)], [int((ro["status"] == s).sum()) for s in (1, 2, 3, 4)]))
print("median differences", np.round(d, 3), "mean %.3f" % np.mean(d))
