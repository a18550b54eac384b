"""PCIe-inclusive rate: host buffers in (x0, u0), results out (xs, us, k, K, cost), 20 fixed-work iterations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0
B, T, iters = 4096, 499, 20
x0 = acrobot_x0(B); u0 = np.zeros((B, T, 1))
g = BatchILQR("acrobot", B, T, 0.02, u_min=-1.5, u_max=1.5, flags=capi.FLAG_FIXED_WORK)
g.init_traj(x0, u0); g.iterate(2); g.synchronize()
from ilqr_amd.batch import _p
# result buffers a caller keeps across solves (touched once: a fresh np.zeros pays ~40 K page faults inside the copy)
oxs = np.zeros((B, T + 1, 4)); ous = np.zeros((B, T, 1)); ok = np.zeros((B, T, 1)); oK = np.zeros((B, T, 4, 1)); oc = np.zeros(B)
for a in (oxs, ous, ok, oK, oc):
    a.fill(1.0)
for rep in range(3):
    fresh = rep == 0
    t0 = time.perf_counter(); g.init_traj(x0, u0); t1 = time.perf_counter()
    g.iterate(iters); g.synchronize(); t2 = time.perf_counter()
    if fresh:
        xs, us = g.trajectory(); k, K = g.gains(); c = g.cost()
    else:
        capi.check(g.lib.ilqr_get_trajectory(g.h, _p(oxs), _p(ous))); capi.check(g.lib.ilqr_get_gains(g.h, _p(ok), _p(oK)))
        capi.check(g.lib.ilqr_get_cost(g.h, _p(oc)))
    t3 = time.perf_counter()
    print(("[fresh result arrays] " if fresh else "[reused result arrays] ") + "upload+init %.2f ms | %d iterations %.2f ms | download xs,us,k,K,cost %.2f ms | resident %.3e ts/s | PCIe-inclusive %.3e ts/s"
          % ((t1 - t0) * 1e3, iters, (t2 - t1) * 1e3, (t3 - t2) * 1e3, B * T * iters / (t2 - t1), B * T * iters / (t3 - t0)))
