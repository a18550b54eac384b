"""What the per-stage HIP-event timers (ilqr_profile) cost inside the timed region of bench.py: 20 fixed-work
iterations of the headline workload with the timers off and on (measured: 0.776 vs 0.787 ms per iteration with two event records per kernel boundary, 0.775-0.781 vs 0.781 with one)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0
B, T = 4096, 499
g = BatchILQR("acrobot", B, T, 0.02, u_min=-1.5, u_max=1.5, flags=capi.FLAG_FIXED_WORK)
x0 = acrobot_x0(B); u0 = np.zeros((B, T, 1))
for prof in (False, True, False, True):
    g.init_traj(x0, u0); g.iterate(3); g.synchronize()
    g.profile(prof)
    t0 = time.perf_counter(); g.iterate(20); g.synchronize(); t1 = time.perf_counter()
    g.profile(False)
    print("profile", prof, "%.4f ms/iter" % ((t1 - t0) * 1e3 / 20))
