"""Stage times of the LQ path end to end at the BASELINE configs[4] shape (n=32, m=16, T=200).
    python scripts/bench_lq.py [B] [iters] [extra ilqr_flags] [ilqr_route]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi
from tests.test_gpu_lq_end_to_end import lq_mats

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
extra_flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # e.g. 16 = exact derivatives
route = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # enum ilqr_route, e.g. 128 = the LDS backward kernel
n, m, T, DT = int(os.environ.get("LQ_N", 32)), int(os.environ.get("LQ_M", 16)), 200, 0.02  # (BASELINE configs[4]: 32, 16)
mats = lq_mats(n, m)
rng = np.random.default_rng(0)
x0 = rng.uniform(-1, 1, (B, n))
u0 = np.zeros((B, T, m))
g = BatchILQR("lq", B, T, DT, u_min=-1.0, u_max=1.0, lq=mats, flags=capi.FLAG_FIXED_WORK | extra_flags, route=route)
c0 = g.init_traj(x0, u0)
g.iterate(1)
g.profile(True)
g.profile_reset()
g.synchronize()
t0 = time.perf_counter()
g.iterate(iters)
g.synchronize()
dt = time.perf_counter() - t0
p = g.profile_read()
print("LQ n=%d m=%d T=%d B=%d: %.1f ms per iteration -> %.3e trajectory-timesteps/s" % (n, m, T, B, dt / iters * 1e3, B * T * iters / dt))
for k, (ms, launches) in p.items():
    if launches:
        print("  %-12s %8.2f ms per launch (%d launches)" % (k, ms / launches, launches))
print("  cost: initial mean %.4g -> %.4g" % (c0.mean(), g.cost().mean()))
