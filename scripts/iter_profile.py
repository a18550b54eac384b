"""Per-iteration stage times over a long fixed-work solve of the bench workload (where does the 0.80 -> 1.1 ms
drift come from?).   python scripts/iter_profile.py [dtype] [limit] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0

dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
B, T = 4096, 499
g = BatchILQR("acrobot", B, T, 0.02, u_min=-lim, u_max=lim, dtype=dtype, flags=capi.FLAG_FIXED_WORK, params=dict(max_iter=iters + 2))
g.init_traj(acrobot_x0(B), np.zeros((B, T, 1)))
g.profile(True)
rows = []
for it in range(iters):
    g.profile_reset()
    g.iterate(1)
    p = g.profile_read()
    lam = g.lambdas()[0]
    st, _, al = g.status()
    rows.append((it, p["backward"][0], p["rollout"][0], p["accept"][0], float(np.mean(lam == 0)), float(np.mean(al >= 0)), float(np.median(g.cost()))))
for r in rows:
    if r[0] < 10 or r[0] % 5 == 0:
        print("it %3d  backward %.3f  rollout %.3f  commit %.3f ms   lambda==0: %.2f  accepted: %.2f  median cost %.1f" % r)
