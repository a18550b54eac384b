"""Double integrator, per-stage route, 10 iterations: stage times; with an experiment build (-DILQR_PHASE_TIMING, ILQR_AMD_LIB)
the generic box-QP's counters are printed when the handle closes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd import BatchILQR, capi
B, T = 4096, 100
g = BatchILQR("integrator", B, T, 0.02, u_min=-0.5, u_max=0.5, goal=[1.0, 0.5, 0.0, 0.0], flags=capi.FLAG_FIXED_WORK | capi.FLAG_STAGED, params=dict(max_iter=100))
rd = np.random.default_rng(4321)
g.init_traj(rd.uniform(-1, 1, size=(B, 4)) * np.array([1.5, 1.5, 0.5, 0.5]), np.zeros((B, T, 2)))
g.iterate(3)
g.profile(True); g.profile_reset()
g.iterate(10)
print({k: (round(ms / max(n, 1), 4), n) for k, (ms, n) in g.profile_read().items()})
k, K = g.gains()
lo, hi = -0.5 - g.trajectory()[1], 0.5 - g.trajectory()[1]
print("clamped fraction", ((np.abs(k - lo) < 1e-9) | (np.abs(k - hi) < 1e-9)).mean())
g.close()
