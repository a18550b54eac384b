"""Counters of the backward wavefront late in a solve (experiment build -DILQR_PHASE_TIMING, ILQR_AMD_LIB): run N iterations,
then one more per-stage iteration whose counters are printed when the handle closes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from ilqr_amd import BatchILQR, capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
B, T = 4096, 499
g = BatchILQR("acrobot", B, T, 0.02, u_min=-1.5, u_max=1.5, dtype=dtype, flags=capi.FLAG_FIXED_WORK | capi.FLAG_STAGED, params=dict(max_iter=1000))
rng = np.random.default_rng(1234)
g.init_traj(rng.uniform(-1, 1, (B, 4)) * np.array([np.pi, np.pi, 1, 1]), np.zeros((B, T, 1)))
g.iterate(N)
g.profile(True)
g.profile_reset()
g.iterate(5)
print({k: round(ms / max(n, 1), 4) for k, (ms, n) in g.profile_read().items() if n})
lam, dlam = g.lambdas()
print("lambda == 0: %.3f   lambda > 1: %.4f" % ((lam == 0).mean(), (lam > 1).mean()))
g.close()
