"""Acrobot headline workload, 10 iterations in the persistent kernel; with an experiment build (-DILQR_PHASE_TIMING,
ILQR_AMD_LIB) the chip clock of each phase (shader cycles over wall ticks) is printed when the handle closes.
    python scripts/clock_probe.py [extra ilqr_flags]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd import BatchILQR, capi
fl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
B, T = 4096, 499
g = BatchILQR("acrobot", B, T, 0.02, u_min=-1.5, u_max=1.5, flags=capi.FLAG_FIXED_WORK | fl, params=dict(max_iter=1000))
rng = np.random.default_rng(0)
g.init_traj(rng.uniform(-1, 1, (B, 4)) * np.array([np.pi, np.pi, 1, 1]) * 0.5, np.zeros((B, T, 1)))
g.iterate(3)
g.profile(True); g.profile_reset()
g.iterate(10)
print({k: round(ms / max(n, 1), 4) for k, (ms, n) in g.profile_read().items() if n})
g.close()
