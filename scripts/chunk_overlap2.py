"""Experiment: 2-3 sub-batches on separate streams with STAGGERED starts (different stages overlap)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilqr_amd import BatchILQR, capi
from tests.util import acrobot_x0
B, T, steps, warm = 4096, 499, 40, 3
x0 = acrobot_x0(B)
for C, stagger_ms in ((1, 0), (2, 0), (2, 0.3), (2, 0.5), (2, 0.7), (3, 0.4), (4, 0.3)):
    per = B // C
    per -= per % 64
    gs = []
    for c in range(C):
        g = BatchILQR("acrobot", per, T, 0.02, u_min=-1.5, u_max=1.5, flags=capi.FLAG_FIXED_WORK)
        g.init_traj(x0[c * per:(c + 1) * per], np.zeros((per, T, 1)))
        g.iterate(warm)
        gs.append(g)
    for g in gs: g.synchronize()
    t0 = time.perf_counter()
    for i, g in enumerate(gs):
        if i and stagger_ms:
            t1 = time.perf_counter()
            while (time.perf_counter() - t1) * 1e3 < stagger_ms: pass
        g.iterate(steps)
    for g in gs: g.synchronize()
    el = time.perf_counter() - t0
    print("chunks %d stagger %.1f ms: %.3f ms/iteration  %.3e timesteps/s" % (C, stagger_ms, el / steps * 1e3, per * C * T * steps / el))
    for g in gs: g.close()
